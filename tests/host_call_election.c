/*
 * host_call_election.c -- TEST HOST (compiled by tests/test_host_mesh.py, CPU only) of the election that decides
 * which of the threads calling compute_gradients_gg_* / compute_psd_flux enqueues the work (host/call_election.c;
 * the reference's convention: every thread of one omp parallel region makes every call, src/solver.c:45-55).
 *
 *   host_call_election SCENARIO      prints "performed P of C"; exit code 0 unless the library stops the run
 */
#include <omp.h>
#include <pthread.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include "call_election.h"

static cfdp_election el;
static int performed = 0, order_ok = 1, last_kind = 0;

static void call(int kind) { /* what dropin.c's gradients() / compute_psd_flux() do around their GPU calls */
  if (!cfdp_elect_begin(&el, kind)) return;
  if (kind == last_kind) order_ok = 0; /* gradients and flux must alternate in the order they were issued */
  last_kind = kind;
  performed++;
  cfdp_elect_end(&el);
}

static void team_iterations(int nthreads, int niter, int stagger) {
#pragma omp parallel num_threads(nthreads)
  for (int i = 0; i < niter; i++) { /* src/solver.c:45-55 */
    if (stagger && omp_get_thread_num() == (i % nthreads)) usleep(200); /* team mates run ahead of this one */
    call(1);
    call(64);
#pragma omp barrier
  }
}

static void *serial_thread(void *arg) {
  (void)arg;
  call(1);
  call(64);
  return NULL;
}

static void *pthread_member(void *arg) {
  (void)arg;
  for (int i = 0; i < 50; i++) {
    call(1);
    call(64);
    if (i % 7 == 0) usleep(100);
  }
  return NULL;
}

int main(int argc, char **argv) {
  const char *sc = argc > 1 ? argv[1] : "team";
  int calls = 0;
  cfdp_elect_init(&el);
  if (!strcmp(sc, "team")) { /* the reference's harness: 4 threads, every thread makes every call */
    team_iterations(4, 200, 1);
    calls = 400;
  } else if (!strcmp(sc, "serial_threads")) { /* serial callers, a different thread each time */
    for (int i = 0; i < 6; i++) {
      pthread_t t;
      pthread_create(&t, NULL, serial_thread, NULL);
      pthread_join(t, NULL);
    }
    serial_thread(NULL);
    calls = 14;
  } else if (!strcmp(sc, "mixed")) { /* serial calls, a team of 4, a team of 2, serial calls, a team of 4 again */
    serial_thread(NULL);
    team_iterations(4, 30, 1);
    team_iterations(2, 17, 1);
    serial_thread(NULL);
    team_iterations(4, 30, 0);
    calls = 2 + 60 + 34 + 2 + 60;
  } else if (!strcmp(sc, "pthread_team")) { /* a team that is not an OpenMP team: 4 pthreads, each makes every call */
    cfdp_set_call_team(4);
    pthread_t t[4];
    for (int i = 0; i < 4; i++) pthread_create(&t[i], NULL, pthread_member, NULL);
    for (int i = 0; i < 4; i++) pthread_join(t[i], NULL);
    cfdp_set_call_team(0);
    calls = 100;
  } else if (!strcmp(sc, "master")) { /* one thread of a team of 4 makes all calls */
#pragma omp parallel num_threads(4)
    for (int i = 0; i < 40; i++) {
#pragma omp master
      {
        call(1);
        call(64);
      }
#pragma omp barrier
    }
    calls = 80;
  } else if (!strcmp(sc, "team_then_master")) { /* the reference's regions, then master sections */
    team_iterations(4, 30, 1);
#pragma omp parallel num_threads(4)
    for (int i = 0; i < 50; i++) {
#pragma omp master
      {
        call(1);
        call(64);
      }
#pragma omp barrier
    }
    calls = 60 + 100;
  } else if (!strcmp(sc, "single")) { /* omp single sections: whichever thread gets there first makes the call */
#pragma omp parallel num_threads(4)
    for (int i = 0; i < 40; i++) {
      if ((omp_get_thread_num() + i) % 3 == 0) usleep(100);
#pragma omp single
      {
        call(1);
        call(64);
      }
    }
    calls = 80;
  }
  pthread_mutex_lock(&el.mtx); /* (the counters were last written under it by threads of an OpenMP pool: a happens-before
                                   edge a thread sanitizer can see -- libgomp's own joins are invisible to it) */
  pthread_mutex_unlock(&el.mtx);
  printf("performed %d of %d%s\n", performed, calls, order_ok ? "" : " (order broken)");
  cfdp_elect_destroy(&el);
  return performed == calls && order_ok ? 0 : 3;
}
