/*
 * call_election.c -- one enqueue per entry-point call, whichever threads make it.
 *
 * The reference calls compute_gradients_gg_<variant>() and compute_psd_flux() from EVERY thread of one
 * `omp parallel` region, with no barrier between the two and one `omp barrier` per iteration
 * (src/solver.c:45-55); its exchange functions elect their first / last thread (src/threads.c:142-179).
 * On the GPU one of the callers has to enqueue the work of a call and the others must not.  Who:
 *
 *   * a caller OUTSIDE a parallel region (or in a team of one) performs every call it makes -- whichever
 *     thread it is: serial hosts, pthread hosts, a host that replaces its calling thread;
 *   * inside a team of T > 1 threads the reference's convention holds: every thread of the team makes every
 *     call.  The first thread to make its k-th team call performs call k; the others find it done.  Threads
 *     may run ahead of each other (gradients and flux are not separated by a barrier), so the ordinal is
 *     counted per thread;
 *   * a team that does NOT follow the convention -- calls issued from `omp single` sections, which a varying
 *     thread executes -- cannot be told from a late team mate when the call arrives; it is detected from the
 *     attendance of earlier calls: when call k is performed, every call up to k - CFDP_ELECT_GRACE must have
 *     been attended by its whole team (the reference's per-iteration barrier guarantees that after two calls).
 *     (A team in which ONE thread makes all calls -- omp master -- loses nothing and is left alone: a call that only
 *     its performer attended counts against the team only if ANOTHER thread performed it.  That is judged per call, not
 *     once per solver: a host may run the reference's every-thread regions first and master sections later (going BACK
 *     to every-thread regions after master sections is not supported in AUTO mode: the threads that sat out cannot be
 *     told from late team mates of the last master calls, and the run stops with the message below).  A team whose
 *     calls rotate evenly over all its T threads produces exactly the arrivals of T late team mates and cannot be
 *     told apart by anything the library sees: such a host has to select CFDP_CALLS_EVERY itself.)
 *     The run then stops with a message naming the fix (cfdp_set_call_mode(CFDP_CALLS_EVERY) or
 *     CFDP_CALL_MODE=every: no election, every call is performed) instead of returning stale gradients.
 *     A thread whose own ordinal points at a call everybody has attended already (it sat out a region run by a
 *     smaller team) is moved forward to the first call that is still open.
 */
#define _GNU_SOURCE
#include "call_election.h"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static int g_mode = -1; /* -1: read CFDP_CALL_MODE on first use */

void cfdp_set_call_mode(int mode) { __atomic_store_n(&g_mode, mode, __ATOMIC_RELAXED); }

int cfdp_get_call_mode(void) { /* (every thread of a team gets here at once: the lazy read is an atomic, -fsanitize=thread found it) */
  int m = __atomic_load_n(&g_mode, __ATOMIC_RELAXED);
  if (m < 0) {
    const char *e = getenv("CFDP_CALL_MODE");
    m = CFDP_CALLS_AUTO;
    if (e && !strcmp(e, "every")) m = CFDP_CALLS_EVERY;
    else if (e && !strcmp(e, "team")) m = CFDP_CALLS_TEAM;
    else if (e && *e && strcmp(e, "auto")) {
      fprintf(stderr, "Error: CFDP_CALL_MODE=%s (one of auto, team, every)\n", e);
      exit(EXIT_FAILURE);
    }
    __atomic_store_n(&g_mode, m, __ATOMIC_RELAXED);
  }
  return m;
}

/* ---- "is the caller inside a parallel team, and how large is it?"
 * A process can hold MORE than one OpenMP runtime (a gcc-built host brings libgomp, a library linked by hipcc brings
 * LLVM's libomp; both export omp_in_parallel, and which one a link-time reference of THIS library reaches is decided
 * by symbol versions, not by who created the region).  The runtime that knows the caller's team is the one the HOST's
 * `#pragma omp parallel` went to: the first in the process's global lookup order, which is what dlsym(RTLD_DEFAULT)
 * returns -- asked at run time, by name.  No other runtime is touched: a call into a runtime the host does not use
 * would INITIALISE it (LLVM's libomp then installs fork handlers and helper threads in a process that never asked
 * for them -- a Python host's next subprocess call hung on exactly that).  No OpenMP runtime in the global scope
 * (a ctypes host, a pthread host): no teams.  cfdp_set_call_team(T) states the team for hosts whose team is not an
 * OpenMP team.                                                                                                    */
typedef int (*omp_int_fn)(void);
static omp_int_fn g_in_parallel = NULL, g_num_threads = NULL;
static int g_team_override = 0;

void cfdp_set_call_team(int nthreads) { __atomic_store_n(&g_team_override, nthreads > 0 ? nthreads : 0, __ATOMIC_RELAXED); }

static int caller_team_size(void) {
  const int ov = __atomic_load_n(&g_team_override, __ATOMIC_RELAXED);
  if (ov) return ov;
  omp_int_fn a = __atomic_load_n(&g_in_parallel, __ATOMIC_ACQUIRE), b = __atomic_load_n(&g_num_threads, __ATOMIC_ACQUIRE);
  if (!a || !b) { /* (looked up again while absent: a host may load its runtime later) */
    a = (omp_int_fn)dlsym(RTLD_DEFAULT, "omp_in_parallel");
    b = (omp_int_fn)dlsym(RTLD_DEFAULT, "omp_get_num_threads");
    if (!a || !b) return 1;
    __atomic_store_n(&g_num_threads, b, __ATOMIC_RELEASE);
    __atomic_store_n(&g_in_parallel, a, __ATOMIC_RELEASE);
  }
  return a() ? b() : 1;
}

#define CFDP_TLS_SLOTS 16
static __thread struct { unsigned long id, n; } tls_calls[CFDP_TLS_SLOTS];
static unsigned long g_ids = 0;

void cfdp_elect_init(cfdp_election *el) {
  memset(el, 0, sizeof(*el));
  pthread_mutex_init(&el->mtx, NULL);
  el->id = __atomic_add_fetch(&g_ids, 1, __ATOMIC_RELAXED);
}

void cfdp_elect_destroy(cfdp_election *el) { pthread_mutex_destroy(&el->mtx); }

static void violation(const cfdp_election *el, unsigned long k, unsigned long late) {
  fprintf(stderr,
          "Error: compute_gradients_gg_* / compute_psd_flux are called from inside a parallel region of %d threads, but not "
          "by every thread of the team: call %lu is being made while call %lu has been attended by %d of %d threads.  "
          "Inside a parallel region the reference's convention is expected (every thread makes every call, "
          "src/solver.c:45-55).  For calls made by one thread at a time (omp single / master sections) select "
          "\"every call is performed\": cfdp_set_call_mode(CFDP_CALLS_EVERY) or CFDP_CALL_MODE=every.\n",
          el->ring[late % CFDP_ELECT_RING].team, k, late, el->ring[late % CFDP_ELECT_RING].attended,
          el->ring[late % CFDP_ELECT_RING].team);
  /* (team mates are inside GPU calls right now: exit() would run the runtimes' exit handlers under them) */
  fflush(NULL);
  _exit(EXIT_FAILURE);
}

static int trace_on(void) { /* CFDP_CALL_TRACE=1: one line per entry-point call on stderr (diagnostics) */
  static int on = -1;
  int v = __atomic_load_n(&on, __ATOMIC_RELAXED);
  if (v < 0) { const char *e = getenv("CFDP_CALL_TRACE"); v = e && *e && *e != '0'; __atomic_store_n(&on, v, __ATOMIC_RELAXED); }
  return v;
}

int cfdp_elect_begin(cfdp_election *el, int kind) {
  /* a team the host has declared itself (cfdp_set_call_team) is taken at its word: no attendance check, which
   * presumes the reference's barrier per iteration */
  const int mode = __atomic_load_n(&g_team_override, __ATOMIC_RELAXED) && cfdp_get_call_mode() == CFDP_CALLS_AUTO ? CFDP_CALLS_TEAM
                                                                                                   : cfdp_get_call_mode();
  const int resync = mode == CFDP_CALLS_AUTO;
  const int team = mode != CFDP_CALLS_EVERY ? caller_team_size() : 1;
  if (team <= 1) { /* serial caller: it performs what it calls */
    if (trace_on()) fprintf(stderr, "cfdp call: thread %lx kind %d serial caller -> performs\n", (unsigned long)pthread_self(), kind);
    pthread_mutex_lock(&el->mtx);
    el->serial_calls++;
    return 1;
  }
  int slot = -1, spare = 0;
  unsigned long oldest = ~0ul;
  for (int i = 0; i < CFDP_TLS_SLOTS && slot < 0; i++) {
    if (tls_calls[i].id == el->id) slot = i;
    else if (tls_calls[i].id < oldest) { oldest = tls_calls[i].id; spare = i; }
  }
  if (slot < 0) { /* first team call of this thread on this election; recycle the oldest one's slot */
    slot = spare;
    tls_calls[slot].id = el->id;
    tls_calls[slot].n = 0;
  }
  pthread_mutex_lock(&el->mtx);
  /* Whether this thread performs is decided by its ordinal alone: call k is new iff k > team_calls.  The ring of
   * recent calls (an entry is trusted only while it still carries its call's ordinal) serves two things: a thread
   * whose ordinal points at a call its whole team has attended was not part of that team and moves forward to the
   * first call still open; and the attendance check of CFDP_CALLS_AUTO.  Threads of a declared team (or of
   * CFDP_CALLS_TEAM) may be any distance apart (no barrier in the host): the one behind finds its calls done. */
  unsigned long k = tls_calls[slot].n + 1;
#define ENTRY(j) el->ring[(j) % CFDP_ELECT_RING]
  /* an OpenMP team (not one the host declared): re-synchronise threads that were not part of earlier teams.  A
   * thread that has never called (an OpenMP runtime replaces pool threads when teams shrink and grow) starts at the
   * oldest call still remembered; every thread passes over calls their whole team has attended */
  if (resync) {
    if (tls_calls[slot].n == 0 && el->team_calls >= CFDP_ELECT_RING) k = el->team_calls - CFDP_ELECT_RING + 1;
    while (k <= el->team_calls && ENTRY(k).ordinal == k && ENTRY(k).attended >= ENTRY(k).team) k++;
  }
  tls_calls[slot].n = k;
  if (trace_on())
    fprintf(stderr, "cfdp call: thread %lx kind %d team %d ordinal %lu of %lu -> %s\n", (unsigned long)pthread_self(), kind, team, k,
            el->team_calls, k <= el->team_calls ? "done by a team mate" : "performs");
  if (k <= el->team_calls) { /* a team mate has performed call k; this thread attends it */
    if (ENTRY(k).ordinal == k) {
      ENTRY(k).attended++;
      if (mode == CFDP_CALLS_AUTO && ENTRY(k).kind != kind) violation(el, el->team_calls, k);
    }
    pthread_mutex_unlock(&el->mtx);
    return 0;
  }
  /* call k is new: this thread performs it */
  el->team_calls = k;
  ENTRY(k).ordinal = k;
  ENTRY(k).attended = 1;
  ENTRY(k).team = team;
  ENTRY(k).kind = kind;
  ENTRY(k).performer = pthread_self();
  /* attendance: an earlier call that its team has not fully attended by now is a violation -- unless this very thread
   * performed it and nobody else came: one thread making all calls (omp master) loses none */
  if (mode == CFDP_CALLS_AUTO)
    for (unsigned long j = k > CFDP_ELECT_RING - 1 ? k - (CFDP_ELECT_RING - 1) : 1; j + CFDP_ELECT_GRACE <= k; j++)
      if (ENTRY(j).ordinal == j && ENTRY(j).attended < ENTRY(j).team &&
          (ENTRY(j).attended > 1 || !pthread_equal(ENTRY(j).performer, pthread_self())))
        violation(el, k, j);
#undef ENTRY
  return 1;
}

void cfdp_elect_end(cfdp_election *el) { pthread_mutex_unlock(&el->mtx); }
