/*
 * host_omp_driver.c -- TEST HOST of the drop-in boundary (compiled by tests/test_gpu_parity.py).
 *
 * Calls the entry points the way the reference's harness does (src/solver.c:45-55): EVERY thread of
 * one `omp parallel` region calls compute_gradients_gg_<variant>() and compute_psd_flux(), with one
 * `omp barrier` per iteration -- for each of the ten variant names in turn -- and then writes
 * grad / psd_flux of the last variant run, so that the Python side can compare with the oracle.
 * The mesh comes through the reference's own loader sequence (src/hybrid.f6.c:54-82) with
 * libnetcdf's nc_open / nc_close as the reference's main() calls them.
 *
 *   host_omp_driver PREFIX LVL NITER OUTPREFIX [VARIANT...]      (OMP_NUM_THREADS = team size)
 */
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cfdproxy_dropin.h"

typedef void (*grad_fn)(comm_data *, solver_data *, int);
static const struct { const char *name; grad_fn fn; } variants[] = {
    {"comm_free", compute_gradients_gg_comm_free},
    {"mpi_bulk_sync", compute_gradients_gg_mpi_bulk_sync},
    {"mpi_early_recv", compute_gradients_gg_mpi_early_recv},
    {"mpi_async", compute_gradients_gg_mpi_async},
    {"gaspi_bulk_sync", compute_gradients_gg_gaspi_bulk_sync},
    {"gaspi_async", compute_gradients_gg_gaspi_async},
    {"mpifence_bulk_sync", compute_gradients_gg_mpifence_bulk_sync},
    {"mpifence_async", compute_gradients_gg_mpifence_async},
    {"mpipscw_bulk_sync", compute_gradients_gg_mpipscw_bulk_sync},
    {"mpipscw_async", compute_gradients_gg_mpipscw_async},
};

int main(int argc, char *argv[]) {
  if (argc < 5) {
    fprintf(stderr, "usage: %s PREFIX LVL NITER OUTPREFIX [VARIANT...]\n", argv[0]);
    return 1;
  }
  const int niter = atoi(argv[3]);
  comm_data cd;
  solver_data sd;
  int ncid, retval;
  init_communication(argc, argv, &cd);
  char fname[4096];
  snprintf(fname, sizeof fname, "%s_domain_%d_lvl_%d", argv[1], cd.iProc, atoi(argv[2]));
  if ((retval = nc_open(fname, NC_NOWRITE, &ncid))) {
    printf("Error: %s\n", nc_strerror(retval));
    return 2;
  }
  read_solver_data(ncid, &sd);
  init_solver_data(&sd, niter);
  read_communication_data(ncid, &cd);
  compute_communication_tables(&cd);
  /* a non-constant field, the same formula the Python side uses */
  for (int i = 0; i < sd.nallpoints; i++)
    for (int e = 0; e < NGRAD; e++) sd.var[i][e] = 1.0 + 0.01 * (double)((7 * i + 13 * e) % 101);
  init_threads(&cd, &sd, omp_get_max_threads());

  int nthreads_seen = 0;
  for (size_t v = 0; v < sizeof variants / sizeof variants[0]; v++) {
    int wanted = argc == 5;
    for (int a = 5; a < argc; a++) wanted = wanted || strcmp(argv[a], variants[v].name) == 0;
    if (!wanted) continue;
    const grad_fn fn = variants[v].fn;
    /* poison what the variant must recompute, so a call that enqueued nothing cannot pass */
    for (int i = 0; i < sd.nownpoints; i++) {
      for (int e = 0; e < NGRAD; e++) sd.grad[i][e][0] = sd.grad[i][e][1] = sd.grad[i][e][2] = -777.0;
      sd.psd_flux[i][0] = sd.psd_flux[i][1] = sd.psd_flux[i][2] = -777.0;
    }
    cfdp_sync_fields_to_device(&sd);
    const double t0 = now();
#pragma omp parallel default(none) shared(cd, sd, fn, nthreads_seen)
    {
#pragma omp single
      nthreads_seen = omp_get_num_threads();
      for (int i = 0; i < sd.niter; ++i) { /* src/solver.c:45-55 */
        int final = (i == sd.niter - 1) ? 1 : 0;
        fn(&cd, &sd, final);
        compute_psd_flux(&sd);
#pragma omp barrier
      }
    }
    printf("variant %-20s threads %d  %d iterations  %.6f s\n", variants[v].name, nthreads_seen, sd.niter, now() - t0);
    cfdp_sync_fields_to_host(&sd);
    char out[4096];
    snprintf(out, sizeof out, "%s_%s_grad.bin", argv[4], variants[v].name);
    FILE *fp = fopen(out, "wb");
    if (!fp) return 3;
    fwrite(&sd.grad[0][0][0], sizeof(double), (size_t)sd.nallpoints * NGRAD * 3, fp);
    fclose(fp);
    snprintf(out, sizeof out, "%s_%s_flux.bin", argv[4], variants[v].name);
    fp = fopen(out, "wb");
    if (!fp) return 3;
    fwrite(&sd.psd_flux[0][0], sizeof(double), (size_t)sd.nallpoints * NFLUX, fp);
    fclose(fp);
  }
  free_communication_ressources(&cd);
  if ((retval = nc_close(ncid))) {
    printf("Error: %s\n", nc_strerror(retval));
    return 2;
  }
  printf("*** SUCCESS\n");
  return 0;
}
