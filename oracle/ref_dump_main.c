/*
 * ref_dump_main.c -- OUR driver around the COMPILED reference objects (oracle/_ref only).
 * TEST INFRASTRUCTURE: produces the golden vectors in tests/golden/ and, optionally, the
 * `"kind": "reference"` CPU baseline.  Contains no reference code: it calls the reference's
 * own entry points in the order of its main() (src/hybrid.f6.c:54-88) and then dumps the
 * arrays the reference never prints (src/solver.c:302-311 prints timings only).
 *
 *   ref_dump dump  PREFIX LVL VARIANT VARFILE OUTPREFIX    (one MPI rank per domain)
 *   ref_dump time  PREFIX LVL NSAMPLES WITH_FLUX           (comm_free timing loop)
 *
 * VARIANT: comm_free | mpi_bulk_sync.  VARFILE: "<VARFILE>_<rank>.bin" holds var[nall][7]
 * (raw doubles) or "-" for the reference's own init (all 1.0, src/solver_data.c:26-36).
 * Output: "<OUTPREFIX>_grad_<rank>.bin" [nall][7][3], "<OUTPREFIX>_flux_<rank>.bin" [nall][3].
 * A single domain of a multi-domain mesh can be run with 1 rank: comm_data stays zeroed
 * with ndomains = 1 and read_communication_data is not called (ASSERT(ndomains == nProc),
 * src/comm_data.c:94).
 */
#include <mpi.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "comm_data.h"
#include "flux.h"
#include "gradients.h"
#include "rangelist.h"
#include "solver_data.h"
#include "util.h"

int cfdp_nc_open(const char *path);
void cfdp_nc_close(int ncid);

static void die(const char *msg) {
  fprintf(stderr, "ref_dump: %s\n", msg);
  exit(1);
}

int main(int argc, char *argv[]) {
  if (argc < 6) die("usage: ref_dump dump|time PREFIX LVL ...");
  const int timing = strcmp(argv[1], "time") == 0;
  const char *prefix = argv[2];
  const int lvl = atoi(argv[3]);
  char *env = getenv("OMP_NUM_THREADS");
  const int NTHREADS = env ? atoi(env) : 1;
  omp_set_num_threads(NTHREADS);

  comm_data cd;
  solver_data sd;
  init_communication(argc, argv, &cd);
  const char *dom_env = getenv("REF_DUMP_DOMAIN"); /* run domain d of a mesh with 1 rank */
  const int domain = dom_env ? atoi(dom_env) : cd.iProc;
  char fname[4096];
  snprintf(fname, sizeof fname, "%s_domain_%d_lvl_%d", prefix, domain, lvl);
  int ncid = cfdp_nc_open(fname);
  read_solver_data(ncid, &sd);
  init_solver_data(&sd, 25);
  if (dom_env && cd.nProc == 1) {
    cd.ndomains = 1;
    cd.nownpoints = sd.nownpoints;
  } else {
    read_communication_data(ncid, &cd);
  }
  compute_communication_tables(&cd);
  init_threads(&cd, &sd, NTHREADS);

  if (timing) {
    const int nsamples = atoi(argv[4]), with_flux = atoi(argv[5]);
    double best = 1e30, *samples = malloc((size_t)nsamples * sizeof(double));
    for (int k = 0; k < nsamples; k++) {
      double t = -now();
#pragma omp parallel default(none) shared(cd, sd, with_flux)
      {
        for (int i = 0; i < sd.niter; ++i) {
          compute_gradients_gg_comm_free(&cd, &sd, i == sd.niter - 1);
          if (with_flux) compute_psd_flux(&sd);
#pragma omp barrier
        }
      }
      t += now();
      samples[k] = t;
      if (t < best) best = t;
    }
    for (int i = 0; i < nsamples; i++) /* insertion sort */
      for (int j = i; j > 0 && samples[j] < samples[j - 1]; j--) {
        double s = samples[j]; samples[j] = samples[j - 1]; samples[j - 1] = s;
      }
    printf("REF_TIME threads=%d niter=%d nsamples=%d with_flux=%d median_s=%.6f best_s=%.6f\n",
           NTHREADS, sd.niter, nsamples, with_flux, samples[nsamples / 2], best);
    free(samples);
  } else {
    if (argc < 7) die("usage: ref_dump dump PREFIX LVL VARIANT VARFILE OUTPREFIX");
    const char *variant = argv[4], *varfile = argv[5], *outprefix = argv[6];
    if (strcmp(varfile, "-") != 0) {
      snprintf(fname, sizeof fname, "%s_%d.bin", varfile, domain);
      FILE *fp = fopen(fname, "rb");
      if (!fp) die("cannot open var file");
      size_t n = (size_t)sd.nallpoints * NGRAD;
      if (fread(&sd.var[0][0], sizeof(double), n, fp) != n) die("short var file");
      fclose(fp);
    }
    const int bulk = strcmp(variant, "mpi_bulk_sync") == 0;
    if (!bulk && strcmp(variant, "comm_free") != 0) die("unknown variant");
    MPI_Barrier(MPI_COMM_WORLD);
#pragma omp parallel default(none) shared(cd, sd, bulk)
    {
      if (bulk) compute_gradients_gg_mpi_bulk_sync(&cd, &sd, 1);
      else compute_gradients_gg_comm_free(&cd, &sd, 1);
#pragma omp barrier
      compute_psd_flux(&sd);
#pragma omp barrier
    }
    MPI_Barrier(MPI_COMM_WORLD);
    snprintf(fname, sizeof fname, "%s_grad_%d.bin", outprefix, domain);
    FILE *fp = fopen(fname, "wb");
    if (!fp) die("cannot write grad");
    fwrite(&sd.grad[0][0][0], sizeof(double), (size_t)sd.nallpoints * NGRAD * 3, fp);
    fclose(fp);
    snprintf(fname, sizeof fname, "%s_flux_%d.bin", outprefix, domain);
    fp = fopen(fname, "wb");
    if (!fp) die("cannot write flux");
    fwrite(&sd.psd_flux[0][0], sizeof(double), (size_t)sd.nallpoints * NFLUX, fp);
    fclose(fp);
  }
  cfdp_nc_close(ncid);
  free_communication_ressources(&cd);
  if (cd.ndomains == 1) MPI_Finalize(); /* the reference only finalises when ndomains > 1 */
  return 0;
}
