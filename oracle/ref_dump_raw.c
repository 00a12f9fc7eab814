/*
 * ref_dump_raw.c -- OUR driver around the COMPILED reference objects (oracle/_ref only).
 * TEST INFRASTRUCTURE: produces the golden vectors in tests/golden/ and the
 * `"kind": "reference"` CPU baseline.  Contains no reference code and links NO product code:
 * the link line of oracle/_ref/ref_dump_raw holds the reference's own translation units
 * (compiled where they lie in /root/reference/src), this file, MPICH and libm -- nothing
 * from cfd-proxy_amd/, no NetCDF reader of any kind.
 *
 * The reference fills solver_data / comm_data from a NetCDF file (read_solver_data,
 * src/solver_data.c:80-160; read_communication_data, src/comm_data.c:74-114).  libnetcdf is not
 * in this image, so this driver fills the same fields from a raw array file instead -- the
 * fields and allocations of src/solver_data.c:96-122 and src/comm_data.c:79-112, no I/O library
 * -- and then calls the reference's own entry points in the order of its main()
 * (src/hybrid.f6.c:54-88): init_communication, init_solver_data, compute_communication_tables,
 * init_threads, compute_gradients_gg_<variant>, compute_psd_flux.  The reference's
 * read_solver_data/read_communication_data/get_nc_* are never referenced and are dropped by
 * --gc-sections together with their libnetcdf calls.
 *
 *   ref_dump_raw dump  RAWPREFIX VARIANT OUTPREFIX      (one MPI rank per domain)
 *   ref_dump_raw time  RAWPREFIX NSAMPLES WITH_FLUX     (comm_free timing loop, 1 rank)
 *
 * Raw file "<RAWPREFIX>_<domain>.raw" (written by oracle/cpu_ref.py:write_raw_domain):
 *   int32  magic 0x43464450, nfaces, nownpoints, nallpoints, ndomains, naddpoints, ncommdomains, has_var
 *   int32  fpoint[nfaces][2];  double fnormal[nfaces][3];  double pvolume[nallpoints]
 *   double var[nallpoints][7]                      (has_var; else the reference's own init, all 1.0)
 *   int32  commpartner[ncommdomains], sendcount[ndomains], recvcount[ndomains],
 *          addpoint_owner[naddpoints], addpoint_id[naddpoints]          (ndomains > 1)
 * VARIANT: comm_free | mpi_bulk_sync.  Output: "<OUTPREFIX>_grad_<domain>.bin" [nall][7][3],
 * "<OUTPREFIX>_flux_<domain>.bin" [nall][3].  REF_DUMP_DOMAIN=d with one rank runs domain d of a
 * multi-domain mesh alone: comm_data keeps ndomains = 1 (ASSERT(ndomains == nProc),
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

static void die(const char *msg) {
  fprintf(stderr, "ref_dump_raw: %s\n", msg);
  exit(1);
}

static void rd(void *dst, size_t sz, size_t n, FILE *fp) {
  if (n && fread(dst, sz, n, fp) != n) die("short raw file");
}

static void *xmalloc(size_t n) {
  void *p = malloc(n ? n : 1);
  if (!p) die("out of memory");
  return p;
}

int main(int argc, char *argv[]) {
  if (argc < 5) die("usage: ref_dump_raw dump RAWPREFIX VARIANT OUTPREFIX | time RAWPREFIX NSAMPLES WITH_FLUX");
  const int timing = strcmp(argv[1], "time") == 0;
  const char *prefix = argv[2];
  char *env = getenv("OMP_NUM_THREADS");
  const int NTHREADS = env ? atoi(env) : 1;
  omp_set_num_threads(NTHREADS);

  comm_data cd;
  solver_data sd;
  init_communication(argc, argv, &cd);
  const char *dom_env = getenv("REF_DUMP_DOMAIN");
  const int domain = dom_env ? atoi(dom_env) : cd.iProc;
  char fname[4096];
  snprintf(fname, sizeof fname, "%s_%d.raw", prefix, domain);
  FILE *fp = fopen(fname, "rb");
  if (!fp) die("cannot open raw file");
  int hdr[8];
  rd(hdr, sizeof(int), 8, fp);
  if (hdr[0] != 0x43464450) die("bad magic");

  /* the fields read_solver_data sets (src/solver_data.c:84-122); the file's colour lists are
   * freed unread by init_threads (src/threads.c:748-749), so one empty RangeList stands for them */
  memset(&sd, 0, sizeof sd);
  sd.nfaces = hdr[1];
  sd.nownpoints = hdr[2];
  sd.nallpoints = hdr[3];
  sd.ncolors = 1;
  sd.fpoint = xmalloc((size_t)sd.nfaces * 2 * sizeof(int));
  sd.fnormal = xmalloc((size_t)sd.nfaces * 3 * sizeof(double));
  sd.pvolume = xmalloc((size_t)sd.nallpoints * sizeof(double));
  sd.var = xmalloc((size_t)sd.nallpoints * NGRAD * sizeof(double));
  sd.grad = xmalloc((size_t)sd.nallpoints * NGRAD * 3 * sizeof(double));
  sd.psd_flux = xmalloc((size_t)sd.nallpoints * NFLUX * sizeof(double));
  sd.fcolor = xmalloc(sizeof(RangeList));
  init_rangelist(sd.fcolor);
  sd.fcolor->all_points_of_color = xmalloc(sizeof(int));
  rd(&sd.fpoint[0][0], sizeof(int), (size_t)sd.nfaces * 2, fp);
  rd(&sd.fnormal[0][0], sizeof(double), (size_t)sd.nfaces * 3, fp);
  rd(sd.pvolume, sizeof(double), (size_t)sd.nallpoints, fp);
  init_solver_data(&sd, 25);
  if (hdr[7]) rd(&sd.var[0][0], sizeof(double), (size_t)sd.nallpoints * NGRAD, fp);

  /* the fields read_communication_data sets (src/comm_data.c:79-112) */
  if (dom_env && cd.nProc == 1) {
    cd.ndomains = 1;
    cd.nownpoints = sd.nownpoints;
  } else {
    cd.ndomains = hdr[4];
    cd.nownpoints = sd.nownpoints;
    if (cd.ndomains != cd.nProc) die("ndomains != number of MPI ranks");
    if (cd.ndomains > 1) {
      cd.naddpoints = hdr[5];
      cd.ncommdomains = hdr[6];
      if (cd.naddpoints <= 0 || cd.ncommdomains <= 0) die("no halo in a multi-domain file");
      cd.commpartner = xmalloc((size_t)cd.ncommdomains * sizeof(int));
      cd.sendcount = xmalloc((size_t)cd.ndomains * sizeof(int));
      cd.recvcount = xmalloc((size_t)cd.ndomains * sizeof(int));
      cd.addpoint_owner = xmalloc((size_t)cd.naddpoints * sizeof(int));
      cd.addpoint_id = xmalloc((size_t)cd.naddpoints * sizeof(int));
      rd(cd.commpartner, sizeof(int), (size_t)cd.ncommdomains, fp);
      rd(cd.sendcount, sizeof(int), (size_t)cd.ndomains, fp);
      rd(cd.recvcount, sizeof(int), (size_t)cd.ndomains, fp);
      rd(cd.addpoint_owner, sizeof(int), (size_t)cd.naddpoints, fp);
      rd(cd.addpoint_id, sizeof(int), (size_t)cd.naddpoints, fp);
    }
  }
  fclose(fp);
  compute_communication_tables(&cd);
  init_threads(&cd, &sd, NTHREADS);

  if (timing) {
    const int nsamples = atoi(argv[3]), with_flux = atoi(argv[4]);
    if (nsamples < 1) die("NSAMPLES < 1");
    double best = 1e30, *samples = xmalloc((size_t)nsamples * sizeof(double));
    for (int k = 0; k < nsamples; k++) {
      double t = -now();
#pragma omp parallel default(none) shared(cd, sd, with_flux)
      {
        for (int i = 0; i < sd.niter; ++i) { /* the timed loop of src/solver.c:42-58 */
          compute_gradients_gg_comm_free(&cd, &sd, i == sd.niter - 1);
          if (with_flux) compute_psd_flux(&sd);
#pragma omp barrier
        }
      }
      t += now();
      samples[k] = t;
      if (t < best) best = t;
    }
    for (int i = 0; i < nsamples; i++)
      for (int j = i; j > 0 && samples[j] < samples[j - 1]; j--) {
        double s = samples[j]; samples[j] = samples[j - 1]; samples[j - 1] = s;
      }
    printf("REF_TIME threads=%d niter=%d nsamples=%d with_flux=%d median_s=%.6f best_s=%.6f\n",
           NTHREADS, sd.niter, nsamples, with_flux, samples[nsamples / 2], best);
    free(samples);
  } else {
    const char *variant = argv[3], *outprefix = argv[4];
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
    fp = fopen(fname, "wb");
    if (!fp) die("cannot write grad");
    fwrite(&sd.grad[0][0][0], sizeof(double), (size_t)sd.nallpoints * NGRAD * 3, fp);
    fclose(fp);
    snprintf(fname, sizeof fname, "%s_flux_%d.bin", outprefix, domain);
    fp = fopen(fname, "wb");
    if (!fp) die("cannot write flux");
    fwrite(&sd.psd_flux[0][0], sizeof(double), (size_t)sd.nallpoints * NFLUX, fp);
    fclose(fp);
  }
  free_communication_ressources(&cd);
  if (cd.ndomains == 1) MPI_Finalize(); /* the reference only finalises when ndomains > 1 */
  return 0;
}
