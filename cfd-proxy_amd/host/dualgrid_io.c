/*
 * dualgrid_io.c -- the NetCDF "dualgrid" loader behind the reference's entry points.
 *
 *   (get_nc_int / get_nc_double / get_nc_val live in read_netcdf_dropin.c)
 *   read_solver_data / init_solver_data       <- reference src/solver_data.c:65-160
 *   read_communication_data                   <- reference src/comm_data.c:74-114
 *
 * Same names, argument meaning and failure behaviour (message + exit); the I/O itself
 * goes through nc_classic.c instead of libnetcdf.  Also the schema writer used by the
 * generator and the fixtures.
 */
#include "cfdproxy_host.h"
#include "nc_classic.h"
#include "host_util.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* --------------------------------------------------------------------- solver_data --- */
void read_solver_data(int ncid, solver_data *sd) {
  CFDP_ASSERT(sd != NULL);
  memset(sd, 0, sizeof(*sd));
  sd->ncolors = get_nc_val(ncid, "ncolors");
  sd->nfaces = get_nc_val(ncid, "nfaces");
  sd->nownpoints = get_nc_val(ncid, "nownpoints");
  sd->nallpoints = get_nc_val(ncid, "nallpoints");
  CFDP_ASSERT(sd->ncolors > 0);
  CFDP_ASSERT(sd->nfaces > 0);
  CFDP_ASSERT(sd->nownpoints > 0);
  CFDP_ASSERT(sd->nallpoints >= sd->nownpoints);

  const size_t nf = (size_t)sd->nfaces, np = (size_t)sd->nallpoints;
  sd->fpoint = cfdp_malloc(nf * 2 * sizeof(int));
  sd->fnormal = cfdp_malloc(nf * 3 * sizeof(double));
  sd->pvolume = cfdp_malloc(np * sizeof(double));
  sd->var = cfdp_malloc(np * NGRAD * sizeof(double));
  sd->grad = cfdp_malloc(np * NGRAD * 3 * sizeof(double));
  sd->psd_flux = cfdp_malloc(np * NFLUX * sizeof(double));
  get_nc_int(ncid, "fpoint", &sd->fpoint[0][0]);
  get_nc_double(ncid, "fnormal", &sd->fnormal[0][0]);
  get_nc_double(ncid, "pvolume", sd->pvolume);
  for (size_t f = 0; f < nf; f++) {
    CFDP_ASSERT(sd->fpoint[f][0] >= 0 && sd->fpoint[f][0] < sd->nallpoints);
    CFDP_ASSERT(sd->fpoint[f][1] >= 0 && sd->fpoint[f][1] < sd->nallpoints);
  }
  /* The colour lists stored in the file (fcolor_npoints / fcolor_points) are read and
   * then discarded by the reference (src/solver_data.c:134-160, src/threads.c:748-749);
   * the GPU tiler rebuilds its own decomposition, so they are not loaded at all.        */
  sd->fcolor = NULL;
  sd->gpu = NULL;
}

void init_solver_data(solver_data *sd, int NITER) {
  CFDP_ASSERT(sd != NULL);
  CFDP_ASSERT(sd->nallpoints != 0);
  const size_t np = (size_t)sd->nallpoints;
  double *v = &sd->var[0][0], *g = &sd->grad[0][0][0], *q = &sd->psd_flux[0][0];
  for (size_t i = 0; i < np * NGRAD; i++) v[i] = 1.0;     /* src/solver_data.c:26-36 */
  for (size_t i = 0; i < np * NGRAD * 3; i++) g[i] = 1.0; /* src/solver_data.c:38-51 */
  for (size_t i = 0; i < np * NFLUX; i++) q[i] = 1.0;     /* src/solver_data.c:53-63 */
  sd->niter = NITER;
}

void cfdp_free_solver_data(solver_data *sd) {
  if (!sd) return;
  free(sd->fpoint); free(sd->fnormal); free(sd->pvolume);
  free(sd->var); free(sd->grad); free(sd->psd_flux);
  memset(sd, 0, sizeof(*sd));
}

/* ----------------------------------------------------------------------- comm_data --- */
void read_communication_data(int ncid, comm_data *cd) {
  CFDP_ASSERT(cd != NULL);
  cd->ndomains = get_nc_val(ncid, "ndomains");
  cd->nownpoints = get_nc_val(ncid, "nownpoints");
  if (cd->ndomains == 1) return; /* "threading model only", src/comm_data.c:82-86 */

  cd->naddpoints = get_nc_val(ncid, "naddpoints");
  cd->ncommdomains = get_nc_val(ncid, "ncommdomains");
  CFDP_ASSERT(cd->ndomains >= 1);
  CFDP_ASSERT(cd->naddpoints > 0);
  CFDP_ASSERT(cd->ncommdomains > 0);
  cd->commpartner = cfdp_malloc((size_t)cd->ncommdomains * sizeof(int));
  cd->sendcount = cfdp_malloc((size_t)cd->ndomains * sizeof(int));
  cd->recvcount = cfdp_malloc((size_t)cd->ndomains * sizeof(int));
  cd->addpoint_owner = cfdp_malloc((size_t)cd->naddpoints * sizeof(int));
  cd->addpoint_id = cfdp_malloc((size_t)cd->naddpoints * sizeof(int));
  get_nc_int(ncid, "commpartner", cd->commpartner);
  get_nc_int(ncid, "sendcount", cd->sendcount);
  get_nc_int(ncid, "recvcount", cd->recvcount);
  get_nc_int(ncid, "addpoint_owner", cd->addpoint_owner);
  get_nc_int(ncid, "addpoint_idx", cd->addpoint_id);
}

void cfdp_free_comm_data(comm_data *cd) {
  if (!cd) return;
  if (cd->sendindex)
    for (int k = 0; k < cd->ndomains; k++) free(cd->sendindex[k]);
  if (cd->recvindex)
    for (int k = 0; k < cd->ndomains; k++) free(cd->recvindex[k]);
  free(cd->sendindex); free(cd->recvindex);
  free(cd->commpartner); free(cd->sendcount); free(cd->recvcount);
  free(cd->addpoint_owner); free(cd->addpoint_id);
  memset(cd, 0, sizeof(*cd));
}

int cfdp_load_domain(const char *prefix, int domain, int lvl, solver_data *sd, comm_data *cd) {
  char fname[4096];
  snprintf(fname, sizeof fname, "%s_domain_%d_lvl_%d", prefix, domain, lvl);
  int ncid = cfdp_nc_open(fname);
  read_solver_data(ncid, sd);
  init_solver_data(sd, 25);
  memset(cd, 0, sizeof(*cd));
  cd->iProc = domain;
  read_communication_data(ncid, cd);
  cd->nProc = cd->ndomains;
  cfdp_nc_close(ncid);
  return 0;
}

/* -------------------------------------------------------------------- schema writer --- */
int cfdp_write_domain_file(const char *path, const solver_data *sd, const comm_data *cd,
                           int cdf_version) {
  cfdp_ncwriter *w = cfdp_ncwriter_create(path, cdf_version);
  if (!w) return CFDP_NC_EIO;
  const int multi = cd && cd->ndomains > 1;
  int d_ncol = cfdp_ncwriter_def_dim(w, "ncolors", 1);
  int d_nf = cfdp_ncwriter_def_dim(w, "nfaces", (size_t)sd->nfaces);
  int d_nown = cfdp_ncwriter_def_dim(w, "nownpoints", (size_t)sd->nownpoints);
  int d_nall = cfdp_ncwriter_def_dim(w, "nallpoints", (size_t)sd->nallpoints);
  int d_two = cfdp_ncwriter_def_dim(w, "two", 2);
  int d_three = cfdp_ncwriter_def_dim(w, "three", 3);
  int d_ndom = cfdp_ncwriter_def_dim(w, "ndomains", (size_t)(cd ? cd->ndomains : 1));
  int d_nadd = -1, d_ncomm = -1;
  (void)d_nown;
  if (multi) {
    d_nadd = cfdp_ncwriter_def_dim(w, "naddpoints", (size_t)cd->naddpoints);
    d_ncomm = cfdp_ncwriter_def_dim(w, "ncommdomains", (size_t)cd->ncommdomains);
  }
  int dims2[2];
  dims2[0] = d_nf; dims2[1] = d_two;
  int v_fp = cfdp_ncwriter_def_var(w, "fpoint", CFDP_NC_INT, 2, dims2);
  dims2[1] = d_three;
  int v_fn = cfdp_ncwriter_def_var(w, "fnormal", CFDP_NC_DOUBLE, 2, dims2);
  int v_vol = cfdp_ncwriter_def_var(w, "pvolume", CFDP_NC_DOUBLE, 1, &d_nall);
  int v_cn = cfdp_ncwriter_def_var(w, "fcolor_npoints", CFDP_NC_INT, 1, &d_ncol);
  int v_cp = cfdp_ncwriter_def_var(w, "fcolor_points", CFDP_NC_INT, 1, &d_nall);
  int v_part = -1, v_sc = -1, v_rc = -1, v_ao = -1, v_ai = -1;
  if (multi) {
    v_part = cfdp_ncwriter_def_var(w, "commpartner", CFDP_NC_INT, 1, &d_ncomm);
    v_sc = cfdp_ncwriter_def_var(w, "sendcount", CFDP_NC_INT, 1, &d_ndom);
    v_rc = cfdp_ncwriter_def_var(w, "recvcount", CFDP_NC_INT, 1, &d_ndom);
    v_ao = cfdp_ncwriter_def_var(w, "addpoint_owner", CFDP_NC_INT, 1, &d_nadd);
    v_ai = cfdp_ncwriter_def_var(w, "addpoint_idx", CFDP_NC_INT, 1, &d_nadd);
  }
  int rc = cfdp_ncwriter_end_def(w);
  if (rc) { cfdp_ncwriter_close(w); return rc; }
  rc |= cfdp_ncwriter_put_int(w, v_fp, &sd->fpoint[0][0]);
  rc |= cfdp_ncwriter_put_double(w, v_fn, &sd->fnormal[0][0]);
  rc |= cfdp_ncwriter_put_double(w, v_vol, sd->pvolume);
  int ncp = sd->nallpoints;
  rc |= cfdp_ncwriter_put_int(w, v_cn, &ncp);
  int *ident = cfdp_malloc((size_t)sd->nallpoints * sizeof(int));
  for (int i = 0; i < sd->nallpoints; i++) ident[i] = i;
  rc |= cfdp_ncwriter_put_int(w, v_cp, ident);
  free(ident);
  if (multi) {
    rc |= cfdp_ncwriter_put_int(w, v_part, cd->commpartner);
    rc |= cfdp_ncwriter_put_int(w, v_sc, cd->sendcount);
    rc |= cfdp_ncwriter_put_int(w, v_rc, cd->recvcount);
    rc |= cfdp_ncwriter_put_int(w, v_ao, cd->addpoint_owner);
    rc |= cfdp_ncwriter_put_int(w, v_ai, cd->addpoint_id);
  }
  int rc2 = cfdp_ncwriter_close(w);
  return rc ? rc : rc2;
}
