/*
 * read_netcdf_dropin.c -- the loader side of the drop-in boundary, backed by nc_classic.c
 * instead of libnetcdf:
 *   get_nc_int / get_nc_double / get_nc_val   reference src/read_netcdf.c:20-60, src/read_netcdf.h:4-6
 *   nc_open / nc_close / nc_strerror          what the reference's main() calls, src/hybrid.f6.c:65,91
 *                                             and ERR(), src/error_handling.h:4-10
 *   nc_inq_dimid / nc_inq_dimlen / nc_inq_varid / nc_get_var_int / nc_get_var_double
 *                                             what the reference's read_netcdf.c calls (:25,28,38,41,53,56)
 * with libnetcdf's signatures and its convention (0 = NC_NOERR, else a code nc_strerror explains),
 * so the reference's hybrid.f6.c AND read_netcdf.c link against this library unchanged.
 * cfdp_nc_open / cfdp_nc_close are the same with the reference's ERR() behaviour built in
 * (message + exit(2)).
 */
#include "nc_classic.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>

int cfdp_nc_open(const char *path);
void cfdp_nc_close(int ncid);
int nc_open(const char *path, int mode, int *ncidp);
int nc_close(int ncid);
const char *nc_strerror(int ncerr);
int nc_inq_dimid(int ncid, const char *name, int *idp);
int nc_inq_dimlen(int ncid, int dimid, size_t *lenp);
int nc_inq_varid(int ncid, const char *name, int *varidp);
int nc_get_var_int(int ncid, int varid, int *ip);
int nc_get_var_double(int ncid, int varid, double *ip);
void get_nc_double(int ncid, const char *name, double *array);
void get_nc_int(int ncid, const char *name, int *array);
int get_nc_val(int ncid, const char *name);

/* ------------------------------------------------------------------ ncid handle table */
/* (a pthread mutex, not `omp critical`: a process may hold two OpenMP runtimes -- the host's and the one this library
 * was linked with -- and a named critical section is a word both would interpret their own way) */
#define MAX_OPEN 256
static cfdp_ncfile *g_open[MAX_OPEN];
static pthread_mutex_t g_open_mtx = PTHREAD_MUTEX_INITIALIZER;

#define CFDP_NC_EBADID (-33)  /* libnetcdf's NC_EBADID */
#define CFDP_NC_ENFILE (-34)  /* libnetcdf's NC_ENFILE */

/* libnetcdf signature (src/hybrid.f6.c:65: nc_open(fname, NC_NOWRITE, &ncid)); read-only */
int nc_open(const char *path, int mode, int *ncidp) {
  (void)mode;
  cfdp_ncfile *f = NULL;
  int rc = cfdp_ncfile_open(path, &f);
  if (rc) return rc;
  pthread_mutex_lock(&g_open_mtx);
  rc = -1;
  for (int i = 0; i < MAX_OPEN; i++)
    if (!g_open[i]) { g_open[i] = f; rc = i; break; }
  pthread_mutex_unlock(&g_open_mtx);
  if (rc < 0) { cfdp_ncfile_close(f); return CFDP_NC_ENFILE; }
  *ncidp = rc + 1; /* ncid > 0 */
  return 0;
}

int cfdp_nc_open(const char *path) {
  int ncid = 0;
  const int rc = nc_open(path, 0, &ncid);
  if (rc) {
    fprintf(stderr, "Error: %s: %s\n", path, nc_strerror(rc));
    exit(2); /* reference ERR(): src/error_handling.h:4-10 */
  }
  return ncid;
}

const char *nc_strerror(int ncerr) {
  if (ncerr == CFDP_NC_EBADID) return "not a valid ncid";
  if (ncerr == CFDP_NC_ENFILE) return "too many open dualgrid files";
  return cfdp_nc_strerror(ncerr);
}

static cfdp_ncfile *nc_lookup(int ncid) {
  return ncid >= 1 && ncid <= MAX_OPEN ? g_open[ncid - 1] : NULL;
}

int nc_close(int ncid) {
  cfdp_ncfile *f = NULL;
  pthread_mutex_lock(&g_open_mtx);
  f = nc_lookup(ncid);
  if (f) g_open[ncid - 1] = NULL;
  pthread_mutex_unlock(&g_open_mtx);
  if (!f) return CFDP_NC_EBADID;
  cfdp_ncfile_close(f);
  return 0;
}

int nc_inq_dimid(int ncid, const char *name, int *idp) {
  cfdp_ncfile *f = nc_lookup(ncid);
  if (!f) return CFDP_NC_EBADID;
  const int id = cfdp_ncfile_dimid(f, name);
  if (id < 0) return id;
  *idp = id;
  return 0;
}

int nc_inq_dimlen(int ncid, int dimid, size_t *lenp) {
  cfdp_ncfile *f = nc_lookup(ncid);
  if (!f) return CFDP_NC_EBADID;
  const char *name = cfdp_ncfile_dimname(f, dimid);
  return name ? cfdp_ncfile_dimlen(f, name, lenp) : CFDP_NC_ENOTFOUND;
}

int nc_inq_varid(int ncid, const char *name, int *varidp) {
  cfdp_ncfile *f = nc_lookup(ncid);
  if (!f) return CFDP_NC_EBADID;
  const int id = cfdp_ncfile_varid(f, name);
  if (id < 0) return id;
  *varidp = id;
  return 0;
}

int nc_get_var_int(int ncid, int varid, int *ip) {
  cfdp_ncfile *f = nc_lookup(ncid);
  if (!f) return CFDP_NC_EBADID;
  const char *name = cfdp_ncfile_varname(f, varid);
  return name ? cfdp_ncfile_get_int(f, name, ip) : CFDP_NC_ENOTFOUND;
}

int nc_get_var_double(int ncid, int varid, double *ip) {
  cfdp_ncfile *f = nc_lookup(ncid);
  if (!f) return CFDP_NC_EBADID;
  const char *name = cfdp_ncfile_varname(f, varid);
  return name ? cfdp_ncfile_get_double(f, name, ip) : CFDP_NC_ENOTFOUND;
}

static cfdp_ncfile *nc_handle(int ncid) {
  if (ncid < 1 || ncid > MAX_OPEN || !g_open[ncid - 1]) {
    fprintf(stderr, "Error: invalid ncid %d\n", ncid);
    exit(2);
  }
  return g_open[ncid - 1];
}

void cfdp_nc_close(int ncid) {
  (void)nc_handle(ncid); /* message + exit(2) on a bad id */
  (void)nc_close(ncid);
}

#define NC_DIE(rc, name)                                                               \
  do {                                                                                 \
    fprintf(stderr, "Error: %s ('%s')\n", cfdp_nc_strerror(rc), name);                 \
    exit(2);                                                                           \
  } while (0)

void get_nc_int(int ncid, const char *name, int *array) {
  int rc = cfdp_ncfile_get_int(nc_handle(ncid), name, array);
  if (rc) NC_DIE(rc, name);
}

void get_nc_double(int ncid, const char *name, double *array) {
  int rc = cfdp_ncfile_get_double(nc_handle(ncid), name, array);
  if (rc) NC_DIE(rc, name);
}

int get_nc_val(int ncid, const char *name) {
  size_t len = 0;
  int rc = cfdp_ncfile_dimlen(nc_handle(ncid), name, &len);
  if (rc) NC_DIE(rc, name);
  return (int)len;
}

