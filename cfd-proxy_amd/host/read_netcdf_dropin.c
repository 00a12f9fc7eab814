/*
 * read_netcdf_dropin.c -- get_nc_int / get_nc_double / get_nc_val with the reference's
 * signatures (reference src/read_netcdf.c:20-60, src/read_netcdf.h:4-6), plus
 * cfdp_nc_open / cfdp_nc_close in place of nc_open / nc_close (src/hybrid.f6.c:65,91).
 * Backed by nc_classic.c instead of libnetcdf.  Failure = message + exit(2), the
 * reference's ERR() convention (src/error_handling.h:4-10).
 *
 * This file and nc_classic.c are also what oracle/Makefile links under the COMPILED
 * reference (whose own read_netcdf.c needs libnetcdf, absent from this image), so the
 * reference's read_solver_data()/read_communication_data() run unmodified on top of it.
 */
#include "nc_classic.h"

#include <stdio.h>
#include <stdlib.h>

int cfdp_nc_open(const char *path);
void cfdp_nc_close(int ncid);
void get_nc_double(int ncid, const char *name, double *array);
void get_nc_int(int ncid, const char *name, int *array);
int get_nc_val(int ncid, const char *name);

/* ------------------------------------------------------------------ ncid handle table */
#define MAX_OPEN 256
static cfdp_ncfile *g_open[MAX_OPEN];

int cfdp_nc_open(const char *path) {
  cfdp_ncfile *f = NULL;
  int rc = cfdp_ncfile_open(path, &f);
  if (rc) {
    fprintf(stderr, "Error: %s: %s\n", path, cfdp_nc_strerror(rc));
    exit(2); /* reference ERR(): src/error_handling.h:4-10 */
  }
#pragma omp critical(cfdp_nc_table)
  {
    rc = -1;
    for (int i = 0; i < MAX_OPEN; i++)
      if (!g_open[i]) { g_open[i] = f; rc = i; break; }
  }
  if (rc < 0) { fprintf(stderr, "Error: too many open dualgrid files\n"); exit(2); }
  return rc + 1; /* ncid > 0 */
}

static cfdp_ncfile *nc_handle(int ncid) {
  if (ncid < 1 || ncid > MAX_OPEN || !g_open[ncid - 1]) {
    fprintf(stderr, "Error: invalid ncid %d\n", ncid);
    exit(2);
  }
  return g_open[ncid - 1];
}

void cfdp_nc_close(int ncid) {
  cfdp_ncfile *f = nc_handle(ncid);
  g_open[ncid - 1] = NULL;
  cfdp_ncfile_close(f);
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

