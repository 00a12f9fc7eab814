/*
 * host_foreign_symbols.c -- TEST HOST (compiled by tests/test_host_io.py, CPU only).  An application that brings its
 * OWN functions with libnetcdf's names (it links a real libnetcdf, say) and its own now(): the library's loader
 * entry points (cfdp_nc_open, get_nc_val, read_solver_data: the reference's src/read_netcdf.c:20-60,
 * src/solver_data.c:80-160) must keep using the library's reader, not the application's functions.
 *
 *   host_foreign_symbols FILE      prints nownpoints and nfaces read through the library
 */
#include <stdio.h>
#include <stdlib.h>

#include "cfdproxy_host.h"

/* the application's own: every one of them fails loudly if the library binds to it */
int nc_open(const char *path, int mode, int *ncidp) { (void)path; (void)mode; (void)ncidp; fprintf(stderr, "foreign nc_open called\n"); exit(9); }
int nc_inq_dimid(int ncid, const char *name, int *idp) { (void)ncid; (void)name; (void)idp; fprintf(stderr, "foreign nc_inq_dimid called\n"); exit(9); }
int nc_inq_dimlen(int ncid, int dimid, size_t *lenp) { (void)ncid; (void)dimid; (void)lenp; fprintf(stderr, "foreign nc_inq_dimlen called\n"); exit(9); }
int nc_inq_varid(int ncid, const char *name, int *varidp) { (void)ncid; (void)name; (void)varidp; fprintf(stderr, "foreign nc_inq_varid called\n"); exit(9); }
int nc_get_var_int(int ncid, int varid, int *ip) { (void)ncid; (void)varid; (void)ip; fprintf(stderr, "foreign nc_get_var_int called\n"); exit(9); }
int nc_get_var_double(int ncid, int varid, double *ip) { (void)ncid; (void)varid; (void)ip; fprintf(stderr, "foreign nc_get_var_double called\n"); exit(9); }
int nc_close(int ncid) { (void)ncid; fprintf(stderr, "foreign nc_close called\n"); exit(9); }
const char *nc_strerror(int e) { (void)e; return "foreign"; }
double now(void) { return -1.0; }

int cfdp_nc_open(const char *path);
void cfdp_nc_close(int ncid);

int main(int argc, char **argv) {
  if (argc < 2) return 1;
  solver_data sd;
  const int ncid = cfdp_nc_open(argv[1]);
  const int n = get_nc_val(ncid, "nownpoints");
  read_solver_data(ncid, &sd);
  cfdp_nc_close(ncid);
  printf("nownpoints %d %d nfaces %d\n", n, sd.nownpoints, sd.nfaces);
  return 0;
}
