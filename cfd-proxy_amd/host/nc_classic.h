/*
 * nc_classic.h -- minimal NetCDF "classic" (CDF-1 / CDF-2, big-endian) reader + writer.
 *
 * The reference links libnetcdf 3.6.3 (src/Makefile:2) and uses it through exactly
 * eight calls (src/read_netcdf.c:25,28,38,41,53,56; src/hybrid.f6.c:65,91).  There is
 * no libnetcdf in this image and only /root/repo travels to the GPU box, so the
 * dualgrid loader is backed by this file instead.  Format follows the published
 * "NetCDF Classic Format Specification" (header: magic, numrecs, dim_list, gatt_list,
 * var_list; fixed-size variables stored contiguously at `begin`, big-endian).
 *
 * Supported: CDF-1 and CDF-2 (64-bit offsets), fixed-size and record variables,
 * all six classic types; attributes are parsed and skipped.
 */
#ifndef CFDP_NC_CLASSIC_H
#define CFDP_NC_CLASSIC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { CFDP_NC_BYTE = 1, CFDP_NC_CHAR = 2, CFDP_NC_SHORT = 3, CFDP_NC_INT = 4,
       CFDP_NC_FLOAT = 5, CFDP_NC_DOUBLE = 6 };

typedef struct cfdp_ncfile cfdp_ncfile;

/* ---- reading ---- all return 0 on success, a negative CFDP_NC_E* code on failure */
enum { CFDP_NC_OK = 0, CFDP_NC_EIO = -1, CFDP_NC_EFORMAT = -2, CFDP_NC_ENOTFOUND = -3,
       CFDP_NC_ETYPE = -4, CFDP_NC_ENOMEM = -5, CFDP_NC_ECDF5 = -6, CFDP_NC_EHDF5 = -7 };

int  cfdp_ncfile_open(const char *path, cfdp_ncfile **out);
void cfdp_ncfile_close(cfdp_ncfile *f);
const char *cfdp_nc_strerror(int code);
int  cfdp_ncfile_dimlen(const cfdp_ncfile *f, const char *name, size_t *len);
int  cfdp_ncfile_varinfo(const cfdp_ncfile *f, const char *name, int *type, size_t *nelems);
/* ids = positions in the header's dimension / variable lists; negative = CFDP_NC_ENOTFOUND */
int  cfdp_ncfile_dimid(const cfdp_ncfile *f, const char *name);
int  cfdp_ncfile_varid(const cfdp_ncfile *f, const char *name);
const char *cfdp_ncfile_dimname(const cfdp_ncfile *f, int dimid);
const char *cfdp_ncfile_varname(const cfdp_ncfile *f, int varid);
/* read a whole variable, converting to the requested C type (int32 / double) */
int  cfdp_ncfile_get_int(cfdp_ncfile *f, const char *name, int *out);
int  cfdp_ncfile_get_double(cfdp_ncfile *f, const char *name, double *out);

/* ---- writing (used by the dualgrid generator and by tests) ---- */
typedef struct cfdp_ncwriter cfdp_ncwriter;
cfdp_ncwriter *cfdp_ncwriter_create(const char *path, int cdf_version /*1 or 2*/);
int  cfdp_ncwriter_def_dim(cfdp_ncwriter *w, const char *name, size_t len); /* -> dimid */
int  cfdp_ncwriter_def_var(cfdp_ncwriter *w, const char *name, int type, int ndims,
                           const int *dimids);                              /* -> varid */
int  cfdp_ncwriter_end_def(cfdp_ncwriter *w);
int  cfdp_ncwriter_put_int(cfdp_ncwriter *w, int varid, const int *data);
int  cfdp_ncwriter_put_double(cfdp_ncwriter *w, int varid, const double *data);
int  cfdp_ncwriter_close(cfdp_ncwriter *w);

#ifdef __cplusplus
}
#endif
#endif
