/*
 * nc_classic.c -- NetCDF classic (CDF-1/CDF-2) reader + writer.  See nc_classic.h.
 * Replaces libnetcdf for the 8 calls the reference makes (src/read_netcdf.c:25-56,
 * src/hybrid.f6.c:65,91).  I/O only: no arithmetic of the hot path lives here.
 */
#define _FILE_OFFSET_BITS 64
#include "nc_classic.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define TAG_DIM 0x0A
#define TAG_VAR 0x0B
#define TAG_ATT 0x0C
#define MAXDIMS 32

typedef struct { char *name; size_t len; } ncdim;
typedef struct {
  char *name;
  int ndims;
  int dimids[MAXDIMS];
  int type;
  uint64_t vsize;   /* as stored (padded, per record for record vars) */
  uint64_t begin;
  int is_record;
  size_t nelems;    /* total elements (records included) */
  size_t rec_elems; /* elements per record (record vars) */
} ncvar;

struct cfdp_ncfile {
  FILE *fp;
  int version;
  size_t numrecs;
  int ndims, nvars;
  ncdim *dims;
  ncvar *vars;
  uint64_t recsize; /* bytes of one record over all record variables */
};

static size_t type_size(int t) {
  switch (t) {
  case CFDP_NC_BYTE: case CFDP_NC_CHAR: return 1;
  case CFDP_NC_SHORT: return 2;
  case CFDP_NC_INT: case CFDP_NC_FLOAT: return 4;
  case CFDP_NC_DOUBLE: return 8;
  default: return 0;
  }
}

const char *cfdp_nc_strerror(int code) {
  switch (code) {
  case CFDP_NC_OK: return "no error";
  case CFDP_NC_EIO: return "I/O error";
  case CFDP_NC_EFORMAT: return "not a NetCDF classic (CDF-1/CDF-2) file or corrupt header";
  case CFDP_NC_ENOTFOUND: return "dimension or variable not found";
  case CFDP_NC_ETYPE: return "unsupported variable type for this accessor";
  case CFDP_NC_ENOMEM: return "out of memory";
  case CFDP_NC_ECDF5: return "CDF-5 (64-bit data) file: convert with `nccopy -k classic` or `nccopy -k 64-bit-offset` "
                             "(the reference's libnetcdf 3.6.3 reads CDF-1/CDF-2 only, and so does this reader)";
  case CFDP_NC_EHDF5: return "NetCDF-4/HDF5 file: convert with `nccopy -k classic` or `nccopy -k 64-bit-offset` "
                             "(the reference's libnetcdf 3.6.3 reads CDF-1/CDF-2 only, and so does this reader)";
  default: return "unknown error";
  }
}

/* ---------------------------------------------------------------- big-endian readers */
static int rd_u32(FILE *fp, uint32_t *v) {
  unsigned char b[4];
  if (fread(b, 1, 4, fp) != 4) return CFDP_NC_EIO;
  *v = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
  return 0;
}
static int rd_u64(FILE *fp, uint64_t *v) {
  uint32_t hi, lo;
  if (rd_u32(fp, &hi) || rd_u32(fp, &lo)) return CFDP_NC_EIO;
  *v = ((uint64_t)hi << 32) | lo;
  return 0;
}
static int rd_name(FILE *fp, char **out) {
  uint32_t n;
  if (rd_u32(fp, &n)) return CFDP_NC_EIO;
  if (n > (1u << 20)) return CFDP_NC_EFORMAT;
  size_t padded = (n + 3u) & ~3u;
  char *s = (char *)malloc(padded + 1);
  if (!s) return CFDP_NC_ENOMEM;
  if (padded && fread(s, 1, padded, fp) != padded) { free(s); return CFDP_NC_EIO; }
  s[n] = 0;
  *out = s;
  return 0;
}
/* attribute list: parsed only to be skipped */
static int skip_att_list(FILE *fp) {
  uint32_t tag, n;
  if (rd_u32(fp, &tag) || rd_u32(fp, &n)) return CFDP_NC_EIO;
  if (tag == 0 && n == 0) return 0;
  if (tag != TAG_ATT) return CFDP_NC_EFORMAT;
  for (uint32_t i = 0; i < n; i++) {
    char *nm = NULL;
    int rc = rd_name(fp, &nm);
    if (rc) return rc;
    free(nm);
    uint32_t type, cnt;
    if (rd_u32(fp, &type) || rd_u32(fp, &cnt)) return CFDP_NC_EIO;
    size_t ts = type_size((int)type);
    if (!ts) return CFDP_NC_EFORMAT;
    uint64_t bytes = ((uint64_t)cnt * ts + 3u) & ~(uint64_t)3u;
    if (fseeko(fp, (off_t)bytes, SEEK_CUR)) return CFDP_NC_EIO;
  }
  return 0;
}

void cfdp_ncfile_close(cfdp_ncfile *f) {
  if (!f) return;
  if (f->fp) fclose(f->fp);
  for (int i = 0; i < f->ndims; i++) free(f->dims[i].name);
  for (int i = 0; i < f->nvars; i++) free(f->vars[i].name);
  free(f->dims);
  free(f->vars);
  free(f);
}

int cfdp_ncfile_open(const char *path, cfdp_ncfile **out) {
  *out = NULL;
  FILE *fp = fopen(path, "rb");
  if (!fp) return CFDP_NC_EIO;
  cfdp_ncfile *f = (cfdp_ncfile *)calloc(1, sizeof(*f));
  if (!f) { fclose(fp); return CFDP_NC_ENOMEM; }
  f->fp = fp;
  int rc = CFDP_NC_EFORMAT;
  unsigned char magic[4];
  if (fread(magic, 1, 4, fp) != 4) { rc = CFDP_NC_EIO; goto fail; }
  /* the formats the reference's pinned libnetcdf 3.6.3 (src/Makefile:2) reads are CDF-1 and CDF-2; name the
   * newer containers instead of calling them corrupt, so the message says what to do (nccopy -k classic) */
  if (magic[0] == 'C' && magic[1] == 'D' && magic[2] == 'F' && magic[3] == 5) { rc = CFDP_NC_ECDF5; goto fail; }
  if (magic[0] == 0x89 && magic[1] == 'H' && magic[2] == 'D' && magic[3] == 'F') { rc = CFDP_NC_EHDF5; goto fail; }
  if (magic[0] != 'C' || magic[1] != 'D' || magic[2] != 'F' || (magic[3] != 1 && magic[3] != 2))
    goto fail;
  f->version = magic[3];
  uint32_t numrecs, tag, n;
  if (rd_u32(fp, &numrecs)) { rc = CFDP_NC_EIO; goto fail; }
  f->numrecs = (numrecs == 0xFFFFFFFFu) ? 0 : numrecs; /* STREAMING marker -> unknown */
  /* dim_list */
  if (rd_u32(fp, &tag) || rd_u32(fp, &n)) { rc = CFDP_NC_EIO; goto fail; }
  if (!(tag == 0 && n == 0)) {
    if (tag != TAG_DIM || n > (1u << 20)) goto fail;
    f->dims = (ncdim *)calloc(n ? n : 1, sizeof(ncdim));
    if (!f->dims) { rc = CFDP_NC_ENOMEM; goto fail; }
    for (uint32_t i = 0; i < n; i++) {
      uint32_t len;
      if ((rc = rd_name(fp, &f->dims[i].name))) goto fail;
      f->ndims = (int)i + 1;
      if (rd_u32(fp, &len)) { rc = CFDP_NC_EIO; goto fail; }
      f->dims[i].len = len;
    }
    rc = CFDP_NC_EFORMAT;
  }
  /* gatt_list */
  if ((rc = skip_att_list(fp))) goto fail;
  rc = CFDP_NC_EFORMAT;
  /* var_list */
  if (rd_u32(fp, &tag) || rd_u32(fp, &n)) { rc = CFDP_NC_EIO; goto fail; }
  if (!(tag == 0 && n == 0)) {
    if (tag != TAG_VAR || n > (1u << 20)) goto fail;
    f->vars = (ncvar *)calloc(n ? n : 1, sizeof(ncvar));
    if (!f->vars) { rc = CFDP_NC_ENOMEM; goto fail; }
    for (uint32_t i = 0; i < n; i++) {
      ncvar *v = &f->vars[i];
      if ((rc = rd_name(fp, &v->name))) goto fail;
      f->nvars = (int)i + 1;
      rc = CFDP_NC_EFORMAT;
      uint32_t nd, type, vsize;
      if (rd_u32(fp, &nd)) { rc = CFDP_NC_EIO; goto fail; }
      if (nd > MAXDIMS) goto fail;
      v->ndims = (int)nd;
      size_t ne = 1;
      for (uint32_t d = 0; d < nd; d++) {
        uint32_t id;
        if (rd_u32(fp, &id)) { rc = CFDP_NC_EIO; goto fail; }
        if ((int)id >= f->ndims) goto fail;
        v->dimids[d] = (int)id;
        if (f->dims[id].len == 0) { /* the record (unlimited) dimension */
          if (d != 0) goto fail;
          v->is_record = 1;
        } else {
          ne *= f->dims[id].len;
        }
      }
      if ((rc = skip_att_list(fp))) goto fail;
      rc = CFDP_NC_EFORMAT;
      if (rd_u32(fp, &type) || rd_u32(fp, &vsize)) { rc = CFDP_NC_EIO; goto fail; }
      if (!type_size((int)type)) goto fail;
      v->type = (int)type;
      v->vsize = vsize;
      if (f->version == 1) {
        uint32_t b;
        if (rd_u32(fp, &b)) { rc = CFDP_NC_EIO; goto fail; }
        v->begin = b;
      } else {
        if (rd_u64(fp, &v->begin)) { rc = CFDP_NC_EIO; goto fail; }
      }
      v->rec_elems = ne;
      v->nelems = v->is_record ? ne * f->numrecs : ne;
    }
  }
  /* record size: sum of padded vsize over record vars; a single record var is unpadded */
  {
    int nrec = 0;
    uint64_t sum = 0;
    for (int i = 0; i < f->nvars; i++)
      if (f->vars[i].is_record) { nrec++; sum += f->vars[i].vsize; }
    if (nrec == 1)
      for (int i = 0; i < f->nvars; i++)
        if (f->vars[i].is_record) sum = (uint64_t)f->vars[i].rec_elems * type_size(f->vars[i].type);
    f->recsize = sum;
  }
  *out = f;
  return 0;
fail:
  cfdp_ncfile_close(f);
  return rc;
}

int cfdp_ncfile_dimlen(const cfdp_ncfile *f, const char *name, size_t *len) {
  for (int i = 0; i < f->ndims; i++)
    if (strcmp(f->dims[i].name, name) == 0) {
      *len = f->dims[i].len ? f->dims[i].len : f->numrecs;
      return 0;
    }
  return CFDP_NC_ENOTFOUND;
}

/* id-based access for the nc_inq_* entry points (ids = positions in the header lists, as in libnetcdf) */
int cfdp_ncfile_dimid(const cfdp_ncfile *f, const char *name) {
  for (int i = 0; i < f->ndims; i++)
    if (strcmp(f->dims[i].name, name) == 0) return i;
  return CFDP_NC_ENOTFOUND;
}
int cfdp_ncfile_varid(const cfdp_ncfile *f, const char *name) {
  for (int i = 0; i < f->nvars; i++)
    if (strcmp(f->vars[i].name, name) == 0) return i;
  return CFDP_NC_ENOTFOUND;
}
const char *cfdp_ncfile_dimname(const cfdp_ncfile *f, int dimid) {
  return dimid >= 0 && dimid < f->ndims ? f->dims[dimid].name : NULL;
}
const char *cfdp_ncfile_varname(const cfdp_ncfile *f, int varid) {
  return varid >= 0 && varid < f->nvars ? f->vars[varid].name : NULL;
}

static const ncvar *find_var(const cfdp_ncfile *f, const char *name) {
  for (int i = 0; i < f->nvars; i++)
    if (strcmp(f->vars[i].name, name) == 0) return &f->vars[i];
  return NULL;
}

int cfdp_ncfile_varinfo(const cfdp_ncfile *f, const char *name, int *type, size_t *nelems) {
  const ncvar *v = find_var(f, name);
  if (!v) return CFDP_NC_ENOTFOUND;
  if (type) *type = v->type;
  if (nelems) *nelems = v->nelems;
  return 0;
}

/* read raw big-endian bytes of a whole variable into buf (nelems*tsize bytes) */
static int read_raw(cfdp_ncfile *f, const ncvar *v, unsigned char *buf) {
  size_t ts = type_size(v->type);
  if (!v->is_record) {
    if (fseeko(f->fp, (off_t)v->begin, SEEK_SET)) return CFDP_NC_EIO;
    size_t bytes = v->nelems * ts;
    if (bytes && fread(buf, 1, bytes, f->fp) != bytes) return CFDP_NC_EIO;
    return 0;
  }
  size_t rb = v->rec_elems * ts;
  for (size_t r = 0; r < f->numrecs; r++) {
    if (fseeko(f->fp, (off_t)(v->begin + r * f->recsize), SEEK_SET)) return CFDP_NC_EIO;
    if (rb && fread(buf + r * rb, 1, rb, f->fp) != rb) return CFDP_NC_EIO;
  }
  return 0;
}

static inline uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }
static inline uint64_t bswap64(uint64_t x) { return __builtin_bswap64(x); }

/* element i of a raw big-endian buffer of type t as double */
static double elem_as_double(const unsigned char *raw, int t, size_t i) {
  switch (t) {
  case CFDP_NC_BYTE: return (double)((const signed char *)raw)[i];
  case CFDP_NC_CHAR: return (double)raw[i];
  case CFDP_NC_SHORT: {
    uint16_t u = (uint16_t)((raw[2 * i] << 8) | raw[2 * i + 1]);
    return (double)(int16_t)u;
  }
  case CFDP_NC_INT: {
    uint32_t u; memcpy(&u, raw + 4 * i, 4); u = bswap32(u);
    return (double)(int32_t)u;
  }
  case CFDP_NC_FLOAT: {
    uint32_t u; memcpy(&u, raw + 4 * i, 4); u = bswap32(u);
    float fl; memcpy(&fl, &u, 4);
    return (double)fl;
  }
  default: {
    uint64_t u; memcpy(&u, raw + 8 * i, 8); u = bswap64(u);
    double d; memcpy(&d, &u, 8);
    return d;
  }
  }
}

int cfdp_ncfile_get_int(cfdp_ncfile *f, const char *name, int *out) {
  const ncvar *v = find_var(f, name);
  if (!v) return CFDP_NC_ENOTFOUND;
  size_t ts = type_size(v->type);
  if (v->type == CFDP_NC_INT && !v->is_record) { /* fast path: read in place, swap */
    if (fseeko(f->fp, (off_t)v->begin, SEEK_SET)) return CFDP_NC_EIO;
    if (v->nelems && fread(out, 4, v->nelems, f->fp) != v->nelems) return CFDP_NC_EIO;
    uint32_t *u = (uint32_t *)out;
    for (size_t i = 0; i < v->nelems; i++) u[i] = bswap32(u[i]);
    return 0;
  }
  unsigned char *raw = (unsigned char *)malloc(v->nelems * ts + 8);
  if (!raw) return CFDP_NC_ENOMEM;
  int rc = read_raw(f, v, raw);
  if (!rc)
    for (size_t i = 0; i < v->nelems; i++) out[i] = (int)elem_as_double(raw, v->type, i);
  free(raw);
  return rc;
}

int cfdp_ncfile_get_double(cfdp_ncfile *f, const char *name, double *out) {
  const ncvar *v = find_var(f, name);
  if (!v) return CFDP_NC_ENOTFOUND;
  size_t ts = type_size(v->type);
  if (v->type == CFDP_NC_DOUBLE && !v->is_record) {
    if (fseeko(f->fp, (off_t)v->begin, SEEK_SET)) return CFDP_NC_EIO;
    if (v->nelems && fread(out, 8, v->nelems, f->fp) != v->nelems) return CFDP_NC_EIO;
    uint64_t *u = (uint64_t *)out;
    for (size_t i = 0; i < v->nelems; i++) u[i] = bswap64(u[i]);
    return 0;
  }
  unsigned char *raw = (unsigned char *)malloc(v->nelems * ts + 8);
  if (!raw) return CFDP_NC_ENOMEM;
  int rc = read_raw(f, v, raw);
  if (!rc)
    for (size_t i = 0; i < v->nelems; i++) out[i] = elem_as_double(raw, v->type, i);
  free(raw);
  return rc;
}

/* ================================================================== writer ======== */
typedef struct { char *name; size_t len; } wdim;
typedef struct {
  char *name;
  int type, ndims;
  int dimids[MAXDIMS];
  uint64_t nelems, vsize, begin;
} wvar;

struct cfdp_ncwriter {
  FILE *fp;
  int version;
  int ndims, nvars, capd, capv;
  wdim *dims;
  wvar *vars;
  int defined;
};

static void wr_u32(FILE *fp, uint32_t v) {
  unsigned char b[4] = {(unsigned char)(v >> 24), (unsigned char)(v >> 16),
                        (unsigned char)(v >> 8), (unsigned char)v};
  fwrite(b, 1, 4, fp);
}
static void wr_u64(FILE *fp, uint64_t v) {
  wr_u32(fp, (uint32_t)(v >> 32));
  wr_u32(fp, (uint32_t)v);
}
static void wr_name(FILE *fp, const char *s) {
  size_t n = strlen(s), padded = (n + 3u) & ~(size_t)3u;
  static const char zero[4] = {0, 0, 0, 0};
  wr_u32(fp, (uint32_t)n);
  fwrite(s, 1, n, fp);
  fwrite(zero, 1, padded - n, fp);
}
static size_t name_bytes(const char *s) { return 4 + ((strlen(s) + 3u) & ~(size_t)3u); }

cfdp_ncwriter *cfdp_ncwriter_create(const char *path, int cdf_version) {
  if (cdf_version != 1 && cdf_version != 2) return NULL;
  FILE *fp = fopen(path, "wb");
  if (!fp) return NULL;
  cfdp_ncwriter *w = (cfdp_ncwriter *)calloc(1, sizeof(*w));
  if (!w) { fclose(fp); return NULL; }
  w->fp = fp;
  w->version = cdf_version;
  return w;
}

int cfdp_ncwriter_def_dim(cfdp_ncwriter *w, const char *name, size_t len) {
  if (w->defined || len == 0) return CFDP_NC_EFORMAT; /* no record dims on write */
  if (w->ndims == w->capd) {
    w->capd = w->capd ? 2 * w->capd : 16;
    w->dims = (wdim *)realloc(w->dims, (size_t)w->capd * sizeof(wdim));
  }
  w->dims[w->ndims].name = strdup(name);
  w->dims[w->ndims].len = len;
  return w->ndims++;
}

int cfdp_ncwriter_def_var(cfdp_ncwriter *w, const char *name, int type, int ndims,
                          const int *dimids) {
  if (w->defined || !type_size(type) || ndims > MAXDIMS) return CFDP_NC_EFORMAT;
  if (w->nvars == w->capv) {
    w->capv = w->capv ? 2 * w->capv : 16;
    w->vars = (wvar *)realloc(w->vars, (size_t)w->capv * sizeof(wvar));
  }
  wvar *v = &w->vars[w->nvars];
  memset(v, 0, sizeof(*v));
  v->name = strdup(name);
  v->type = type;
  v->ndims = ndims;
  v->nelems = 1;
  for (int d = 0; d < ndims; d++) {
    if (dimids[d] < 0 || dimids[d] >= w->ndims) return CFDP_NC_EFORMAT;
    v->dimids[d] = dimids[d];
    v->nelems *= w->dims[dimids[d]].len;
  }
  v->vsize = (v->nelems * type_size(type) + 3u) & ~(uint64_t)3u;
  return w->nvars++;
}

int cfdp_ncwriter_end_def(cfdp_ncwriter *w) {
  if (w->defined) return CFDP_NC_EFORMAT;
  /* header size */
  uint64_t hs = 4 + 4 + 8 + 8 + 8;
  for (int i = 0; i < w->ndims; i++) hs += name_bytes(w->dims[i].name) + 4;
  for (int i = 0; i < w->nvars; i++)
    hs += name_bytes(w->vars[i].name) + 4 + 4u * (unsigned)w->vars[i].ndims + 8 + 4 + 4 +
          (w->version == 1 ? 4 : 8);
  uint64_t off = hs;
  for (int i = 0; i < w->nvars; i++) {
    w->vars[i].begin = off;
    off += w->vars[i].vsize;
  }
  if (w->version == 1 && off > 0x7FFFFFFFull) return CFDP_NC_EFORMAT; /* needs CDF-2 */
  FILE *fp = w->fp;
  fputc('C', fp); fputc('D', fp); fputc('F', fp); fputc(w->version, fp);
  wr_u32(fp, 0); /* numrecs */
  if (w->ndims) { wr_u32(fp, TAG_DIM); wr_u32(fp, (uint32_t)w->ndims); }
  else { wr_u32(fp, 0); wr_u32(fp, 0); }
  for (int i = 0; i < w->ndims; i++) {
    wr_name(fp, w->dims[i].name);
    wr_u32(fp, (uint32_t)w->dims[i].len);
  }
  wr_u32(fp, 0); wr_u32(fp, 0); /* no global attributes */
  if (w->nvars) { wr_u32(fp, TAG_VAR); wr_u32(fp, (uint32_t)w->nvars); }
  else { wr_u32(fp, 0); wr_u32(fp, 0); }
  for (int i = 0; i < w->nvars; i++) {
    wvar *v = &w->vars[i];
    wr_name(fp, v->name);
    wr_u32(fp, (uint32_t)v->ndims);
    for (int d = 0; d < v->ndims; d++) wr_u32(fp, (uint32_t)v->dimids[d]);
    wr_u32(fp, 0); wr_u32(fp, 0); /* no attributes */
    wr_u32(fp, (uint32_t)v->type);
    wr_u32(fp, v->vsize > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v->vsize);
    if (w->version == 1) wr_u32(fp, (uint32_t)v->begin);
    else wr_u64(fp, v->begin);
  }
  if ((uint64_t)ftello(fp) != hs) return CFDP_NC_EFORMAT;
  w->defined = 1;
  return 0;
}

static int put_swapped(cfdp_ncwriter *w, int varid, const void *data, size_t esz, int type) {
  if (!w->defined || varid < 0 || varid >= w->nvars) return CFDP_NC_EFORMAT;
  wvar *v = &w->vars[varid];
  if (v->type != type) return CFDP_NC_ETYPE;
  if (fseeko(w->fp, (off_t)v->begin, SEEK_SET)) return CFDP_NC_EIO;
  enum { CH = 1 << 16 };
  unsigned char *buf = (unsigned char *)malloc((size_t)CH * esz);
  if (!buf) return CFDP_NC_ENOMEM;
  uint64_t done = 0;
  while (done < v->nelems) {
    size_t n = (size_t)((v->nelems - done) < CH ? (v->nelems - done) : CH);
    if (esz == 4) {
      const uint32_t *s = (const uint32_t *)data + done;
      uint32_t *d = (uint32_t *)buf;
      for (size_t i = 0; i < n; i++) d[i] = bswap32(s[i]);
    } else {
      const uint64_t *s = (const uint64_t *)data + done;
      uint64_t *d = (uint64_t *)buf;
      for (size_t i = 0; i < n; i++) d[i] = bswap64(s[i]);
    }
    if (fwrite(buf, esz, n, w->fp) != n) { free(buf); return CFDP_NC_EIO; }
    done += n;
  }
  free(buf);
  return 0;
}

int cfdp_ncwriter_put_int(cfdp_ncwriter *w, int varid, const int *data) {
  return put_swapped(w, varid, data, 4, CFDP_NC_INT);
}
int cfdp_ncwriter_put_double(cfdp_ncwriter *w, int varid, const double *data) {
  return put_swapped(w, varid, data, 8, CFDP_NC_DOUBLE);
}

int cfdp_ncwriter_close(cfdp_ncwriter *w) {
  if (!w) return 0;
  int rc = 0;
  if (w->fp) {
    /* make sure the file extends to the end of the last variable (padding included) */
    if (w->defined && w->nvars) {
      wvar *last = &w->vars[w->nvars - 1];
      uint64_t end = last->begin + last->vsize;
      fseeko(w->fp, 0, SEEK_END);
      uint64_t cur = (uint64_t)ftello(w->fp);
      while (cur < end) { fputc(0, w->fp); cur++; }
    }
    if (fclose(w->fp)) rc = CFDP_NC_EIO;
  }
  for (int i = 0; i < w->ndims; i++) free(w->dims[i].name);
  for (int i = 0; i < w->nvars; i++) free(w->vars[i].name);
  free(w->dims);
  free(w->vars);
  free(w);
  return rc;
}
