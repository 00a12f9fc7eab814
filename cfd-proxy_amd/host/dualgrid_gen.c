#define _GNU_SOURCE
/*
 * dualgrid_gen.c -- deterministic F6-like dualgrid generator.
 *
 * The reference reads pre-partitioned DLR-F6 dual meshes "f6/dualgrid.N" which are NOT
 * part of the checkout (/root/reference/.MISSING_LARGE_BLOBS:1-6).  This generator
 * produces stand-ins with the same schema (reference src/solver_data.c:98-144,
 * src/comm_data.c:79-112): a lattice with Kuhn/Freudenthal connectivity (7 edge
 * directions, ~6.8 faces per point like an unstructured dual grid), partitioned by
 * recursive coordinate bisection into `ndomains` boxes, one file per domain, with ghost
 * points ("addpoints") identified by (owner domain, owner-local id).
 *
 * Everything (normals, volumes) is a pure function of (seed, global id, direction), so a
 * face stored in two domain files is bit-identical in both.
 *
 * connectivity = CFDP_CONN_IRREGULAR: the same schema on an IRREGULAR graph -- what an unstructured dual grid looks like
 * to the face loops, which see point numbers, normals and volumes, never coordinates: the edge graph of a RANDOM
 * tetrahedralisation of the lattice's cubes (face_exists) with a hub point every 1024 -- 14 incidences per point on
 * average like the Kuhn lattice (13.6), but 8 to 24 from point to point and 60+ at the hubs, no two tiles alike.
 * numbering = 1 scrambles the file numbering of every domain's points (a real mesh file is not numbered along x).
 */
#include "cfdproxy_host.h"
#include "host_util.h"

#include <math.h>
#include <string.h>
#include <sched.h>
#include <time.h>

double cfdp_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

const char *cfdp_host_version(void) { return "cfdproxy-mi355x host 0.1"; }

/* Threads the library's own OpenMP regions use: the cores this process may really run on.  A container (and a GPU box
 * that grants a share of a bigger host) shows every CPU of the host but grants a quota: an OpenMP runtime left to its
 * default starts one thread per visible CPU, and the host stages of the plan then run SLOWER than on one thread
 * (measured here: 8 granted of many visible CPUs: the point->face CSR 0.69 s by default, 0.27 s on one thread, 0.09 s
 * on 8).  min(affinity mask, cgroup CPU quota, OMP_NUM_THREADS if set); CFDP_HOST_THREADS overrides. */
int cfdp_host_threads(void) {
  static int cached = 0;
  if (cached > 0) return cached;
  int n = 0;
  const char *e = getenv("CFDP_HOST_THREADS");
  if (e && atoi(e) > 0) n = atoi(e);
  if (n <= 0) {
    cpu_set_t set;
    CPU_ZERO(&set);
    n = sched_getaffinity(0, sizeof set, &set) == 0 ? CPU_COUNT(&set) : 1;
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r"); /* cgroup v2: "<quota> <period>" or "max <period>" */
    if (f) {
      char q[32];
      long period = 0;
      if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
        const long quota = atol(q);
        if (quota > 0 && quota / period < n) n = (int)(quota / period);
      }
      fclose(f);
    } else if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"))) { /* cgroup v1 */
      long quota = -1, period = 0;
      if (fscanf(f, "%ld", &quota) != 1) quota = -1;
      fclose(f);
      if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r"))) {
        if (fscanf(f, "%ld", &period) != 1) period = 0;
        fclose(f);
      }
      if (quota > 0 && period > 0 && quota / period < n) n = (int)(quota / period);
    }
    const char *o = getenv("OMP_NUM_THREADS");
    if (o && atoi(o) > 0 && atoi(o) < n) n = atoi(o);
    if (n > 64) n = 64;
    if (n < 1) n = 1;
  }
  cached = n;
  return n;
}

static const int DIRS[7][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0},
                               {0, 1, 1}, {1, 0, 1}, {1, 1, 1}};

/* the candidate directions of a generated mesh: the 7 (3) lattice directions, or -- irregular -- the 62 lexicographically
 * positive offsets of the 5x5x5 neighbourhood, the 13 with no component beyond +-1 first */
typedef struct { int n, reach, irregular; int d[62][3]; } dirset;
static void make_dirs(const cfdp_gen_params *gp, dirset *D) {
  D->irregular = gp->connectivity == CFDP_CONN_IRREGULAR;
  if (!D->irregular) {
    D->n = gp->connectivity;
    D->reach = 1;
    memcpy(D->d, DIRS, sizeof DIRS);
    return;
  }
  D->n = 0;
  D->reach = 2;
  for (int far = 0; far < 2; far++)
    for (int dz = -2; dz <= 2; dz++)
      for (int dy = -2; dy <= 2; dy++)
        for (int dx = -2; dx <= 2; dx++) {
          if (!(dz > 0 || (dz == 0 && (dy > 0 || (dy == 0 && dx > 0))))) continue;
          const int is_far = abs(dx) > 1 || abs(dy) > 1 || abs(dz) > 1;
          if (is_far != far) continue;
          D->d[D->n][0] = dx; D->d[D->n][1] = dy; D->d[D->n][2] = dz;
          D->n++;
        }
  CFDP_ASSERT(D->n == 62);
}
static inline int is_hub(const cfdp_gen_params *gp, long gid) {
  if (gp->hubs < 0) return 0; /* (irregular meshes without hub points: what the hubs alone cost the kernels) */
  return (cfdp_mix64(gp->seed * 0xD1B54A32D192ED03ull + 0x77ull + (uint64_t)gid) & 1023u) == 0;
}
/* does the face from point gid0 in direction d (to gid1) exist?  Lattice meshes: always.  Irregular: the lattice's unit
 * cubes are cut into tetrahedra AT RANDOM -- every cube face takes one of its two diagonals, every cube one of its four
 * body diagonals (a pure function of seed and face / cube), the axis edges always exist: the edge graph of an
 * unstructured tetrahedral mesh, 14 incidences per point on average like the Kuhn lattice but 8 to 24 from point to
 * point -- and one point in 1024 is a hub joined to its whole 3x3x3 neighbourhood and a third of the 5x5x5 one. */
static inline int face_exists(const cfdp_gen_params *gp, const dirset *D, long gid0, long gid1, int d) {
  if (!D->irregular) return 1;
  const int hub = is_hub(gp, gid0) || is_hub(gp, gid1);
  if (d >= 13) /* beyond the 3x3x3 neighbourhood: hubs only */
    return hub && cfdp_u01(gp->seed * 0xA24BAED4963EE407ull + 0x1234567ull + ((uint64_t)gid0 * 64u + (uint64_t)d)) < 0.355;
  if (hub) return 1;
  const int *v = D->d[d];
  const int nz = (v[0] != 0) + (v[1] != 0) + (v[2] != 0);
  if (nz == 1) return 1;
  const long x = gid0 % gp->nx, y = (gid0 / gp->nx) % gp->ny, z = gid0 / ((long)gp->nx * gp->ny);
  /* the face / cube this diagonal lies in, named by its corner with the smallest coordinates */
  const long cx = v[0] < 0 ? x - 1 : x, cy = v[1] < 0 ? y - 1 : y, cz = v[2] < 0 ? z - 1 : z;
  const uint64_t cell = (uint64_t)((cz * gp->ny + cy) * gp->nx + cx);
  if (nz == 2) {
    const int plane = v[0] == 0 ? 0 : (v[1] == 0 ? 1 : 2);
    const int a = plane == 0 ? v[1] : v[0], b = plane == 2 ? v[1] : v[2]; /* the two non-zero components */
    const int main_diag = (a > 0) == (b > 0);
    const int pick = (int)(cfdp_mix64(gp->seed * 0x8CB92BA72F3D8DD7ull + 0x1111ull + cell * 4u + (uint64_t)plane) & 1u);
    return pick == main_diag;
  }
  /* a body diagonal: which of the cube's four, by the signs of dx and dy relative to dz (> 0 always) */
  const int which = (v[0] < 0 ? 1 : 0) | (v[1] < 0 ? 2 : 0);
  return (int)(cfdp_mix64(gp->seed * 0xC2B2AE3D27D4EB4Full + 0x2222ull + cell) & 3u) == which;
}
/* file numbering of a domain's own points: lexicographic in its box, or (numbering = 1) scrambled by an affine
 * permutation i -> (a*i + b) mod n with gcd(a, n) = 1 -- computable point by point, so a ghost's owner-local id needs no
 * table of the owner's numbering */
static inline long scramble_mult(long n) {
  long a = (long)(0.6180339887 * (double)n) | 1;
  for (;; a += 2) {
    long x = a, y = n;
    while (y) { long t = x % y; x = y; y = t; }
    if (x == 1) return a;
  }
}
static inline int file_id(const cfdp_gen_params *gp, long lex, long n) {
  if (!gp->numbering || n < 3) return (int)lex;
  return (int)((scramble_mult(n) % n * lex + n / 3) % n);
}

typedef struct { int lo[3], hi[3]; } box_t; /* [lo,hi) */

/* ---- recursive coordinate bisection: domain ids in tree order (spatially coherent) ---- */
static void rcb(box_t b, int first, int count, box_t *out) {
  if (count == 1) { out[first] = b; return; }
  int ax = 0, len = b.hi[0] - b.lo[0];
  for (int a = 1; a < 3; a++)
    if (b.hi[a] - b.lo[a] > len) { len = b.hi[a] - b.lo[a]; ax = a; }
  int c1 = count / 2;
  int cut = b.lo[ax] + (int)(((long)len * c1 + count / 2) / count);
  if (cut <= b.lo[ax]) cut = b.lo[ax] + 1;
  if (cut >= b.hi[ax]) cut = b.hi[ax] - 1;
  CFDP_ASSERT(cut > b.lo[ax] && cut < b.hi[ax]); /* more domains than lattice planes */
  box_t l = b, r = b;
  l.hi[ax] = cut;
  r.lo[ax] = cut;
  rcb(l, first, c1, out);
  rcb(r, first + c1, count - c1, out);
}

static box_t *make_boxes(const cfdp_gen_params *gp) {
  CFDP_ASSERT(gp->ndomains >= 1);
  box_t *boxes = cfdp_malloc((size_t)gp->ndomains * sizeof(box_t));
  box_t all = {{0, 0, 0}, {gp->nx, gp->ny, gp->nz}};
  rcb(all, 0, gp->ndomains, boxes);
  return boxes;
}

static inline int in_box(const box_t *b, int x, int y, int z) {
  return x >= b->lo[0] && x < b->hi[0] && y >= b->lo[1] && y < b->hi[1] && z >= b->lo[2] &&
         z < b->hi[2];
}
static inline int box_local(const box_t *b, int x, int y, int z) {
  int bx = b->hi[0] - b->lo[0], by = b->hi[1] - b->lo[1];
  return ((z - b->lo[2]) * by + (y - b->lo[1])) * bx + (x - b->lo[0]);
}
/* owner of a lattice point: linear scan over boxes is fine for <= a few hundred domains
 * when only boundary-layer points are queried; a coarse cache keeps it cheap.           */
static int owner_of(const box_t *boxes, int nd, int x, int y, int z, int hint) {
  if (hint >= 0 && in_box(&boxes[hint], x, y, z)) return hint;
  for (int d = 0; d < nd; d++)
    if (in_box(&boxes[d], x, y, z)) return d;
  return -1;
}

static inline long gid_of(const cfdp_gen_params *gp, int x, int y, int z) {
  return ((long)z * gp->ny + y) * gp->nx + x;
}

static void face_normal(const cfdp_gen_params *gp, long gid0, int d, double h, double *n) {
  if (gp->normals == 0) {
    for (int c = 0; c < 3; c++) n[c] = h * h * DIRS[d][c];
    return;
  }
  uint64_t key = gp->seed * 0x100000001B3ull + ((uint64_t)gid0 * (gp->connectivity == CFDP_CONN_IRREGULAR ? 64u : 8u) + (uint64_t)d) * 4u;
  const double twopi = 6.283185307179586476925286766559;
  double u1 = cfdp_u01(key + 0), u2 = cfdp_u01(key + 1);
  double u3 = cfdp_u01(key + 2), u4 = cfdp_u01(key + 3);
  double r1 = sqrt(-2.0 * log(u1)), r2 = sqrt(-2.0 * log(u3));
  n[0] = h * h * r1 * cos(twopi * u2);
  n[1] = h * h * r1 * sin(twopi * u2);
  n[2] = h * h * r2 * cos(twopi * u4);
}

static double point_volume(const cfdp_gen_params *gp, long gid, double h) {
  if (gp->volumes == 0) return h * h * h;
  double u = cfdp_u01(gp->seed * 0x9E3779B1ull + 0x5151515151ull + (uint64_t)gid * 3u);
  return (0.5 + 1.5 * u) * h * h * h;
}

int cfdp_gen_domain(const cfdp_gen_params *gp, int domain, solver_data *sd, comm_data *cd) {
  CFDP_ASSERT(gp->connectivity == 7 || gp->connectivity == 3 || gp->connectivity == CFDP_CONN_IRREGULAR);
  CFDP_ASSERT(domain >= 0 && domain < gp->ndomains);
  dirset D;
  make_dirs(gp, &D);
  const int ndir = D.n;
  const double h = 1.0 / (double)gp->nx;
  box_t *boxes = make_boxes(gp);
  const box_t B = boxes[domain];
  /* expanded box (as many layers as the directions reach), clipped to the lattice */
  box_t E = B;
  const int dim[3] = {gp->nx, gp->ny, gp->nz};
  for (int a = 0; a < 3; a++) {
    E.lo[a] = E.lo[a] - D.reach > 0 ? E.lo[a] - D.reach : 0;
    E.hi[a] = E.hi[a] + D.reach < dim[a] ? E.hi[a] + D.reach : dim[a];
  }
  const int ex = E.hi[0] - E.lo[0], ey = E.hi[1] - E.lo[1], ez = E.hi[2] - E.lo[2];
  const size_t ne = (size_t)ex * ey * ez;
  int *slot = cfdp_malloc(ne * sizeof(int)); /* local id of each expanded-box point or -1 */
#define EIDX(x, y, z) ((((size_t)(z)-E.lo[2]) * ey + ((y)-E.lo[1])) * ex + ((x)-E.lo[0]))
#define IN_LATTICE(x, y, z) ((x) >= 0 && (y) >= 0 && (z) >= 0 && (x) < dim[0] && (y) < dim[1] && (z) < dim[2])
  const int nown = (B.hi[0] - B.lo[0]) * (B.hi[1] - B.lo[1]) * (B.hi[2] - B.lo[2]);

  /* pass 1: own points, and ghosts = outside points joined by a face (any direction, either
   * orientation) to an own point; ghosts numbered in ascending global id               */
  int nadd = 0;
  for (int z = E.lo[2]; z < E.hi[2]; z++)
    for (int y = E.lo[1]; y < E.hi[1]; y++)
      for (int x = E.lo[0]; x < E.hi[0]; x++) {
        int s = -1;
        if (in_box(&B, x, y, z)) {
          s = file_id(gp, box_local(&B, x, y, z), nown);
        } else if (gp->ndomains > 1) {
          int adj = 0;
          const long g = gid_of(gp, x, y, z);
          for (int d = 0; d < ndir && !adj; d++)
            for (int sg = -1; sg <= 1 && !adj; sg += 2) {
              const int x1 = x + sg * D.d[d][0], y1 = y + sg * D.d[d][1], z1 = z + sg * D.d[d][2];
              if (!in_box(&B, x1, y1, z1)) continue;
              const long g1 = gid_of(gp, x1, y1, z1);
              adj = sg > 0 ? face_exists(gp, &D, g, g1, d) : face_exists(gp, &D, g1, g, d);
            }
          if (adj) s = nown + nadd++;
        }
        slot[EIDX(x, y, z)] = s;
      }
  const int nall = nown + nadd;

  /* pass 2: count faces, then fill */
  size_t nf = 0;
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) {
      memset(sd, 0, sizeof(*sd));
      sd->nfaces = (int)nf;
      sd->nallfaces = (int)nf;
      sd->nownpoints = nown;
      sd->nallpoints = nall;
      sd->ncolors = 1;
      sd->fpoint = cfdp_malloc(nf * 2 * sizeof(int));
      sd->fnormal = cfdp_malloc(nf * 3 * sizeof(double));
      nf = 0;
    }
    for (int z = E.lo[2]; z < E.hi[2]; z++)
      for (int y = E.lo[1]; y < E.hi[1]; y++)
        for (int x = E.lo[0]; x < E.hi[0]; x++) {
          int s0 = slot[EIDX(x, y, z)];
          if (s0 < 0) continue;
          const long g0 = gid_of(gp, x, y, z);
          for (int d = 0; d < ndir; d++) {
            int x1 = x + D.d[d][0], y1 = y + D.d[d][1], z1 = z + D.d[d][2];
            if (x1 < E.lo[0] || y1 < E.lo[1] || x1 >= E.hi[0] || y1 >= E.hi[1] || z1 >= E.hi[2]) continue;
            int s1 = slot[EIDX(x1, y1, z1)];
            if (s1 < 0) continue;
            if (s0 >= nown && s1 >= nown && !gp->ghost_faces) continue;
            if (!face_exists(gp, &D, g0, gid_of(gp, x1, y1, z1), d)) continue;
            if (pass == 1) {
              sd->fpoint[nf][0] = s0;
              sd->fpoint[nf][1] = s1;
              face_normal(gp, g0, d, h, sd->fnormal[nf]);
            }
            nf++;
          }
        }
  }
  CFDP_ASSERT(nf > 0);

  sd->pvolume = cfdp_malloc((size_t)nall * sizeof(double));
  sd->var = cfdp_malloc((size_t)nall * NGRAD * sizeof(double));
  sd->grad = cfdp_malloc((size_t)nall * NGRAD * 3 * sizeof(double));
  sd->psd_flux = cfdp_malloc((size_t)nall * NFLUX * sizeof(double));
  memset(cd, 0, sizeof(*cd));
  cd->nProc = gp->ndomains;
  cd->iProc = domain;
  cd->ndomains = gp->ndomains;
  cd->nownpoints = nown;
  cd->naddpoints = nadd;
  if (nadd) {
    cd->addpoint_owner = cfdp_malloc((size_t)nadd * sizeof(int));
    cd->addpoint_id = cfdp_malloc((size_t)nadd * sizeof(int));
    cd->sendcount = cfdp_calloc((size_t)gp->ndomains, sizeof(int));
    cd->recvcount = cfdp_calloc((size_t)gp->ndomains, sizeof(int));
  }
  int hint = -1;
  for (int z = E.lo[2]; z < E.hi[2]; z++)
    for (int y = E.lo[1]; y < E.hi[1]; y++)
      for (int x = E.lo[0]; x < E.hi[0]; x++) {
        int s = slot[EIDX(x, y, z)];
        if (s < 0) continue;
        sd->pvolume[s] = point_volume(gp, gid_of(gp, x, y, z), h);
        if (s >= nown) {
          int k = owner_of(boxes, gp->ndomains, x, y, z, hint);
          CFDP_ASSERT(k >= 0 && k != domain);
          hint = k;
          const box_t *K = &boxes[k];
          cd->addpoint_owner[s - nown] = k;
          cd->addpoint_id[s - nown] = file_id(gp, box_local(K, x, y, z),
                                              (long)(K->hi[0] - K->lo[0]) * (K->hi[1] - K->lo[1]) * (K->hi[2] - K->lo[2]));
          cd->recvcount[k]++;
        }
      }
  /* sendcount[k]: own points joined by a face to a point owned by k */
  if (nadd) {
    for (int z = B.lo[2]; z < B.hi[2]; z++)
      for (int y = B.lo[1]; y < B.hi[1]; y++)
        for (int x = B.lo[0]; x < B.hi[0]; x++) {
          int interior = x >= B.lo[0] + D.reach && x < B.hi[0] - D.reach && y >= B.lo[1] + D.reach && y < B.hi[1] - D.reach &&
                         z >= B.lo[2] + D.reach && z < B.hi[2] - D.reach;
          if (interior) continue;
          const long g = gid_of(gp, x, y, z);
          int seen[124], ns = 0;
          for (int d = 0; d < ndir; d++)
            for (int sg = -1; sg <= 1; sg += 2) {
              int x1 = x + sg * D.d[d][0], y1 = y + sg * D.d[d][1], z1 = z + sg * D.d[d][2];
              if (!IN_LATTICE(x1, y1, z1)) continue;
              if (in_box(&B, x1, y1, z1)) continue;
              const long g1 = gid_of(gp, x1, y1, z1);
              if (!(sg > 0 ? face_exists(gp, &D, g, g1, d) : face_exists(gp, &D, g1, g, d))) continue;
              int k = owner_of(boxes, gp->ndomains, x1, y1, z1, hint);
              hint = k;
              int dup = 0;
              for (int i = 0; i < ns; i++) dup |= (seen[i] == k);
              if (!dup) { seen[ns++] = k; cd->sendcount[k]++; }
            }
        }
    int nc = 0;
    for (int k = 0; k < gp->ndomains; k++)
      if (cd->sendcount[k] > 0 || cd->recvcount[k] > 0) nc++;
    cd->ncommdomains = nc;
    cd->commpartner = cfdp_malloc((size_t)nc * sizeof(int));
    nc = 0;
    for (int k = 0; k < gp->ndomains; k++)
      if (cd->sendcount[k] > 0 || cd->recvcount[k] > 0) cd->commpartner[nc++] = k;
  }
  init_solver_data(sd, 25);
  free(slot);
  free(boxes);
#undef EIDX
  return 0;
}

int cfdp_gen_global_ids(const cfdp_gen_params *gp, int domain, int *gid) {
  dirset D;
  make_dirs(gp, &D);
  const int ndir = D.n;
  box_t *boxes = make_boxes(gp);
  const box_t B = boxes[domain];
  box_t E = B;
  const int dim[3] = {gp->nx, gp->ny, gp->nz};
  for (int a = 0; a < 3; a++) {
    E.lo[a] = E.lo[a] - D.reach > 0 ? E.lo[a] - D.reach : 0;
    E.hi[a] = E.hi[a] + D.reach < dim[a] ? E.hi[a] + D.reach : dim[a];
  }
  const int nown = (B.hi[0] - B.lo[0]) * (B.hi[1] - B.lo[1]) * (B.hi[2] - B.lo[2]);
  int nadd = 0;
  for (int z = E.lo[2]; z < E.hi[2]; z++)
    for (int y = E.lo[1]; y < E.hi[1]; y++)
      for (int x = E.lo[0]; x < E.hi[0]; x++) {
        if (in_box(&B, x, y, z)) {
          gid[file_id(gp, box_local(&B, x, y, z), nown)] = (int)gid_of(gp, x, y, z);
        } else if (gp->ndomains > 1) {
          int adj = 0;
          const long g = gid_of(gp, x, y, z);
          for (int d = 0; d < ndir && !adj; d++)
            for (int sg = -1; sg <= 1 && !adj; sg += 2) {
              const int x1 = x + sg * D.d[d][0], y1 = y + sg * D.d[d][1], z1 = z + sg * D.d[d][2];
              if (!in_box(&B, x1, y1, z1)) continue;
              const long g1 = gid_of(gp, x1, y1, z1);
              adj = sg > 0 ? face_exists(gp, &D, g, g1, d) : face_exists(gp, &D, g1, g, d);
            }
          if (adj) gid[nown + nadd++] = (int)gid_of(gp, x, y, z);
        }
      }
  free(boxes);
  return nown + nadd;
}

int cfdp_gen_write_domain(const cfdp_gen_params *gp, int domain, const char *prefix, int lvl) {
  solver_data sd;
  comm_data cd;
  cfdp_gen_domain(gp, domain, &sd, &cd);
  char fname[4096];
  snprintf(fname, sizeof fname, "%s_domain_%d_lvl_%d", prefix, domain, lvl);
  int rc = cfdp_write_domain_file(fname, &sd, &cd, gp->cdf_version ? gp->cdf_version : 1);
  cfdp_free_solver_data(&sd);
  cfdp_free_comm_data(&cd);
  return rc;
}

/* var fields: `one` is the reference default
 * (src/solver_data.c:26-36); `hash` is decorrelated and used for parity; `linear` is the
 * known-answer field (Green-Gauss is exact for it on the Cartesian lattice).            */
void cfdp_fill_var(double (*var)[NGRAD], const int *gid, int npoints, int kind, int nx, int ny,
                   int nz) {
  (void)nz;
  const double h = 1.0 / (double)nx;
  for (int i = 0; i < npoints; i++) {
    long g = gid ? gid[i] : i;
    for (int eq = 0; eq < NGRAD; eq++) {
      double v = 1.0;
      if (kind == CFDP_VAR_HASH) {
        v = 1.0 + 0.01 * (double)((7 * g + 13 * eq) % 101);
      } else if (kind == CFDP_VAR_LINEAR) {
        double x = h * (double)(g % nx), y = h * (double)((g / nx) % ny),
               z = h * (double)(g / ((long)nx * ny));
        v = (eq + 1.0) * x + (2.0 * eq - 3.0) * y + (0.5 * eq + 1.0) * z + (double)eq;
      }
      var[i][eq] = v;
    }
  }
}

double cfdp_algo_bytes_grad(long nfaces, long nown, long nadd) {
  return 32.0 * (double)nfaces + 232.0 * (double)nown + 56.0 * (double)nadd;
}
double cfdp_algo_bytes_flux(long nfaces, long nown, long nadd) {
  return 32.0 * (double)nfaces + 72.0 * (double)(nown + nadd) + 24.0 * (double)nown;
}
