/*
 * cfdproxy_host.h -- host-side (plain C, no HIP) services around the drop-in boundary:
 *
 *   1. dualgrid generator  -- deterministic F6-like stand-in meshes written in the exact
 *      NetCDF schema the reference loader reads (the real f6/dualgrid.N.tgz files are
 *      stripped from the reference checkout: /root/reference/.MISSING_LARGE_BLOBS:1-6).
 *   2. domain merger       -- N dualgrid domains -> one partition per GPU
 *      (replaces the MPI index exchange of reference src/comm_data.c:116-255).
 *   3. tiler ("plan")      -- the GPU analogue of init_threads()
 *      (reference src/threads.c:730-788, src/rangelist.c:320-764, src/points_of_color.c):
 *      owner-computes point tiles with duplicated cross faces instead of thread domains
 *      + colours, and pack lists instead of per-colour send lists (src/thread_comm.c).
 */
#ifndef CFDPROXY_HOST_H
#define CFDPROXY_HOST_H

#include "cfdproxy_dropin.h"
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ======================================================================= generator === */
typedef struct {
  int nx, ny, nz;          /* lattice points per axis                                     */
  int ndomains;            /* recursive-coordinate-bisection partitions                    */
  int connectivity;        /* 7: Freudenthal/Kuhn edges (F6-like, ~6.8 faces/point);       */
                           /* 3: Cartesian edges (known-answer mesh);                      */
                           /* CFDP_CONN_IRREGULAR: the edge graph of a random               */
                           /* tetrahedralisation of the lattice's cubes + a hub point every  */
                           /* 1024: ~7 faces/point, 8 to 24 incidences per point, hubs of 60 */
  int normals;             /* 0: lattice normals h^2*e_d ; 1: N(0,1)^3 * h^2 (hash-seeded) */
  int volumes;             /* 0: h^3 ; 1: U(0.5,2)*h^3 (hash-seeded)                       */
  int ghost_faces;         /* 1: also store faces between two ghost points                 */
  int cdf_version;         /* 1 or 2                                                       */
  uint64_t seed;
  int numbering;           /* 0: a domain's points numbered along x, y, z; 1: scrambled      */
  int hubs;                /* irregular meshes: 0 a hub point every 1024 (default), -1 none  */
} cfdp_gen_params;
enum { CFDP_CONN_IRREGULAR = 62 };

/* fill `sd` (mesh arrays, fields = 1.0) and `cd` (halo topology) for one domain, in memory */
int  cfdp_gen_domain(const cfdp_gen_params *gp, int domain, solver_data *sd, comm_data *cd);
/* write one domain as "<prefix>_domain_<domain>_lvl_<lvl>" (reference src/hybrid.f6.c:58-62) */
int  cfdp_gen_write_domain(const cfdp_gen_params *gp, int domain, const char *prefix, int lvl);
/* write any (sd,cd) pair in the dualgrid schema */
int  cfdp_write_domain_file(const char *path, const solver_data *sd, const comm_data *cd,
                            int cdf_version);
/* global lattice id of every point of a generated domain (tests / var-field helpers) */
int  cfdp_gen_global_ids(const cfdp_gen_params *gp, int domain, int *gid /*[nallpoints]*/);

void cfdp_free_solver_data(solver_data *sd);
void cfdp_free_comm_data(comm_data *cd);

/* load "<prefix>_domain_<domain>_lvl_<lvl>" through the drop-in loader
 * (read_solver_data + init_solver_data + read_communication_data)                        */
int  cfdp_load_domain(const char *prefix, int domain, int lvl, solver_data *sd, comm_data *cd);

/* ========================================================================== merger === */
typedef struct {
  int G, r;                /* GPU ranks, my rank                                           */
  int ndomains_total;
  int ndom_local;
  int *domain_ids;         /* [ndom_local] ascending                                       */
  int *own_offset;         /* [ndom_local+1] merged id of each local domain's first point  */
  int **local2merged;      /* [ndom_local][nallpoints_d] file numbering -> merged id       */
  int nghost;              /* merged ghosts                                                */
  int *ghost_domain;       /* [nghost] owner domain of merged ghost j                      */
  int *ghost_idx;          /* [nghost] owner-domain-local id                               */
  int npartners;
  int *partner;            /* [npartners] partner ranks, ascending                         */
  int *want_off;           /* [npartners+1] into ghost_* (ghosts are grouped by partner)   */
  long nfaces_in, nfaces_dropped;
} cfdp_merge_info;

/* rank that owns `domain`: contiguous blocks of domain ids by default (cfdp_rank_domains gives
 * a rank's block), or the explicit map installed with cfdp_set_domain_map (NULL removes it) --
 * for dualgrid files whose domain numbering is not spatially coherent, built by
 * cfdp_cluster_domains() from the commpartner graph of the files (cfdp_domain_graph; reference
 * fields commpartner / recvcount, src/comm_data.c:79-112).  cfdp_rank_domain_list: a rank's
 * domains under whichever mapping is active, ascending; returns their number.               */
int  cfdp_domain_rank(int domain, int ndomains_total, int G);
void cfdp_rank_domains(int r, int ndomains_total, int G, int *first, int *count);
void cfdp_set_domain_map(const int *rank_of_domain, int ndomains_total, int G);
int  cfdp_rank_domain_list(int r, int ndomains_total, int G, int *ids);
long cfdp_cluster_domains(int ndomains_total, int G, const int *xadj, const int *adj, const int *wgt,
                          int *rank_of_domain);
int  cfdp_domain_graph(const char *prefix, int lvl, int ndomains_total, int **xadj, int **adj, int **wgt);

/* merge `ndom_local` loaded domains into one partition.  out_cd gets nProc=G, iProc=r,
 * ndomains=G, commpartner/recvcount/recvindex filled; sendcount/sendindex are filled by
 * cfdp_merge_set_send() once the partners' request lists are known.                      */
int  cfdp_merge_domains(int ndom_local, const int *domain_ids, const solver_data *sds,
                        const comm_data *cds, int ndomains_total, int G, int r,
                        solver_data *out_sd, comm_data *out_cd, cfdp_merge_info **info);
/* partner `s` wants `count` of my points, given as (domain, idx) pairs in message order */
int  cfdp_merge_set_send(comm_data *out_cd, const cfdp_merge_info *info, int s, int count,
                         const int *want_domain, const int *want_idx);
/* all G ranks in one process: wire every rank's send lists from its partners' requests   */
void cfdp_merge_link_group(int G, comm_data **cds, cfdp_merge_info **infos);
/* scatter a merged per-point field (rowlen doubles) back to domain `dl` file numbering   */
void cfdp_merge_scatter(const cfdp_merge_info *info, int dl, int npoints_d, int rowlen,
                        const double *merged, double *out);
void cfdp_merge_info_free(cfdp_merge_info *info);

/* ============================================================================ tiler === */
typedef struct {
  int pstart;              /* first point (new numbering) owned by the tile               */
  int npts;                /* owned points                                                 */
  int nhalo;               /* points read but not owned                                    */
  int nfaces;              /* faces touching an owned point (cross-tile faces duplicated)  */
  int ninc;                /* (point,face) incidences                                      */
  int halo_off;            /* into cfdp_plan.halo_idx                                      */
  int blob_off;            /* into cfdp_plan.blob, in 16-byte units                        */
  int blob_qw;             /* blob length in 16-byte units                                 */
} cfdp_tile_desc;

/* incidence word: [15:0] local index of the OTHER end point (>= npts: halo slot),
 * [30:16] tile-local face, [31] 1 if the owned point is p1 of the face (contribution is
 * subtracted, reference src/gradients.c:87-107,126-131)                                  */
#define CFDP_INC_NBR(w)  ((w) & 0xFFFFu)
#define CFDP_INC_FACE(w) (((w) >> 16) & 0x7FFFu)
#define CFDP_INC_SIGN(w) ((w) >> 31)

typedef struct {
  int tile_points;         /* owned points per tile (<= 1024)                              */
  int boundary_first;      /* tile send points first (comm/compute overlap)                */
  int supertile;           /* tiles per cluster of the launch order (L2 reuse); <= 1: off  */
} cfdp_plan_opts;

typedef struct cfdp_plan {
  int nown, nall;
  long nfaces_used;        /* faces with >=1 owned end                                     */
  int ntiles, nbtiles;     /* tiles [0,nbtiles) hold every send point                      */
  int tile_points;
  int *new2old, *old2new;  /* [nall] device numbering <-> file numbering                   */
  cfdp_tile_desc *tiles;
  int *halo_idx;  long nhalo_total;
  unsigned char *blob; long blob_bytes;
  double *vol;             /* [nown] pvolume, new numbering                                */
  int *degree;             /* [nown] incidences per point, new numbering                   */
  long lds_grad, lds_flux; /* dynamic LDS bytes the kernels need (max over all tiles)      */
  long lds_grad_cls[2], lds_flux_cls[2]; /* same, [0] boundary tiles, [1] interior tiles    */
  long nfaces_dup, ninc_total;
  /* exchange (empty when the partition has no partners) */
  int npartners;
  int *partner;            /* [npartners] ranks                                            */
  int *send_off;           /* [npartners+1]                                                */
  int *send_idx;           /* [send_off[npartners]] NEW ids, message order                 */
  int *recv_off;           /* [npartners+1]; partner s fills rows nown+recv_off[s]...      */
  /* launch groups of the interior tiles: group k = tiles [group_begin[k], group_begin[k+1]), group_begin[0] == nbtiles,
   * group_begin[ngroups] == ntiles.  Tiles are grouped by the capacity class of the kernels they fit (cfdp_tile_class):
   * tiles no fixed-capacity kernel holds (a hub point with hundreds of faces) form the LAST group and get a launch of
   * their own, so that one such tile does not put every tile of the partition into the slowest kernel form; the small
   * and the large image are only separated where both sets are big enough to pay for a second launch                 */
  int ngroups;
  int group_begin[5];
  int group_class[4];      /* the largest class in the group                                */
} cfdp_plan;

/* capacity classes of a tile (its workgroup has block = 4 lanes x tile_points threads, rounded up to whole waves; pieces are
 * 16 bytes per thread): SMALL fits gg_fused_split_kernel<5, 3, 3, 3> (blob <= 5 pieces, staged rows <= 3 pieces at 4 per
 * row), LARGE fits <6, 4, 3, 4>, GENERIC needs the kernels that take any tile shape                                  */
enum { CFDP_TILE_SMALL = 0, CFDP_TILE_LARGE = 1, CFDP_TILE_GENERIC = 2 };
static inline int cfdp_tile_class(int tile_points, int rows, long blob_bytes) {
  const long block = (((long)tile_points * 4 + 63) / 64) * 64;
  if (blob_bytes <= 5 * block * 16 && rows <= 3 * block / 4) return CFDP_TILE_SMALL;
  if (blob_bytes <= 6 * block * 16 && rows <= block) return CFDP_TILE_LARGE;
  return CFDP_TILE_GENERIC;
}
int cfdp_tile_class_of(int tile_points, int rows, long blob_bytes); /* the same, callable through the ABI */

void cfdp_plan_default_opts(cfdp_plan_opts *o);
cfdp_plan *cfdp_plan_build(const solver_data *sd, const comm_data *cd, const cfdp_plan_opts *o);

/* The two heavy, data-parallel stages of the plan can be done elsewhere -- on the device
 * (cfdp_plan_build_gpu, cfdproxy_hip.h: the reference's init_thread_rangelist / thread_comm preprocessing,
 * src/rangelist.c:500-764, src/thread_comm.c:27-432, as HIP kernels) -- while tile growth, tile order,
 * renumbering and the pack lists stay on the host.  A provider must produce what the host stage produces,
 * bit for bit (tests compare the plans):
 *   csr    stage 1: xadj[nown+1], adj_face[nadj] (file face id | bit 31 when the owned end is p1),
 *          adj_other[nadj] (file id of the other end), a point's entries in file face order; malloc'd
 *   blobs  stage 5: P->tiles, blob, blob_bytes, halo_idx, nhalo_total, lds_grad/_flux(_cls), nfaces_dup,
 *          ninc_total from the tiling (P->ntiles, nbtiles, nown, old2new are set); malloc'd            */
typedef struct {
  const int *xadj, *adj_face, *adj_other;  /* stage 1 */
  const int *order;                        /* [nown] file ids of the owned points, tile-major            */
  const int *tile_first;                   /* [ntiles+1] into order                                      */
  const int *tile_of;                      /* [nown] tile of an owned point (file id)                    */
} cfdp_tiling;
typedef struct {
  int (*csr)(const solver_data *sd, int **xadj, int **adj_face, int **adj_other, long *nfaces_used, void *ctx);
  int (*blobs)(const solver_data *sd, const cfdp_tiling *tl, struct cfdp_plan *P, void *ctx);
  void *ctx;
} cfdp_plan_stages;
cfdp_plan *cfdp_plan_build_with(const solver_data *sd, const comm_data *cd, const cfdp_plan_opts *o,
                                const cfdp_plan_stages *stages);
int cfdp_plan_host_blobs(const solver_data *sd, const cfdp_tiling *tl, struct cfdp_plan *P); /* the host's stage 5 */
void cfdp_plan_free(cfdp_plan *p);
/* layout of a tile blob: [nx[E] | ny[E] | nz[E]] (one 16-byte padded plane per normal component),
 * then the incidence words, then the per-point offsets */
static inline long cfdp_blob_plane_bytes(int nfaces) { return ((long)nfaces * 8 + 15) & ~15L; }
static inline long cfdp_blob_fn_bytes(int nfaces) { return 3 * cfdp_blob_plane_bytes(nfaces); }
static inline long cfdp_blob_inc_bytes(int ninc) { return ((long)ninc * 4 + 15) & ~15L; }
static inline long cfdp_blob_off_bytes(int npts) { return ((long)(npts + 1) * 4 + 15) & ~15L; }
/* LONG INCIDENCE LISTS (round 6).  A lane group walks its point's whole list, batch after batch of dependent LDS round trips, so
 * a point with 60-75 faces (one in a thousand on an unstructured mesh) keeps its wave -- and the tile that waits for it --
 * busy five times as long as its neighbours.  A list of more than CFDP_LONG_LIST entries is therefore cut into nchunks =
 * ceil(deg / CFDP_LIST_CHUNK) (<= CFDP_MAX_CHUNKS) chunks of ceil(deg / nchunks) entries: the point's own lane group takes the
 * first, HELPER lane groups of the same tile -- tile-local slots npts .. npts + nhelp - 1, which own no row -- take the others
 * at the same time, and the partial sums are added to the point's in LDS in chunk order (deterministic; every kernel form of
 * a plan shares the chunking).  In the blob: offsets word li = offset | (nchunks - 1) << 24; behind the offsets, only in tiles
 * that have helpers, [nhelp | helper h: target li | chunk << 16 | ... pad 16][192 bytes of scratch per helper].  A tile whose
 * points and helpers together exceed tile_points (a tiler that did not plan for them) simply has no list cut.              */
enum { CFDP_LONG_LIST = 32, CFDP_LIST_CHUNK = 28, CFDP_MAX_CHUNKS = 16 };
/* (the two thresholds as the library uses them: the constants above unless CFDP_LONG_LIST / CFDP_LIST_CHUNK in the environment say
 * otherwise -- development; read once, host/tiling.c) */
int cfdp_long_list(void);
int cfdp_list_chunk(void);
static inline int cfdp_list_chunks(int deg, int tile_points) { /* (a tile has tile_points lane groups: at most a quarter per list) */
  if (deg <= cfdp_long_list()) return 1;
  int cap = tile_points / 4 < CFDP_MAX_CHUNKS ? tile_points / 4 : CFDP_MAX_CHUNKS;
  if (cap < 1) cap = 1;
  const int n = (deg + cfdp_list_chunk() - 1) / cfdp_list_chunk();
  return n > cap ? cap : n;
}
int cfdp_list_chunks_of(int deg, int tile_points); /* the same, callable through the ABI */
static inline long cfdp_blob_help_bytes(int nhelp) { return nhelp > 0 ? (((long)(1 + nhelp) * 4 + 15) & ~15L) + 192L * nhelp : 0; }
#define CFDP_OFF_START(w)  ((w) & 0xFFFFFFu)
#define CFDP_OFF_CHUNKS(w) (((w) >> 24) + 1u)
/* algorithmic bytes of one gradient / flux pass */
double cfdp_algo_bytes_grad(long nfaces, long nown, long nadd);
double cfdp_algo_bytes_flux(long nfaces, long nown, long nadd);

/* var fields used by tests, bench and the driver */
enum { CFDP_VAR_ONE = 0, CFDP_VAR_HASH = 1, CFDP_VAR_LINEAR = 2 };
void cfdp_fill_var(double (*var)[NGRAD], const int *gid, int npoints, int kind,
                   int nx, int ny, int nz);

const char *cfdp_host_version(void);

/* Experiment switches (host/experiments.c): environment variables that make the library compute something other than
 * the product path (timing experiments with wrong values, protocol ablations, fault / failure injection, test delays).
 * cfdp_experiment_getenv(name): the variable's value if it is set AND CFDP_EXPERIMENTS=1, else NULL; says once on
 * stderr which of the two happened.  cfdp_experiment_switches(): the registered names, space separated.
 * cfdp_experiments_active(buf, len): how many of them are set with the master key present (what a benchmark must
 * refuse to report under); buf receives "NAME=value ..." */
const char *cfdp_experiment_getenv(const char *name);
const char *cfdp_experiment_switches(void);
int cfdp_experiments_active(char *buf, size_t len);

#ifdef __cplusplus
}
#endif
#endif
