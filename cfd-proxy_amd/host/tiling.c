/*
 * tiling.c -- the GPU analogue of init_threads() (reference src/threads.c:730-788).
 *
 * What the reference builds for OpenMP threads, and what replaces it here:
 *
 *   reference                                        |  this file
 *   -------------------------------------------------+---------------------------------------
 *   init_thread_id: contiguous point ranges per      |  point TILES grown breadth-first over
 *     thread (src/rangelist.c:320-398)               |    the face graph (compact, ~128 points),
 *                                                    |    one workgroup each
 *   init_thread_rangelist: every thread copies each  |  every tile gets a private copy of each
 *     face touching one of its points; face class    |    face touching one of its points
 *     says which end it may write; cross faces are   |    ("owner computes", cross-tile faces
 *     processed once per side (:513-523,567-608)     |    duplicated) -> no atomics, no colouring
 *   colours of <=96 faces + first/last point lists   |  per point incidence lists (CSR): a
 *     for strip-mined zero-init and finalise         |    lane accumulates a point in registers,
 *     (:654-703, src/points_of_color.c)              |    starts from 0 and scales by 1/volume once
 *   faces touching sent points sorted first          |  send points are tiled FIRST (tiles
 *     (ttype 0-2 before 3-5, :572-607)               |    [0,nbtiles)): their kernel + pack run
 *                                                    |    ahead of / beside the interior tiles
 *   gather_sendcount/recvcount per colour            |  one pack list per partner; ghost rows
 *     (src/thread_comm.c:27-432)                     |    renumbered so a message IS a row block
 *
 * Points are renumbered tile-major (new2old/old2new); results are returned in file order.
 */
#include "cfdproxy_host.h"
#include "host_util.h"

#include <omp.h>
#include <stdlib.h>
#include <string.h>

int cfdp_tile_class_of(int tile_points, int rows, long blob_bytes) { return cfdp_tile_class(tile_points, rows, blob_bytes); }
int cfdp_list_chunks_of(int deg, int tile_points) { return cfdp_list_chunks(deg, tile_points); }
static int g_long_list = CFDP_LONG_LIST, g_list_chunk = CFDP_LIST_CHUNK;
int cfdp_long_list(void) { return g_long_list; }
int cfdp_list_chunk(void) { return g_list_chunk; }
static void list_thresholds_from_env(void) { /* at the start of every plan build (development: CFDP_LONG_LIST, CFDP_LIST_CHUNK) */
  const char *a = getenv("CFDP_LONG_LIST"), *b = getenv("CFDP_LIST_CHUNK");
  g_long_list = a && atoi(a) >= 4 ? atoi(a) : CFDP_LONG_LIST;
  g_list_chunk = b && atoi(b) >= 2 ? atoi(b) : CFDP_LIST_CHUNK;
}

void cfdp_plan_default_opts(cfdp_plan_opts *o) {
  o->tile_points = 64;
  o->boundary_first = 1;
  o->supertile = 64;
  const char *e = getenv("CFDP_SUPERTILE"); /* experiments */
  if (e) o->supertile = atoi(e);
}

typedef struct {
  int nown;
  const int *xadj;       /* [nown+1] */
  const int *adj_other;  /* other end point (old id) */
  int *tile_of;          /* [nown] */
  int *stamp;            /* [nown] */
  unsigned char *seeded; /* [nown] */
  int *seedq; int sq_head, sq_tail;
  int *lq;               /* local queue scratch [nown] */
  int *order; int norder;
  int *tile_first; int ntiles, cap_tiles;
  int cursor;
  /* optional bound on a tile's halo (points it reads but does not own): hseen[q] == tile + 1
   * marks q as a member or a halo point of the tile being grown; NULL = unbounded */
  int *hseen;
  int halo_cap;
  /* hard budgets of a tile, checked BEFORE a point joins (0 = none): bytes of its blob and rows it stages (own +
   * halo), so that every tile of the launch fits the fastest kernel's LDS image -- one oversized tile would
   * otherwise put the whole launch into the next capacity class (fewer workgroups per CU) */
  long blob_cap;
  int rows_cap;
  /* the NEXT capacity of the kernels (0 = none): a tile that the budgets above would close below fill_min points goes on
   * under these -- on graphs with more faces and neighbours per point than the lattice stand-ins (unstructured dual
   * grids) a tile cut to the smallest image keeps a quarter of its lanes idle, which costs more than the larger image */
  long blob_cap2;
  int rows_cap2, fill_min;
  int tile_points; /* of the plan (boundary tiles are grown to half of it) */
  int split_lists; /* long incidence lists are cut into chunks that helper lane groups of the tile walk (cfdproxy_host.h) */
} tiler;

static void tiler_open_tile(tiler *T) {
  if (T->ntiles + 1 >= T->cap_tiles) {
    T->cap_tiles = T->cap_tiles ? 2 * T->cap_tiles : 1024;
    T->tile_first = realloc(T->tile_first, (size_t)(T->cap_tiles + 1) * sizeof(int));
  }
  T->tile_first[T->ntiles] = T->norder;
}

/* tile all points with mask[p]==want (mask==NULL: all) into tiles of <= TP points */
static void tiler_pass(tiler *T, const unsigned char *mask, int want, int TP) {
  const int nown = T->nown;
  int remaining = 0;
  for (int p = 0; p < nown; p++)
    if (T->tile_of[p] < 0 && (!mask || mask[p] == want)) remaining++;
  T->cursor = 0;
  while (remaining > 0) {
    const int t = T->ntiles;
    tiler_open_tile(T);
    int cnt = 0, head = 0, tail = 0, seen = 0, rejected = -1;
    int nhelp = 0; /* helper lane groups the tile's long lists take (cfdproxy_host.h): they share the tile's TP groups */
    long ninc = 0, ninternal = 0; /* incidences of the tile so far; faces with both ends in it */
    long bcap = T->blob_cap;     /* this tile's budgets: the smallest image first */
    int rcap = T->rows_cap, hcap = T->halo_cap, upgraded = 0;
#define TILE_NEXT_CAPACITY() (!upgraded && T->blob_cap2 > 0 && cnt < T->fill_min && TP >= T->fill_min && \
                              (upgraded = 1, bcap = T->blob_cap2, rcap = T->rows_cap2, hcap = T->rows_cap2 - TP, 1))
    while (cnt + nhelp < TP) {
      /* a tile made of leftovers scattered between finished tiles reads ~14 rows per point; close
       * it early rather than let one such tile size the LDS image of the whole launch */
      if (T->hseen && cnt >= 4 && seen - cnt > hcap && !(TILE_NEXT_CAPACITY() && seen - cnt <= hcap)) break;
      if (head == tail) { /* need a seed */
        int seed = -1;
        while (T->sq_head < T->sq_tail) {
          int q = T->seedq[T->sq_head++];
          if (T->tile_of[q] < 0 && (!mask || mask[q] == want) && T->stamp[q] != t + 1) {
            seed = q;
            break;
          }
        }
        while (seed < 0 && T->cursor < nown) {
          int q = T->cursor++;
          if (T->tile_of[q] < 0 && (!mask || mask[q] == want) && T->stamp[q] != t + 1) seed = q;
        }
        if (seed < 0) break;
        T->stamp[seed] = t + 1;
        T->lq[tail++] = seed;
      }
      int p = T->lq[head];
      const int help_add = T->split_lists ? cfdp_list_chunks(T->xadj[p + 1] - T->xadj[p], T->tile_points) - 1 : 0;
      if (cnt >= 1 && cnt + 1 + nhelp + help_add > TP) { /* no lane groups left for this point's chunks: it seeds a later tile */
        rejected = p;
        break;
      }
      int internal_add = 0;
      for (int e = T->xadj[p]; e < T->xadj[p + 1]; e++) {
        int q = T->adj_other[e];
        if (q < nown && q != p && T->tile_of[q] == t) internal_add++;
      }
      if (T->hseen && T->blob_cap > 0 && cnt >= 1) { /* would the tile still fit with p in it? (exact for the blob) */
        int newrows = T->hseen[p] != t + 1 ? 1 : 0;
        for (int e = T->xadj[p]; e < T->xadj[p + 1]; e++)
          if (T->hseen[T->adj_other[e]] != t + 1) newrows++; /* (an upper bound: parallel faces count twice) */
        const long I2 = ninc + (T->xadj[p + 1] - T->xadj[p]), E2 = I2 - (ninternal + internal_add);
        const long blob = cfdp_blob_fn_bytes((int)E2) + cfdp_blob_inc_bytes((int)I2) + cfdp_blob_off_bytes(cnt + 1) +
                          cfdp_blob_help_bytes(nhelp + help_add);
        if ((blob > bcap || seen + newrows > rcap) &&
            !(TILE_NEXT_CAPACITY() && blob <= bcap && seen + newrows <= rcap)) {
          rejected = p; /* stays un-tiled: it seeds a later tile */
          break;
        }
      }
      head++;
      T->tile_of[p] = t;
      T->order[T->norder++] = p;
      cnt++;
      nhelp += help_add;
      remaining--;
      ninc += T->xadj[p + 1] - T->xadj[p];
      ninternal += internal_add;
      if (T->hseen && T->hseen[p] != t + 1) { T->hseen[p] = t + 1; seen++; }
      for (int e = T->xadj[p]; e < T->xadj[p + 1]; e++) {
        int q = T->adj_other[e];
        if (T->hseen && T->hseen[q] != t + 1) { T->hseen[q] = t + 1; seen++; }
        if (q >= nown || T->tile_of[q] >= 0 || T->stamp[q] == t + 1) continue;
        if (mask && mask[q] != want) continue;
        T->stamp[q] = t + 1;
        T->lq[tail++] = q;
      }
    }
    /* frontier of the closed tile seeds the following tiles (keeps tiles adjacent) */
    for (int i = head; i < tail; i++) {
      int q = T->lq[i];
      if (!T->seeded[q]) { T->seeded[q] = 1; T->seedq[T->sq_tail++] = q; }
      else if (q == rejected) T->seedq[T->sq_tail++] = q; /* it was popped as a seed already: queue it again */
    }
    T->ntiles++;
    CFDP_ASSERT(cnt > 0);
  }
#undef TILE_NEXT_CAPACITY
}

/* EXPERIMENT (CFDP_GROW_FRONTS=M > 1; SURVEY section 8f-1 "tile growth on the device"): the growth a frontier-parallel
 * device kernel would perform -- M tiles grow AT THE SAME TIME, one point per tile and round, a point joining the tile
 * that claims it first -- run sequentially here so that its tiles can be measured (halo rows, face duplication, LDS
 * class, kernel time) against the one-front growth above before anything is moved to the device.  Same budgets per
 * tile; halo accounting by a per-front hash set.  The result is deterministic (fronts take turns in order). */
typedef struct { int *key; unsigned *gen; unsigned cur; int mask; int count; } hset;
static void hset_init(hset *h, int max_keys) {
  int cap = 64;
  while (cap < 4 * max_keys) cap *= 2;
  h->key = cfdp_malloc((size_t)cap * sizeof(int));
  h->gen = cfdp_calloc((size_t)cap, sizeof(unsigned));
  h->cur = 1; h->mask = cap - 1; h->count = 0;
}
static void hset_reset(hset *h) {
  if (++h->cur == 0) { memset(h->gen, 0, (size_t)(h->mask + 1) * sizeof(unsigned)); h->cur = 1; }
  h->count = 0;
}
static int hset_has(const hset *h, int k) {
  unsigned x = ((unsigned)k * 2654435761u) >> 7;
  for (;;) {
    x &= (unsigned)h->mask;
    if (h->gen[x] != h->cur) return 0;
    if (h->key[x] == k) return 1;
    x++;
  }
}
static void hset_add(hset *h, int k) {
  unsigned x = ((unsigned)k * 2654435761u) >> 7;
  for (;;) {
    x &= (unsigned)h->mask;
    if (h->gen[x] != h->cur) { h->gen[x] = h->cur; h->key[x] = k; h->count++; return; }
    if (h->key[x] == k) return;
    x++;
  }
}
typedef struct {
  int active, t, cnt, head, tail, first;  /* first: where the tile's points start in the front's member list */
  long ninc, ninternal;
  int *q;      /* BFS queue of this front: candidates in discovery order */
  int *members;
  hset seen;   /* members + halo points of the tile */
} front;

static void tiler_pass_multifront(tiler *T, const unsigned char *mask, int want, int TP, int M) {
  const int nown = T->nown;
  int remaining = 0;
  for (int p = 0; p < nown; p++)
    if (T->tile_of[p] < 0 && (!mask || mask[p] == want)) remaining++;
  T->cursor = 0;
  const int qcap = 64 * TP + 4096;
  front *F = cfdp_calloc((size_t)M, sizeof(front));
  for (int i = 0; i < M; i++) {
    F[i].q = cfdp_malloc((size_t)qcap * sizeof(int));
    F[i].members = cfdp_malloc((size_t)TP * sizeof(int));
    hset_init(&F[i].seen, 40 * TP);
  }
  /* tiles are numbered when they OPEN; their points are appended to T->order when they CLOSE, so T->tile_first is
   * rebuilt at the end from the closing order */
  int *close_tile = cfdp_malloc(((size_t)nown + 1) * sizeof(int)), nclosed = 0;
  int *close_first = cfdp_malloc(((size_t)nown + 2) * sizeof(int));
  const int t_base = T->ntiles;
  int opened = 0;
  while (remaining > 0) {
    int progressed = 0;
    for (int i = 0; i < M && remaining > 0; i++) {
      front *f = &F[i];
      if (!f->active) { /* open a tile from a seed */
        int seed = -1;
        while (T->sq_head < T->sq_tail) {
          int q = T->seedq[T->sq_head++];
          if (T->tile_of[q] < 0 && (!mask || mask[q] == want)) { seed = q; break; }
        }
        while (seed < 0 && T->cursor < nown) {
          int q = T->cursor++;
          if (T->tile_of[q] < 0 && (!mask || mask[q] == want)) seed = q;
        }
        if (seed < 0) continue;
        f->active = 1; f->t = t_base + opened++; f->cnt = 0; f->head = 0; f->tail = 0; f->ninc = f->ninternal = 0;
        hset_reset(&f->seen);
        f->q[f->tail++] = seed;
      }
      /* one point for this front: the next unclaimed candidate */
      int took = 0, close = 0;
      while (f->head < f->tail && !took) {
        const int p = f->q[f->head++];
        if (T->tile_of[p] >= 0) continue; /* claimed by another front meanwhile */
        int internal_add = 0, newrows = hset_has(&f->seen, p) ? 0 : 1;
        for (int e = T->xadj[p]; e < T->xadj[p + 1]; e++) {
          const int q = T->adj_other[e];
          if (q < nown && q != p && T->tile_of[q] == f->t) internal_add++;
          if (!hset_has(&f->seen, q)) newrows++;
        }
        if (T->blob_cap > 0 && f->cnt >= 1) {
          const long I2 = f->ninc + (T->xadj[p + 1] - T->xadj[p]), E2 = I2 - (f->ninternal + internal_add);
          const long blob = cfdp_blob_fn_bytes((int)E2) + cfdp_blob_inc_bytes((int)I2) + cfdp_blob_off_bytes(f->cnt + 1);
          if (blob > T->blob_cap || f->seen.count + newrows > T->rows_cap) {
            f->head--; /* p stays un-tiled and seeds a later tile */
            close = 1;
            break;
          }
        }
        T->tile_of[p] = f->t;
        f->members[f->cnt++] = p;
        remaining--;
        f->ninc += T->xadj[p + 1] - T->xadj[p];
        f->ninternal += internal_add;
        hset_add(&f->seen, p);
        for (int e = T->xadj[p]; e < T->xadj[p + 1]; e++) {
          const int q = T->adj_other[e];
          hset_add(&f->seen, q);
          if (q >= nown || T->tile_of[q] >= 0) continue;
          if (mask && mask[q] != want) continue;
          if (f->tail < qcap) f->q[f->tail++] = q; /* (duplicates are skipped when popped) */
        }
        took = 1;
        progressed = 1;
      }
      if (!took && !close) close = 1; /* the front ran dry */
      if (f->cnt >= TP) close = 1;
      if (f->cnt >= 4 && f->seen.count - f->cnt > T->halo_cap) close = 1;
      if (close) {
        if (f->cnt > 0) {
          close_tile[nclosed] = f->t;
          close_first[nclosed] = T->norder;
          for (int k = 0; k < f->cnt; k++) T->order[T->norder++] = f->members[k];
          nclosed++;
        } else {
          opened--; /* (the tile number is given back only if it was the last one opened; else it stays empty) */
          if (f->t != t_base + opened) { opened++; close_tile[nclosed] = f->t; close_first[nclosed] = T->norder; nclosed++; }
        }
        for (int k = f->head; k < f->tail; k++) { /* its frontier seeds later tiles */
          const int q = f->q[k];
          if (T->tile_of[q] < 0 && T->sq_tail < 2 * nown) T->seedq[T->sq_tail++] = q;
        }
        f->active = 0;
        progressed = 1;
      }
    }
    if (!progressed) break;
  }
  for (int i = 0; i < M; i++) /* close what is still open */
    if (F[i].active && F[i].cnt > 0) {
      close_tile[nclosed] = F[i].t;
      close_first[nclosed] = T->norder;
      for (int k = 0; k < F[i].cnt; k++) T->order[T->norder++] = F[i].members[k];
      nclosed++;
    }
  CFDP_ASSERT(remaining == 0);
  /* renumber the tiles of this pass in closing order (the order their points were appended in) */
  int *renum = cfdp_malloc(((size_t)opened + 1) * sizeof(int));
  for (int k = 0; k < opened; k++) renum[k] = -1;
  int nt = 0;
  for (int k = 0; k < nclosed; k++) {
    const int next_first = k + 1 < nclosed ? close_first[k + 1] : T->norder;
    if (next_first == close_first[k]) continue; /* an empty tile */
    renum[close_tile[k] - t_base] = t_base + nt;
    tiler_open_tile(T);
    T->tile_first[T->ntiles] = close_first[k];
    T->ntiles++;
    nt++;
  }
  for (int i = 0; i < nown; i++)
    if (T->tile_of[i] >= t_base && T->tile_of[i] - t_base < opened && renum[T->tile_of[i] - t_base] >= 0 &&
        (!mask || mask[i] == want))
      T->tile_of[i] = renum[T->tile_of[i] - t_base];
  free(renum); free(close_tile); free(close_first);
  for (int i = 0; i < M; i++) { free(F[i].q); free(F[i].members); free(F[i].seen.key); free(F[i].seen.gen); }
  free(F);
}

/* small open-addressing map key -> first-touch index, cleared in O(1) by a generation count */
typedef struct { int *key, *val; unsigned *gen; unsigned cur; int mask; } lmap;
static void lmap_init(lmap *m, int max_keys) {
  int cap = 64;
  while (cap < 4 * max_keys) cap *= 2;
  m->key = cfdp_malloc((size_t)cap * sizeof(int));
  m->val = cfdp_malloc((size_t)cap * sizeof(int));
  m->gen = cfdp_calloc((size_t)cap, sizeof(unsigned));
  m->cur = 0;
  m->mask = cap - 1;
}
static void lmap_free(lmap *m) { free(m->key); free(m->val); free(m->gen); }
static inline void lmap_reset(lmap *m) {
  if (++m->cur == 0) { memset(m->gen, 0, (size_t)(m->mask + 1) * sizeof(unsigned)); m->cur = 1; }
}
/* index of k; a new key gets (*next)++ */
static inline int lmap_index(lmap *m, int k, int *next) {
  unsigned h = ((unsigned)k * 2654435761u) >> 7;
  for (;;) {
    h &= (unsigned)m->mask;
    if (m->gen[h] != m->cur) { m->gen[h] = m->cur; m->key[h] = k; m->val[h] = (*next)++; return m->val[h]; }
    if (m->key[h] == k) return m->val[h];
    h++;
  }
}

/* stage 1 on the host.  Every thread streams the whole face list but only keeps the ends that fall into its own
 * range of points: the scattered writes stay inside a cache-sized slice, and a point's list is in file order
 * whatever the thread count (the order the kernels add a point's faces in).
 * Out: xadj[nown+1], adj_face[nadj] (bit 31: the owned end is p1), adj_other[nadj] (the other end, file id). */
static int host_csr(const solver_data *sd, int **xadj_out, int **adj_face_out, int **adj_other_out, long *used_out, void *ctx) {
  (void)ctx;
  const int nown = sd->nownpoints, nf = sd->nfaces;
  int *xadj = cfdp_calloc((size_t)nown + 2, sizeof(int));
  long used = 0;
#pragma omp parallel reduction(+ : used) num_threads(cfdp_host_threads())
  {
    const int nth = omp_get_num_threads(), th = omp_get_thread_num();
    const int lo = (int)((long)nown * th / nth), hi = (int)((long)nown * (th + 1) / nth);
    for (int f = 0; f < nf; f++) {
      int a = sd->fpoint[f][0], b = sd->fpoint[f][1];
      if (a >= lo && a < hi) xadj[a + 1]++;
      if (b >= lo && b < hi) xadj[b + 1]++;
      if (th == 0 && (a < nown || b < nown)) used++;
    }
  }
  for (int p = 0; p < nown; p++) xadj[p + 1] += xadj[p];
  const int nadj = xadj[nown];
  int *adj_face = cfdp_malloc((size_t)(nadj ? nadj : 1) * sizeof(int)); /* bit31: owned end is p1 */
  int *adj_other = cfdp_malloc((size_t)(nadj ? nadj : 1) * sizeof(int));
  int *fill = cfdp_malloc((size_t)nown * sizeof(int));
  memcpy(fill, xadj, (size_t)nown * sizeof(int));
#pragma omp parallel num_threads(cfdp_host_threads())
  {
    const int nth = omp_get_num_threads(), th = omp_get_thread_num();
    const int lo = (int)((long)nown * th / nth), hi = (int)((long)nown * (th + 1) / nth);
    for (int f = 0; f < nf; f++) {
      int a = sd->fpoint[f][0], b = sd->fpoint[f][1];
      if (a >= lo && a < hi) { adj_face[fill[a]] = f; adj_other[fill[a]++] = b; }
      if (b >= lo && b < hi) { adj_face[fill[b]] = (int)((unsigned)f | 0x80000000u); adj_other[fill[b]++] = a; }
    }
  }
  free(fill);
  *xadj_out = xadj; *adj_face_out = adj_face; *adj_other_out = adj_other; *used_out = used;
  return 0;
}

/* helper lane groups of a tile (cfdproxy_host.h, long incidence lists): the chunks beyond the first of every long list -- or
 * none at all where points and helpers together do not fit the tile's lane groups */
static int tile_helpers(const cfdp_tiling *tl, int tile_points, int ts, int np, int split_lists) {
  if (!split_lists) return 0;
  int nh = 0;
  for (int li = 0; li < np; li++) {
    const int p = tl->order[ts + li];
    nh += cfdp_list_chunks(tl->xadj[p + 1] - tl->xadj[p], tile_points) - 1;
  }
  return np + nh <= tile_points ? nh : 0;
}

/* stage 5 on the host.  Tiles are independent: pass A sizes every tile, a prefix sum places its blob and its halo
 * list, pass B fills them -- both passes in parallel over tiles.  The tile-local numbering of faces and halo
 * points (first touch, walking the tile's points and their faces in file order) lives in small per-thread hash
 * maps instead of mesh-sized stamp arrays.  Needs P->ntiles, nbtiles, nown, old2new; fills tiles, blob, halo_idx,
 * the LDS sizes and the totals.                                                                              */
static int host_blobs(const solver_data *sd, const cfdp_tiling *tl, cfdp_plan *P, void *ctx) {
  (void)ctx;
  const int nown = P->nown;
  P->tiles = cfdp_calloc((size_t)P->ntiles, sizeof(cfdp_tile_desc));
  int max_inc = 1;
  for (int t = 0; t < P->ntiles; t++) {
    int n = 0;
    for (int i = tl->tile_first[t]; i < tl->tile_first[t + 1]; i++) n += tl->xadj[tl->order[i] + 1] - tl->xadj[tl->order[i]];
    if (n > max_inc) max_inc = n;
  }
  long lds_g[2] = {0, 0}, lds_f[2] = {0, 0};
  long *boff = cfdp_calloc((size_t)P->ntiles + 1, sizeof(long)); /* blob offsets, bytes */
  long *hoff = cfdp_calloc((size_t)P->ntiles + 1, sizeof(long)); /* halo offsets, entries */
  long dup_total = 0, inc_total = 0;
  int bad = 0;
  /* EXPERIMENT (CFDP_EXP_OWNED_NORMALS=1, timing only -- the values are WRONG): the normals of a face cut by two tiles
   * are stored with ONE of them (the lower tile id); the other tile's planes simply do not hold them (its incidence
   * words point past its planes).  What the fused pass and its movement floor take then is an UPPER BOUND on what
   * "cut-face normals owned by one tile, fetched by the other through L2" can gain: the fetch itself is free here. */
  const int split_lists = !(getenv("CFDP_SPLIT_LISTS") && atoi(getenv("CFDP_SPLIT_LISTS")) == 0);
  const char *exp_owned_s = cfdp_experiment_getenv("CFDP_EXP_OWNED_NORMALS"); /* honoured only with CFDP_EXPERIMENTS=1 */
  const int exp_owned = exp_owned_s && atoi(exp_owned_s) != 0;
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) {
      for (int t = 0; t < P->ntiles; t++) {
        boff[t + 1] += boff[t];
        hoff[t + 1] += hoff[t];
      }
      P->blob_bytes = boff[P->ntiles];
      P->nhalo_total = hoff[P->ntiles];
      if (P->blob_bytes / 16 >= 0x7FFFFFFF || P->nhalo_total >= 0x7FFFFFFF) /* (measured up to 32.1 GB: 110.6 M points, 771 M faces) */
        fprintf(stderr, "cfdp_plan: %.1f GB of tile blobs / %ld halo row numbers in ONE partition: a tile's offsets are 32-bit (16-byte units), "
                        "a partition holds at most 34 GB of blobs -- about 118 M points of a mesh like the F6 dual grid; cut the mesh "
                        "into more domains and give each GPU several ranks\n", (double)P->blob_bytes / 1e9, (long)P->nhalo_total);
      CFDP_ASSERT(P->blob_bytes % 16 == 0 && P->blob_bytes / 16 < 0x7FFFFFFF && P->nhalo_total < 0x7FFFFFFF);
      P->blob = cfdp_malloc((size_t)(P->blob_bytes ? P->blob_bytes : 16));
      P->halo_idx = cfdp_malloc((size_t)(P->nhalo_total ? P->nhalo_total : 1) * sizeof(int));
    }
#pragma omp parallel reduction(+ : dup_total, inc_total) reduction(| : bad) num_threads(cfdp_host_threads())
    {
      lmap fmap, hmap;
      lmap_init(&fmap, max_inc);
      lmap_init(&hmap, max_inc);
      long tg[2] = {0, 0}, tf[2] = {0, 0};
#pragma omp for schedule(dynamic, 64)
      for (int t = 0; t < P->ntiles; t++) {
        const int ts = tl->tile_first[t], te = tl->tile_first[t + 1], np = te - ts;
        cfdp_tile_desc *td = &P->tiles[t];
        lmap_reset(&fmap);
        lmap_reset(&hmap);
        int E = 0, H = 0, I = 0;
        if (pass == 0) {
          int E_foreign = 0; /* (experiment) cut faces whose normals live with the other tile */
          for (int li = 0; li < np; li++) {
            int p = tl->order[ts + li];
            for (int e = tl->xadj[p]; e < tl->xadj[p + 1]; e++) {
              int q = tl->adj_other[e];
              int f = tl->adj_face[e] & 0x7FFFFFFF, sgn = (unsigned)tl->adj_face[e] >> 31;
              int in_tile = q < nown && tl->tile_of[q] == t;
              I++;
              /* an internal face is listed by both ends and numbered at its p0 end */
              if (!in_tile || sgn == 0) {
                const int before = E;
                lmap_index(&fmap, f, &E);
                if (exp_owned && E > before && !in_tile && q < nown && tl->tile_of[q] < t) E_foreign++;
              }
              if (!in_tile) lmap_index(&hmap, q, &H);
            }
          }
          if (np + H > 65535 || E > 32767 || I >= (1 << 24)) bad = 1;
          E -= E_foreign; /* (experiment: the planes hold the owned faces only) */
          td->pstart = ts; td->npts = np;
          td->nhalo = H; td->nfaces = E; td->ninc = I;
          const long b_fn = cfdp_blob_fn_bytes(E), b_inc = cfdp_blob_inc_bytes(I), b_off = cfdp_blob_off_bytes(np);
          const long b_help = cfdp_blob_help_bytes(tile_helpers(tl, P->tile_points, ts, np, split_lists));
          td->blob_qw = (int)((b_fn + b_inc + b_off + b_help) / 16);
          boff[t + 1] = b_fn + b_inc + b_off + b_help;
          hoff[t + 1] = H;
          dup_total += E;
          inc_total += I;
          const int cls = t < P->nbtiles ? 0 : 1;
          long lg = (long)td->blob_qw * 16 + (long)(np + H) * 64;
          long lf = (long)td->blob_qw * 16 + (long)(np + H) * 80;
          if (lg > tg[cls]) tg[cls] = lg;
          if (lf > tf[cls]) tf[cls] = lf;
        } else {
          td->blob_off = (int)(boff[t] / 16);
          td->halo_off = (int)hoff[t];
          E = td->nfaces;
          const long b_fn = cfdp_blob_fn_bytes(E), b_inc = cfdp_blob_inc_bytes(td->ninc);
          unsigned char *bp = P->blob + boff[t];
          memset(bp, 0, (size_t)(boff[t + 1] - boff[t])); /* alignment padding is defined */
          int *hp = P->halo_idx + hoff[t];
          double *fn = (double *)bp;
          const long plane = cfdp_blob_plane_bytes(E) / 8; /* doubles per normal-component plane */
          uint32_t *inc = (uint32_t *)(bp + b_fn);
          uint32_t *ioff = (uint32_t *)(bp + b_fn + b_inc);
          /* numbering pass first (an internal face may be met at its p1 end before its p0 end) */
          int En = 0, Hn = 0;
          for (int li = 0; li < np; li++) {
            int p = tl->order[ts + li];
            for (int e = tl->xadj[p]; e < tl->xadj[p + 1]; e++) {
              int q = tl->adj_other[e];
              int f = tl->adj_face[e] & 0x7FFFFFFF, sgn = (unsigned)tl->adj_face[e] >> 31;
              int in_tile = q < nown && tl->tile_of[q] == t;
              const int foreign = exp_owned && !in_tile && q < nown && tl->tile_of[q] < t;
              if ((!in_tile || sgn == 0) && !foreign) lmap_index(&fmap, f, &En);
              if (!in_tile) lmap_index(&hmap, q, &Hn);
            }
          }
          int Eall = En; /* (experiment) the foreign faces are numbered behind the owned ones */
          if (exp_owned)
            for (int li = 0; li < np; li++) {
              int p = tl->order[ts + li];
              for (int e = tl->xadj[p]; e < tl->xadj[p + 1]; e++) {
                int q = tl->adj_other[e];
                if (q < nown && tl->tile_of[q] != t && tl->tile_of[q] < t) lmap_index(&fmap, tl->adj_face[e] & 0x7FFFFFFF, &Eall);
              }
            }
          /* long lists cut into chunks (cfdproxy_host.h): the table of helper lane groups behind the offsets, one entry per
           * chunk beyond a point's first, in (point, chunk) order; the scratch behind it stays zero */
          const int nh = tile_helpers(tl, P->tile_points, ts, np, split_lists);
          uint32_t *htab = (uint32_t *)(bp + b_fn + b_inc + cfdp_blob_off_bytes(np));
          int hfill = 0;
          if (nh) htab[0] = (uint32_t)nh;
          int Ic = 0;
          for (int li = 0; li < np; li++) {
            int p = tl->order[ts + li];
            const int nch = nh ? cfdp_list_chunks(tl->xadj[p + 1] - tl->xadj[p], P->tile_points) : 1;
            ioff[li] = (uint32_t)Ic | ((uint32_t)(nch - 1) << 24);
            for (int c = 1; c < nch; c++) htab[1 + hfill++] = (uint32_t)li | ((uint32_t)c << 16);
            for (int e = tl->xadj[p]; e < tl->xadj[p + 1]; e++) {
              int q = tl->adj_other[e];
              int f = tl->adj_face[e] & 0x7FFFFFFF;
              unsigned sgn = (unsigned)tl->adj_face[e] >> 31;
              int in_tile = q < nown && tl->tile_of[q] == t;
              int dummy = 0;
              int lf = lmap_index(&fmap, f, &dummy);
              /* an internal face is listed by both ends; the normal is stored once */
              if (lf < E) {
                fn[lf] = sd->fnormal[f][0];
                fn[plane + lf] = sd->fnormal[f][1];
                fn[2 * plane + lf] = sd->fnormal[f][2];
              }
              unsigned nbr;
              if (in_tile) nbr = (unsigned)(P->old2new[q] - ts);
              else {
                int hv = lmap_index(&hmap, q, &dummy);
                nbr = (unsigned)(np + hv);
                hp[hv] = P->old2new[q];
              }
              inc[Ic++] = nbr | ((unsigned)lf << 16) | (sgn << 31);
            }
          }
          ioff[np] = (uint32_t)Ic;
          if (Ic != td->ninc || En != E || Hn != td->nhalo || hfill != nh) bad = 1;
        }
      }
#pragma omp critical
      for (int c = 0; c < 2; c++) {
        if (tg[c] > lds_g[c]) lds_g[c] = tg[c];
        if (tf[c] > lds_f[c]) lds_f[c] = tf[c];
      }
      lmap_free(&fmap);
      lmap_free(&hmap);
    }
    if (bad) { free(boff); free(hoff); return 1; }
  }
  P->nfaces_dup = dup_total;
  P->ninc_total = inc_total;
  free(boff); free(hoff);
  for (int c = 0; c < 2; c++) { P->lds_grad_cls[c] = lds_g[c]; P->lds_flux_cls[c] = lds_f[c]; }
  P->lds_grad = lds_g[0] > lds_g[1] ? lds_g[0] : lds_g[1];
  P->lds_flux = lds_f[0] > lds_f[1] ? lds_f[0] : lds_f[1];
  return 0;
}

/* the host's stage 5, for a provider that has to decline a mesh (a tile too big for its scratch memory) */
int cfdp_plan_host_blobs(const solver_data *sd, const cfdp_tiling *tl, cfdp_plan *P) { return host_blobs(sd, tl, P, NULL); }

cfdp_plan *cfdp_plan_build(const solver_data *sd, const comm_data *cd, const cfdp_plan_opts *opts) {
  return cfdp_plan_build_with(sd, cd, opts, NULL);
}

/* every owned point into a tile: the boundary sheet first (when there is one to put in front), then the interior.
 * Returns the number of boundary tiles. */
static int grow_all(tiler *T, const unsigned char *is_send, const cfdp_plan_opts *o, int grow_fronts) {
  int nbtiles = 0;
  if (is_send) {
    int btp = o->tile_points / 2 < 8 ? 8 : o->tile_points / 2; /* sheets have big halos */
    {
      const char *e = getenv("CFDP_BOUNDARY_POINTS"); /* experiments: the point cap of the sheet's tiles */
      if (e && atoi(e) >= 8 && atoi(e) <= o->tile_points) btp = atoi(e);
    }
    tiler_pass(T, is_send, 1, btp);
    nbtiles = T->ntiles;
    /* the interior grows from ONE seed, layer by layer, like an un-partitioned mesh: seeding it
     * from the whole inner side of the boundary sheet makes fronts collide everywhere and leaves
     * ragged tiles (mean halo 130 instead of 113 rows, maximum 187 instead of 122 -- enough to
     * push the kernels into the next LDS capacity class and down to 2 workgroups per CU) */
    memset(T->seeded, 0, (size_t)T->nown);
    T->sq_head = T->sq_tail = 0;
  }
  if (grow_fronts > 1) tiler_pass_multifront(T, is_send, 0, o->tile_points, grow_fronts);
  else tiler_pass(T, is_send, 0, o->tile_points);
  return nbtiles;
}

/* forget the tiles grown so far (growth is run again under other budgets) */
static void tiler_restart(tiler *T, int nall) {
  for (int p = 0; p < T->nown; p++) T->tile_of[p] = -1;
  memset(T->stamp, 0, (size_t)T->nown * sizeof(int));
  memset(T->seeded, 0, (size_t)T->nown);
  memset(T->hseen, 0, (size_t)nall * sizeof(int));
  T->sq_head = T->sq_tail = 0;
  T->norder = 0;
  T->ntiles = 0;
}

/* the plan with stage 1 (point->face CSR) and / or stage 5 (tile blobs) done by `stages` -- e.g. on the device
 * (csrc/plan_kernels.hip) -- and everything else (tile growth, tile order, renumbering, pack lists) here */
cfdp_plan *cfdp_plan_build_with(const solver_data *sd, const comm_data *cd, const cfdp_plan_opts *opts,
                                const cfdp_plan_stages *stages) {
  cfdp_plan_opts o;
  if (opts) o = *opts; else cfdp_plan_default_opts(&o);
  CFDP_ASSERT(o.tile_points >= 8 && o.tile_points <= 1024);
  list_thresholds_from_env();
  const int nown = sd->nownpoints, nall = sd->nallpoints;
  CFDP_ASSERT(nown > 0 && nall >= nown);
  CFDP_ASSERT((long)sd->nfaces * 2 < 0x7FFFFFFF); /* the point->face lists are indexed by ints: 2 entries per face */
  const int has_comm = cd && cd->ndomains > 1 && cd->ncommdomains > 0;

  const int trace = getenv("CFDP_PLAN_TRACE") != NULL;
  double t_prev = cfdp_now();
#define PLAN_STAGE(name)                                                              \
  do {                                                                                \
    if (trace) { double t_ = cfdp_now(); fprintf(stderr, "[plan] %-28s %.3f s\n", name, t_ - t_prev); t_prev = t_; } \
  } while (0)
  cfdp_plan *P = cfdp_calloc(1, sizeof(*P));
  P->nown = nown; P->nall = nall; P->tile_points = o.tile_points;

  /* ---- 1. point -> incident faces (CSR over owned points, file face order): host or device ---- */
  int *xadj = NULL, *adj_face = NULL, *adj_other = NULL;
  long used = 0;
  {
    int rc = stages && stages->csr ? stages->csr(sd, &xadj, &adj_face, &adj_other, &used, stages->ctx) : -1;
    if (rc != 0) { /* no provider, or the provider failed (it has said why and freed what it had): the host stage
                      produces the same arrays, bit for bit */
      if (rc > 0) fprintf(stderr, "cfdp_plan: the device stage 'point->face CSR' failed (%d); using the host stage\n", rc);
      xadj = adj_face = adj_other = NULL;
      used = 0;
      rc = host_csr(sd, &xadj, &adj_face, &adj_other, &used, NULL);
    }
    CFDP_ASSERT(rc == 0 && xadj && adj_face && adj_other);
  }
  P->nfaces_used = used;
  PLAN_STAGE("point->face CSR");
  /* ---- 2. which owned points are sent (reference htype 2, src/rangelist.c:129-141) ---- */
  unsigned char *is_send = cfdp_calloc((size_t)nown, 1);
  int *first_partner = NULL; /* per sent point: the first partner (position in commpartner) it is sent to */
  int any_send = 0;
  if (has_comm && cd->sendindex) {
    first_partner = cfdp_malloc((size_t)(nown ? nown : 1) * sizeof(int));
    for (int p = 0; p < nown; p++) first_partner[p] = cd->ncommdomains;
    for (int i = 0; i < cd->ncommdomains; i++) {
      int k = cd->commpartner[i];
      for (int j = 0; j < cd->sendcount[k]; j++) {
        int pnt = cd->sendindex[k][j];
        CFDP_ASSERT(pnt >= 0 && pnt < nown);
        is_send[pnt] = 1;
        if (i < first_partner[pnt]) first_partner[pnt] = i;
        any_send = 1;
      }
    }
  }

  /* ---- 3. grow tiles ---- */
  tiler T;
  memset(&T, 0, sizeof T);
  T.nown = nown; T.xadj = xadj; T.adj_other = adj_other;
  T.tile_of = cfdp_malloc((size_t)nown * sizeof(int));
  for (int p = 0; p < nown; p++) T.tile_of[p] = -1;
  T.stamp = cfdp_calloc((size_t)nown, sizeof(int));
  T.seeded = cfdp_calloc((size_t)nown, 1);
  T.seedq = cfdp_malloc(((size_t)2 * nown + 2) * sizeof(int)); /* first-time seeds + re-queued rejected points */
  T.lq = cfdp_malloc((size_t)nown * sizeof(int));
  T.order = cfdp_malloc((size_t)nown * sizeof(int));
  /* bound the halo so that own + halo rows fit the smallest staging capacity of the kernels:
   * (tile_points + halo) * 5 sixteen-byte pieces <= 4 * (4 lanes * tile_points) */
  T.hseen = cfdp_calloc((size_t)nall, sizeof(int));
  /* (a point adds at most 14 halo rows on these meshes and the check precedes the addition: the cap
   * leaves room for that, 126 + 14 = 140 <= 2.2 * 64; small tiles: only the scattered ones) */
  T.halo_cap = o.tile_points * 2 - 2 < 96 ? 96 : o.tile_points * 2 - 2;
  T.split_lists = !(getenv("CFDP_SPLIT_LISTS") && atoi(getenv("CFDP_SPLIT_LISTS")) == 0);
  T.tile_points = o.tile_points;
  int two_level_forced = 0;
  {
    /* what the two capacities of the fused pass stage per tile (gg_fused_split_kernel, 4 lanes per point, 16-byte pieces
     * per thread): <5, 3, 3, 3> -- 5 blob pieces, var rows in 3 pieces at 4 per 64-byte row: a 32-KiB image at 64-point
     * tiles, five workgroups per CU -- and <6, 4, 3, 4> -- 6 blob pieces, 4 row pieces: 40 KiB, four per CU.
     * cfdp_tile_class() is the same statement for a finished tile. */
    const int block = ((o.tile_points * 4 + 63) / 64) * 64;
    T.blob_cap = (long)5 * block * 16;
    T.rows_cap = 3 * block / 4;
    T.blob_cap2 = (long)6 * block * 16;
    T.rows_cap2 = block;
    T.fill_min = o.tile_points - o.tile_points / 8; /* 7/8 full */
    const char *e = getenv("CFDP_TILE_BUDGET"); /* 0: tiles close by point count and the soft halo bound only; 1: the small
                                                   image only; 2: both levels whatever the mesh; default: decided below */
    if (e && atoi(e) == 0) T.blob_cap = T.blob_cap2 = 0;
    if (e && atoi(e) == 1) T.blob_cap2 = 0;
    two_level_forced = e && atoi(e) == 2;
    /* with the hard row budget in force the soft halo bound may go up to it: tiles of meshes with many
     * neighbours per point (15+) then fill up instead of closing at 2/3 of their points */
    if (T.blob_cap > 0 && T.rows_cap - o.tile_points > T.halo_cap) T.halo_cap = T.rows_cap - o.tile_points;
    const char *hc = getenv("CFDP_HALO_CAP"); /* experiments */
    if (hc && atoi(hc) > 0) T.halo_cap = atoi(hc);
  }
  int grow_fronts = 1; /* experiment: tiles grown at the same time (what a frontier-parallel device growth does) */
  {
    const char *e = getenv("CFDP_GROW_FRONTS");
    if (e && atoi(e) > 1) grow_fronts = atoi(e);
  }
  /* Which budgets?  The second level is for meshes whose tiles the small image closes early -- 2/3 full on an unstructured
   * graph with 15 incidences per point, a quarter of the lanes idle.  On the lattice stand-ins the small image holds full
   * tiles, and the few leftover tiles that WOULD go on under the large image put the whole launch into it (the launch-wide
   * maxima pick the kernel form: four workgroups per CU instead of five, longer row lists; measured on the rank partitions
   * of the 8-GPU configs: +1 to +3 % per iteration).  So: grow under the small image alone, look at the fill of the
   * interior tiles, and grow again with both levels only if they stayed below 7/8 full on average.  Growth is the cheap
   * part of the plan (0.02 s at 64^3). */
  const long blob_cap2 = T.blob_cap2;
  if (!two_level_forced) T.blob_cap2 = 0;
  P->nbtiles = grow_all(&T, any_send && o.boundary_first ? is_send : NULL, &o, grow_fronts);
  if (!two_level_forced && blob_cap2 > 0 && T.ntiles > P->nbtiles + 1) {
    const long interior_points = nown - T.tile_first[P->nbtiles], interior_tiles = T.ntiles - P->nbtiles;
    if (interior_points * 8 < interior_tiles * (long)o.tile_points * 7) { /* below 7/8 full: again, both levels */
      tiler_restart(&T, nall);
      T.blob_cap2 = blob_cap2;
      P->nbtiles = grow_all(&T, any_send && o.boundary_first ? is_send : NULL, &o, grow_fronts);
    }
  }
  CFDP_ASSERT(T.norder == nown);
  T.tile_first[T.ntiles] = nown;
  P->ntiles = T.ntiles;

  PLAN_STAGE("grow tiles");
  /* ---- 3b. order the tiles for L2 reuse.  Growth order sweeps the mesh in layers, so a
   * tile's neighbours (whose var rows it gathers as halo rows) can be a whole layer --
   * hundreds of tiles, several MB of stream -- apart.  Cluster the TILE graph the same way
   * the point graph was clustered ("supertiles" of o.supertile tiles) and visit the tiles
   * supertile by supertile: most neighbours are then a few dozen tiles apart and their rows
   * are still in the XCD's 4 MiB L2.  Boundary tiles stay in front (their own clustering). */
  if (o.supertile > 1 && T.ntiles > 2 * o.supertile) {
    const int nt = T.ntiles;
    int *txadj = cfdp_calloc((size_t)nt + 2, sizeof(int));
    int *tstamp = cfdp_malloc((size_t)nt * sizeof(int));
    for (int t = 0; t < nt; t++) tstamp[t] = -1;
    for (int pass = 0; pass < 2; pass++) { /* count, then fill */
      int *tadj = NULL, *tfill = NULL;
      if (pass == 1) {
        for (int t = 0; t < nt; t++) txadj[t + 1] += txadj[t];
        tadj = cfdp_malloc((size_t)(txadj[nt] ? txadj[nt] : 1) * sizeof(int));
        tfill = cfdp_malloc((size_t)nt * sizeof(int));
        memcpy(tfill, txadj, (size_t)nt * sizeof(int));
        for (int t = 0; t < nt; t++) tstamp[t] = -1;
      }
      /* (tiles are independent here: every thread keeps its own stamp array; a tile's neighbours are found in
       * the same order whatever the thread count, so the plan does not depend on it) */
#pragma omp parallel num_threads(cfdp_host_threads())
      {
        int *st = cfdp_malloc((size_t)nt * sizeof(int));
        for (int t = 0; t < nt; t++) st[t] = -1;
#pragma omp for schedule(static)
        for (int t = 0; t < nt; t++)
          for (int i = T.tile_first[t]; i < T.tile_first[t + 1]; i++) {
            int p = T.order[i];
            for (int e = xadj[p]; e < xadj[p + 1]; e++) {
              int q = adj_other[e];
              if (q >= nown) continue;
              int u = T.tile_of[q];
              if (u == t || st[u] == t) continue;
              st[u] = t;
              if (pass == 0) txadj[t + 1]++;
              else tadj[tfill[t]++] = u;
            }
          }
        free(st);
      }
      PLAN_STAGE(pass == 0 ? "  (tile graph: count)" : "  (tile graph: fill)");
      if (pass == 1) {
        tiler S;
        memset(&S, 0, sizeof S);
        S.nown = nt; S.xadj = txadj; S.adj_other = tadj;
        S.tile_of = cfdp_malloc((size_t)nt * sizeof(int));
        for (int t = 0; t < nt; t++) S.tile_of[t] = -1;
        S.stamp = cfdp_calloc((size_t)nt, sizeof(int));
        S.seeded = cfdp_calloc((size_t)nt, 1);
        S.seedq = cfdp_malloc((size_t)nt * sizeof(int));
        S.lq = cfdp_malloc((size_t)nt * sizeof(int));
        S.order = cfdp_malloc((size_t)nt * sizeof(int));
        unsigned char *cls = cfdp_malloc((size_t)nt);
        for (int t = 0; t < nt; t++) cls[t] = t < P->nbtiles ? 1 : 0;
        if (P->nbtiles) tiler_pass(&S, cls, 1, o.supertile);
        memset(S.seeded, 0, (size_t)nt);
        S.sq_head = S.sq_tail = 0;
        tiler_pass(&S, cls, 0, o.supertile);
        CFDP_ASSERT(S.norder == nt);
        PLAN_STAGE("  (supertile growth)");
        /* rebuild the point order, tile starts and tile_of for the new tile sequence */
        int *norder = cfdp_malloc((size_t)nown * sizeof(int));
        int *nfirst = cfdp_malloc((size_t)(nt + 1) * sizeof(int));
        int n = 0;
        for (int k = 0; k < nt; k++) {
          int t = S.order[k];
          nfirst[k] = n;
          for (int i = T.tile_first[t]; i < T.tile_first[t + 1]; i++) {
            norder[n] = T.order[i];
            T.tile_of[T.order[i]] = k;
            n++;
          }
        }
        nfirst[nt] = n;
        CFDP_ASSERT(n == nown);
        memcpy(T.order, norder, (size_t)nown * sizeof(int));
        memcpy(T.tile_first, nfirst, (size_t)(nt + 1) * sizeof(int));
        free(norder); free(nfirst); free(cls);
        free(S.tile_of); free(S.stamp); free(S.seeded); free(S.seedq); free(S.lq); free(S.order);
        free(S.tile_first);
        free(tadj); free(tfill);
      }
    }
    free(txadj); free(tstamp);
  }

  /* ---- 3c. boundary tiles of one partner next to each other at the front of the grid.  The boundary tile that
   * completes a partner's rows raises that partner's flag at once (per-partner notification, csrc/gg_kernels.hip
   * push_tile_done; the reference fires partner k's send when k's buffer is complete, src/threads.c:268-311): with the
   * tiles sorted by their first partner the flags go up partner by partner instead of all near the end of the sheet.
   * Stable: inside a partner's run the growth order (spatial neighbours) stays. */
  if (first_partner && P->nbtiles > 1 && !(getenv("CFDP_BTILE_ORDER") && atoi(getenv("CFDP_BTILE_ORDER")) == 0)) {
    const int nb = P->nbtiles, nk = cd->ncommdomains + 1;
    int *key = cfdp_malloc((size_t)nb * sizeof(int)), *start = cfdp_calloc((size_t)nk + 1, sizeof(int));
    for (int t = 0; t < nb; t++) {
      int k = cd->ncommdomains;
      for (int i = T.tile_first[t]; i < T.tile_first[t + 1]; i++)
        if (first_partner[T.order[i]] < k) k = first_partner[T.order[i]];
      key[t] = k;
      start[k + 1]++;
    }
    for (int k = 0; k < nk; k++) start[k + 1] += start[k];
    int *seq = cfdp_malloc((size_t)nb * sizeof(int));
    for (int t = 0; t < nb; t++) seq[start[key[t]]++] = t;
    const int nbp = T.tile_first[nb]; /* points of the boundary tiles: a prefix of the order */
    int *norder = cfdp_malloc((size_t)(nbp ? nbp : 1) * sizeof(int)), *nfirst = cfdp_malloc((size_t)(nb + 1) * sizeof(int));
    int n = 0;
    for (int k = 0; k < nb; k++) {
      const int t = seq[k];
      nfirst[k] = n;
      for (int i = T.tile_first[t]; i < T.tile_first[t + 1]; i++) norder[n++] = T.order[i];
    }
    CFDP_ASSERT(n == nbp);
    memcpy(T.order, norder, (size_t)nbp * sizeof(int));
    memcpy(T.tile_first, nfirst, (size_t)nb * sizeof(int));
    for (int k = 0; k < nb; k++)
      for (int i = T.tile_first[k]; i < T.tile_first[k + 1]; i++) T.tile_of[T.order[i]] = k;
    free(key); free(start); free(seq); free(norder); free(nfirst);
  }
  PLAN_STAGE("supertile order");
  /* ---- 3e. launch groups of the interior tiles by capacity class (cfdp_plan::group_begin).  The launch-wide maxima pick
   * the kernel form, so ONE tile beyond the fixed capacities (a hub point with hundreds of faces) used to put every
   * tile of the partition into the register-staged kernels; such tiles now go last and get a launch of their own.
   * The reference's blocking does not care what the graph looks like either (colours of <= 96 faces whatever the
   * degrees, src/rangelist.c:654-703).  The small and the large image are separated only where both sets are big: a
   * second launch costs a kernel boundary (~5 us), the fifth workgroup per CU it buys the small tiles ~1 % of
   * their time.  Stable: inside a group the supertile order stays. */
  P->ngroups = 1;
  P->group_begin[0] = P->nbtiles;
  P->group_begin[1] = T.ntiles;
  P->group_class[0] = CFDP_TILE_SMALL;
  {
    const int nt = T.ntiles;
    unsigned char *cls = cfdp_malloc((size_t)(nt ? nt : 1));
    int max_inc = 1;
    for (int t = 0; t < nt; t++) {
      int n = 0;
      for (int i = T.tile_first[t]; i < T.tile_first[t + 1]; i++) n += xadj[T.order[i] + 1] - xadj[T.order[i]];
      if (n > max_inc) max_inc = n;
    }
#pragma omp parallel num_threads(cfdp_host_threads())
    {
      lmap hmap;
      lmap_init(&hmap, max_inc);
#pragma omp for schedule(dynamic, 64)
      for (int t = 0; t < nt; t++) { /* a tile's rows and blob bytes: the counts of host_blobs, pass 0 */
        lmap_reset(&hmap);
        const int np = T.tile_first[t + 1] - T.tile_first[t];
        int H = 0;
        long I = 0, internal = 0;
        for (int i = T.tile_first[t]; i < T.tile_first[t + 1]; i++) {
          const int p = T.order[i];
          for (int e = xadj[p]; e < xadj[p + 1]; e++) {
            const int q = adj_other[e];
            I++;
            if (q < nown && T.tile_of[q] == t) internal++; /* an internal face is listed by both ends */
            else lmap_index(&hmap, q, &H);
          }
        }
        const long E = I - internal / 2;
        const cfdp_tiling tl0 = {xadj, NULL, adj_other, T.order, T.tile_first, T.tile_of};
        cls[t] = (unsigned char)cfdp_tile_class(o.tile_points, np + H,
                                                cfdp_blob_fn_bytes((int)E) + cfdp_blob_inc_bytes((int)I) + cfdp_blob_off_bytes(np) +
                                                    cfdp_blob_help_bytes(tile_helpers(&tl0, o.tile_points, T.tile_first[t], np, T.split_lists)));
      }
      lmap_free(&hmap);
    }
    long n_of[3] = {0, 0, 0};
    for (int t = P->nbtiles; t < nt; t++) n_of[cls[t]]++;
    const long split_min = getenv("CFDP_CLASS_SPLIT_MIN") ? atol(getenv("CFDP_CLASS_SPLIT_MIN")) : 8192;
    const int split_sl = n_of[CFDP_TILE_SMALL] >= split_min && n_of[CFDP_TILE_LARGE] >= split_min;
    /* group of a class */
    int grp_of[3] = {0, split_sl ? 1 : 0, 0};
    int ng = split_sl ? 2 : 1;
    if (n_of[CFDP_TILE_GENERIC] > 0 && n_of[CFDP_TILE_GENERIC] < nt - P->nbtiles) grp_of[CFDP_TILE_GENERIC] = ng++;
    if (ng > 1) {
      const int nbt = P->nbtiles, ni = nt - nbt;
      int *seq = cfdp_malloc((size_t)ni * sizeof(int));
      int n = 0;
      for (int g = 0; g < ng; g++) {
        P->group_begin[g] = nbt + n;
        for (int t = nbt; t < nt; t++)
          if (grp_of[cls[t]] == g) seq[n++] = t;
      }
      CFDP_ASSERT(n == ni);
      P->group_begin[ng] = nt;
      const int p0 = T.tile_first[nbt];
      int *norder = cfdp_malloc((size_t)(nown - p0 ? nown - p0 : 1) * sizeof(int)), *nfirst = cfdp_malloc((size_t)(ni + 1) * sizeof(int));
      n = 0;
      for (int k = 0; k < ni; k++) {
        nfirst[k] = p0 + n;
        for (int i = T.tile_first[seq[k]]; i < T.tile_first[seq[k] + 1]; i++) norder[n++] = T.order[i];
      }
      CFDP_ASSERT(p0 + n == nown);
      memcpy(T.order + p0, norder, (size_t)n * sizeof(int));
      memcpy(T.tile_first + nbt, nfirst, (size_t)ni * sizeof(int));
      unsigned char *ncls = cfdp_malloc((size_t)ni);
      for (int k = 0; k < ni; k++) ncls[k] = cls[seq[k]];
      memcpy(cls + nbt, ncls, (size_t)ni);
      for (int k = nbt; k < nt; k++)
        for (int i = T.tile_first[k]; i < T.tile_first[k + 1]; i++) T.tile_of[T.order[i]] = k;
      free(seq); free(norder); free(nfirst); free(ncls);
    }
    P->ngroups = ng;
    for (int g = 0; g < ng; g++) {
      int c = CFDP_TILE_SMALL;
      for (int t = P->group_begin[g]; t < P->group_begin[g + 1]; t++)
        if (cls[t] > c) c = cls[t];
      P->group_class[g] = c;
    }
    free(cls);
  }
  PLAN_STAGE("launch groups");
  /* ---- 3d. inside a tile: points of like degree next to each other.  A lane walks the incidence list of its point, so a
   * wave (16 points at 4 lanes each) is busy for as long as its LONGEST list takes: on a lattice every list has 14
   * entries, on an unstructured dual grid 8 to 30 and more (the reference balances its thread ranges by face degree for
   * the same reason, src/rangelist.c:320-398, min_size :354).  Sorted by descending degree (stable: growth order among
   * equals) the lists of a wave differ by one or two entries instead of ten, and a tile's SIMD time is the sum of its
   * waves' maxima instead of four times the tile's.  The values do not depend on it (a point's faces are added in file
   * order); on the lattice stand-ins only the tiles at the mesh surface change. */
  if (!(getenv("CFDP_DEGREE_SORT") && atoi(getenv("CFDP_DEGREE_SORT")) == 0)) {
    int maxdeg = 0;
    for (int p = 0; p < nown; p++)
      if (xadj[p + 1] - xadj[p] > maxdeg) maxdeg = xadj[p + 1] - xadj[p];
#pragma omp parallel num_threads(cfdp_host_threads())
    {
      int *cnt = cfdp_malloc(((size_t)maxdeg + 2) * sizeof(int));
      int *tmp = cfdp_malloc((size_t)(o.tile_points > 0 ? o.tile_points : 1) * sizeof(int));
#pragma omp for schedule(static)
      for (int t = 0; t < T.ntiles; t++) {
        const int ts = T.tile_first[t], np = T.tile_first[t + 1] - ts;
        int lo = maxdeg, hi = 0;
        for (int i = 0; i < np; i++) {
          const int d = xadj[T.order[ts + i] + 1] - xadj[T.order[ts + i]];
          if (d < lo) lo = d;
          if (d > hi) hi = d;
        }
        if (hi == lo || np > o.tile_points) continue;
        memset(cnt + lo, 0, (size_t)(hi - lo + 2) * sizeof(int)); /* counting sort, descending, stable */
        for (int i = 0; i < np; i++) cnt[xadj[T.order[ts + i] + 1] - xadj[T.order[ts + i]]]++;
        int at = 0;
        for (int d = hi; d >= lo; d--) { const int c = cnt[d]; cnt[d] = at; at += c; }
        for (int i = 0; i < np; i++) {
          const int p = T.order[ts + i];
          tmp[cnt[xadj[p + 1] - xadj[p]]++] = p;
        }
        memcpy(T.order + ts, tmp, (size_t)np * sizeof(int));
      }
      free(cnt); free(tmp);
    }
  }
  PLAN_STAGE("degree order");
  /* ---- 4. renumber: owned points tile-major; ghosts grouped by partner, message order ---- */
  P->new2old = cfdp_malloc((size_t)nall * sizeof(int));
  P->old2new = cfdp_malloc((size_t)nall * sizeof(int));
  for (int i = 0; i < nall; i++) P->old2new[i] = -1;
  for (int i = 0; i < nown; i++) { P->new2old[i] = T.order[i]; P->old2new[T.order[i]] = i; }
  int ng = nown;
  if (has_comm) {
    int np = 0;
    for (int i = 0; i < cd->ncommdomains; i++) {
      int k = cd->commpartner[i];
      if (cd->sendcount[k] > 0 || cd->recvcount[k] > 0) np++;
    }
    P->npartners = np;
    P->partner = cfdp_malloc((size_t)(np ? np : 1) * sizeof(int));
    P->send_off = cfdp_calloc((size_t)np + 1, sizeof(int));
    P->recv_off = cfdp_calloc((size_t)np + 1, sizeof(int));
    np = 0;
    for (int i = 0; i < cd->ncommdomains; i++) {
      int k = cd->commpartner[i];
      if (!(cd->sendcount[k] > 0 || cd->recvcount[k] > 0)) continue;
      P->partner[np] = k;
      P->send_off[np + 1] = P->send_off[np] + cd->sendcount[k];
      P->recv_off[np + 1] = P->recv_off[np] + cd->recvcount[k];
      for (int j = 0; j < cd->recvcount[k]; j++) {
        int old = cd->recvindex[k][j];
        CFDP_ASSERT(old >= nown && old < nall && P->old2new[old] < 0);
        P->old2new[old] = ng;
        P->new2old[ng++] = old;
      }
      np++;
    }
    P->send_idx = cfdp_malloc((size_t)(P->send_off[np] ? P->send_off[np] : 1) * sizeof(int));
    for (int s = 0; s < np; s++) {
      int k = P->partner[s];
      for (int j = 0; j < cd->sendcount[k]; j++)
        P->send_idx[P->send_off[s] + j] = P->old2new[cd->sendindex[k][j]];
    }
  }
  for (int i = nown; i < nall; i++) /* ghosts nobody sends us (or no comm tables at all) */
    if (P->old2new[i] < 0) { P->old2new[i] = ng; P->new2old[ng++] = i; }
  CFDP_ASSERT(ng == nall);

  P->vol = cfdp_malloc((size_t)nown * sizeof(double));
  P->degree = cfdp_malloc((size_t)nown * sizeof(int));
  for (int i = 0; i < nown; i++) {
    int old = P->new2old[i];
    P->vol[i] = sd->pvolume[old];
    P->degree[i] = xadj[old + 1] - xadj[old];
  }

  PLAN_STAGE("renumber");
  /* ---- 5. per-tile face copies, halo lists, incidence lists: host or device ---- */
  {
    cfdp_tiling tl = {xadj, adj_face, adj_other, T.order, T.tile_first, T.tile_of};
    int rc = stages && stages->blobs ? stages->blobs(sd, &tl, P, stages->ctx) : -1;
    if (rc != 0) { /* as above: the host stage takes over from a provider that failed */
      if (rc > 0) fprintf(stderr, "cfdp_plan: the device stage 'tile blobs' failed (%d); using the host stage\n", rc);
      free(P->tiles); free(P->blob); free(P->halo_idx);
      P->tiles = NULL; P->blob = NULL; P->halo_idx = NULL;
      P->blob_bytes = 0; P->nhalo_total = 0;
      rc = host_blobs(sd, &tl, P, NULL);
    }
    CFDP_ASSERT(rc == 0); /* tile too large for 16-bit neighbour / 15-bit face slots, or an internal error */
  }
  PLAN_STAGE("tile blobs");
#undef PLAN_STAGE
  free(T.hseen);
  free(T.tile_of); free(T.stamp); free(T.seeded); free(T.seedq); free(T.lq);
  free(T.order); free(T.tile_first);
  free(xadj); free(adj_face); free(adj_other); free(is_send); free(first_partner);
  return P;
}

void cfdp_plan_free(cfdp_plan *p) {
  if (!p) return;
  free(p->new2old); free(p->old2new); free(p->tiles); free(p->halo_idx); free(p->blob);
  free(p->vol); free(p->degree); free(p->partner); free(p->send_off); free(p->send_idx);
  free(p->recv_off);
  free(p);
}
