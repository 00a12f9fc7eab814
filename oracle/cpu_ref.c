/*
 * cpu_ref.c -- ORACLE: a CPU restatement of the CFD-Proxy hot path.   TEST INFRASTRUCTURE.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke() and
 * bench.py's `cpu_baseline` leg may load it.  The GPU path (cfd-proxy_amd/) never links
 * or calls anything in oracle/.
 *
 * It restates, in plain C + OpenMP, the algorithm CLASS of the reference so that it is
 * both a value oracle and a fair CPU baseline on machines where the reference source
 * cannot go (the GPU box only receives this repository):
 *
 *   thread domains            <- init_thread_id            reference src/rangelist.c:320-398
 *   halo classes              <- init_halo_type            reference src/rangelist.c:118-148
 *   per-thread face copies,   <- init_thread_rangelist     reference src/rangelist.c:500-764
 *     class 0..5, sort by        sort key (class, p1, p0)  reference src/util.c:130-136
 *     (class,p1,p0), colours     MAX_FACES_IN_COLOR 96     reference src/rangelist.c:654
 *   first/last point lists    <- set_{all,first,last}_points_of_color
 *                                                          reference src/points_of_color.c:26-287
 *   gradient face loop        <- private_compute_gradients_gg   src/gradients.c:25-147
 *   pseudo-flux face loop     <- private_compute_psd_flux       src/flux.c:111-190
 *   pack / unpack             <- exchange_dbl_copy_in/out       src/threads.c:791-839
 *   timed loop                <- test_solver (comm_free)        src/solver.c:42-58
 *
 * Parity pin: validated against the COMPILED reference (oracle/_ref, built from
 * /root/reference by oracle/Makefile) through the golden vectors in tests/golden/
 * (tests/test_oracle_golden.py).  The reference has no tests or golden files of its own.
 */
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#define NGRAD 7
#define NFLUX 3
#define MAX_FACES_IN_COLOR 96

typedef struct {
  int start, stop, ftype;      /* ftype: 1 writes p1, 2 writes p0, 3 writes both */
  int nfirst, nlast;
  int *first, *last;
} colour_t;

typedef struct {
  int nfaces;
  int (*fpoint)[2];
  double (*fnormal)[3];
  int ncolours;
  colour_t *colours;
  int *first_store, *last_store;
} thread_dom_t;

typedef struct oracle_plan {
  int nthreads, nown, nall, nfaces;
  double *pvolume;
  thread_dom_t *td;
} oracle_plan;

static void *xmalloc(size_t n) {
  void *p = malloc(n ? n : 1);
  if (!p) { fprintf(stderr, "oracle: out of memory\n"); exit(1); }
  return p;
}

/* ---------------------------------------------------------------- thread decomposition */
static void assign_thread_ids(int nthreads, int nfaces, const int (*fpoint)[2], int nown, int nall,
                              int *pid) {
  int *weight = xmalloc((size_t)nall * sizeof(int));
  for (int i = 0; i < nall; i++) { pid[i] = -1; weight[i] = 0; }
  for (int f = 0; f < nfaces; f++) { weight[fpoint[f][0]]++; weight[fpoint[f][1]]++; }
  const int min_size = 2 * nfaces / nthreads;
  int start = 0;
  for (int k = 0; k < nthreads; k++) {
    int acc = 0, i;
    for (i = start; i < nall; i++) {
      acc += weight[i];
      if (acc >= min_size && k < nthreads - 1) break;
    }
    int stop = i + 1 < nall ? i + 1 : nall;
    for (i = start; i < stop; i++) pid[i] = k;
    start = stop;
  }
  free(weight);
  /* a ghost end point follows the thread of the owned point it is attached to */
  for (int f = 0; f < nfaces; f++) {
    int p0 = fpoint[f][0], p1 = fpoint[f][1];
    if (p1 >= nown && pid[p0] != pid[p1]) pid[p1] = pid[p0];
    if (p0 >= nown && pid[p1] != pid[p0]) pid[p0] = pid[p1];
  }
  for (int i = 0; i < nall; i++)
    if (pid[i] < 0) { fprintf(stderr, "oracle: point %d has no thread\n", i); exit(1); }
}

typedef struct { int cls, p1, p0, idx; } sort_key;
static int key_cmp(const void *a, const void *b) {
  const sort_key *x = a, *y = b;
  if (x->cls != y->cls) return x->cls < y->cls ? -1 : 1;
  if (x->p1 != y->p1) return x->p1 < y->p1 ? -1 : 1;
  if (x->p0 != y->p0) return x->p0 < y->p0 ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

static void build_thread_domain(thread_dom_t *T, int tid, int nfaces, const int (*fpoint)[2],
                                const double (*fnormal)[3], int nown, int nall, const int *pid,
                                const int *htype) {
  memset(T, 0, sizeof(*T));
  int n = 0;
  for (int f = 0; f < nfaces; f++) {
    int p0 = fpoint[f][0], p1 = fpoint[f][1];
    if ((pid[p0] == tid && htype[p0] < 3) || (pid[p1] == tid && htype[p1] < 3)) n++;
  }
  if (!n) return;
  sort_key *keys = xmalloc((size_t)n * sizeof(sort_key));
  int *src = xmalloc((size_t)n * sizeof(int));
  n = 0;
  for (int f = 0; f < nfaces; f++) {
    int p0 = fpoint[f][0], p1 = fpoint[f][1];
    if (!((pid[p0] == tid && htype[p0] < 3) || (pid[p1] == tid && htype[p1] < 3))) continue;
    /* faces touching a sent point come first (classes 0-2), so that halo points are
     * final as early as possible; within each group: writes-p1 / writes-p0 / writes-both */
    int base = (htype[p1] == 2 || htype[p0] == 2) ? 0 : 3;
    int cls;
    if (pid[p0] != tid || htype[p0] == 3) cls = base + 0;
    else if (pid[p1] != tid || htype[p1] == 3) cls = base + 1;
    else cls = base + 2;
    keys[n].cls = cls; keys[n].p1 = p1; keys[n].p0 = p0; keys[n].idx = n;
    src[n] = f;
    n++;
  }
  qsort(keys, (size_t)n, sizeof(sort_key), key_cmp);
  T->nfaces = n;
  T->fpoint = xmalloc((size_t)n * 2 * sizeof(int));
  T->fnormal = xmalloc((size_t)n * 3 * sizeof(double));
  int *cls = xmalloc((size_t)n * sizeof(int));
  for (int i = 0; i < n; i++) {
    int f = src[keys[i].idx];
    T->fpoint[i][0] = fpoint[f][0];
    T->fpoint[i][1] = fpoint[f][1];
    memcpy(T->fnormal[i], fnormal[f], 3 * sizeof(double));
    cls[i] = keys[i].cls;
  }
  free(keys);
  free(src);

  /* colours: runs of one class, at most MAX_FACES_IN_COLOR faces */
  int nc = 1, cnt = 0;
  for (int i = 1; i < n; i++)
    if (++cnt == MAX_FACES_IN_COLOR || cls[i] != cls[i - 1]) { cnt = 0; nc++; }
  T->ncolours = nc;
  T->colours = calloc((size_t)nc, sizeof(colour_t));
  int c = 0, start = 0;
  cnt = 0;
  for (int i = 1; i < n; i++)
    if (++cnt == MAX_FACES_IN_COLOR || cls[i] != cls[i - 1]) {
      T->colours[c].start = start;
      T->colours[c].stop = i;
      start = i;
      cnt = 0;
      c++;
    }
  T->colours[c].start = start;
  T->colours[c].stop = n;
  for (c = 0; c < nc; c++) {
    int k = cls[T->colours[c].start] % 3;
    T->colours[c].ftype = k == 0 ? 1 : (k == 1 ? 2 : 3);
  }
  free(cls);

  /* first-touch (zero-init) and last-touch (finalise) lists of owned points */
  int *touch = calloc((size_t)nall, sizeof(int));
  int *seen = calloc((size_t)nall, sizeof(int));
  int nfirst_total = 0;
  for (int i = 0; i < n; i++)
    for (int e = 0; e < 2; e++) {
      int p = T->fpoint[i][e];
      if (pid[p] == tid && p < nown) { if (!touch[p]) nfirst_total++; touch[p]++; }
    }
  T->first_store = xmalloc((size_t)nfirst_total * sizeof(int));
  T->last_store = xmalloc((size_t)nfirst_total * sizeof(int));
  int nf = 0, nl = 0;
  for (c = 0; c < nc; c++) {
    colour_t *col = &T->colours[c];
    col->first = T->first_store + nf;
    col->last = T->last_store + nl;
    /* first: order of first appearance (p0 before p1 within a face) */
    for (int i = col->start; i < col->stop; i++)
      for (int e = 0; e < 2; e++) {
        int p = T->fpoint[i][e];
        if (pid[p] == tid && p < nown && seen[p] == 0) { T->first_store[nf++] = p; col->nfirst++; }
        if (pid[p] == tid && p < nown) {
          seen[p]++;
          if (seen[p] == touch[p]) { T->last_store[nl++] = p; col->nlast++; }
        }
      }
  }
  if (nf != nfirst_total || nl != nfirst_total) {
    fprintf(stderr, "oracle: first/last list mismatch (%d %d %d)\n", nf, nl, nfirst_total);
    exit(1);
  }
  free(touch);
  free(seen);
}

oracle_plan *oracle_init_threads(int nfaces, const int *fpoint_flat, const double *fnormal_flat,
                                 int nown, int nall, const double *pvolume, int nthreads,
                                 int nsendpoints, const int *sendpoints) {
  const int(*fpoint)[2] = (const int(*)[2])fpoint_flat;
  const double(*fnormal)[3] = (const double(*)[3])fnormal_flat;
  oracle_plan *P = calloc(1, sizeof(*P));
  P->nthreads = nthreads; P->nown = nown; P->nall = nall; P->nfaces = nfaces;
  P->pvolume = xmalloc((size_t)nall * sizeof(double));
  memcpy(P->pvolume, pvolume, (size_t)nall * sizeof(double));
  int *pid = xmalloc((size_t)nall * sizeof(int));
  int *htype = xmalloc((size_t)nall * sizeof(int));
  assign_thread_ids(nthreads, nfaces, fpoint, nown, nall, pid);
  for (int i = 0; i < nown; i++) htype[i] = 1;            /* owned, interior            */
  for (int j = 0; j < nsendpoints; j++) htype[sendpoints[j]] = 2; /* owned and sent     */
  for (int i = nown; i < nall; i++) htype[i] = 3;         /* ghost                      */
  P->td = calloc((size_t)nthreads, sizeof(thread_dom_t));
#pragma omp parallel for schedule(static, 1) num_threads(nthreads)
  for (int t = 0; t < nthreads; t++)
    build_thread_domain(&P->td[t], t, nfaces, fpoint, fnormal, nown, nall, pid, htype);
  free(pid);
  free(htype);
  return P;
}

void oracle_free(oracle_plan *P) {
  if (!P) return;
  for (int t = 0; t < P->nthreads; t++) {
    thread_dom_t *T = &P->td[t];
    free(T->fpoint); free(T->fnormal); free(T->colours); free(T->first_store); free(T->last_store);
  }
  free(P->td); free(P->pvolume); free(P);
}

/* invariants of reference src/eval.c:126-231: every owned point with a face is zeroed by
 * exactly one colour of exactly one thread and finalised exactly once; ghosts never.
 * Returns 0 when they hold.                                                             */
int oracle_check_invariants(const oracle_plan *P) {
  int *nfirst = calloc((size_t)P->nall, sizeof(int)), *nlast = calloc((size_t)P->nall, sizeof(int));
  int *touched = calloc((size_t)P->nall, sizeof(int));
  int bad = 0;
  for (int t = 0; t < P->nthreads; t++) {
    const thread_dom_t *T = &P->td[t];
    for (int c = 0; c < T->ncolours; c++) {
      const colour_t *col = &T->colours[c];
      if (col->stop - col->start > MAX_FACES_IN_COLOR || col->stop <= col->start) bad++;
      for (int i = 0; i < col->nfirst; i++) nfirst[col->first[i]]++;
      for (int i = 0; i < col->nlast; i++) nlast[col->last[i]]++;
      for (int f = col->start; f < col->stop; f++) {
        if (col->ftype & 2) touched[T->fpoint[f][0]] = 1;
        if (col->ftype & 1) touched[T->fpoint[f][1]] = 1;
      }
    }
  }
  for (int p = 0; p < P->nall; p++) {
    if (p >= P->nown) bad += (nfirst[p] != 0) + (nlast[p] != 0) + (touched[p] != 0);
    else if (touched[p]) bad += (nfirst[p] != 1) + (nlast[p] != 1);
    else bad += (nfirst[p] != 0) + (nlast[p] != 0);
  }
  free(nfirst); free(nlast); free(touched);
  return bad;
}

int oracle_total_colours(const oracle_plan *P) {
  int n = 0;
  for (int t = 0; t < P->nthreads; t++) n += P->td[t].ncolours;
  return n;
}
long oracle_total_thread_faces(const oracle_plan *P) {
  long n = 0;
  for (int t = 0; t < P->nthreads; t++) n += P->td[t].nfaces;
  return n;
}

/* ------------------------------------------------------------------ the two face loops */
static void gradient_colour(const thread_dom_t *T, const colour_t *col, const double *pvolume,
                            const double (*var)[NGRAD], double (*grad)[NGRAD][3]) {
  for (int i = 0; i < col->nfirst; i++) {
    int pnt = col->first[i];
    for (int eq = 0; eq < NGRAD; eq++) grad[pnt][eq][0] = grad[pnt][eq][1] = grad[pnt][eq][2] = 0.0;
  }
  const int w0 = (col->ftype & 2) != 0, w1 = (col->ftype & 1) != 0;
  for (int f = col->start; f < col->stop; f++) {
    const int p0 = T->fpoint[f][0], p1 = T->fpoint[f][1];
    const double anx = T->fnormal[f][0], any = T->fnormal[f][1], anz = T->fnormal[f][2];
    for (int eq = 0; eq < NGRAD; eq++) {
      const double val = 0.5 * (var[p0][eq] + var[p1][eq]);
      const double vx = anx * val, vy = any * val, vz = anz * val;
      if (w0) { grad[p0][eq][0] += vx; grad[p0][eq][1] += vy; grad[p0][eq][2] += vz; }
      if (w1) { grad[p1][eq][0] -= vx; grad[p1][eq][1] -= vy; grad[p1][eq][2] -= vz; }
    }
  }
  for (int i = 0; i < col->nlast; i++) {
    int pnt = col->last[i];
    const double tmp = 1 / pvolume[pnt];
    for (int eq = 0; eq < NGRAD; eq++) {
      grad[pnt][eq][0] *= tmp; grad[pnt][eq][1] *= tmp; grad[pnt][eq][2] *= tmp;
    }
  }
}

/* mode 0: consistent (an end point is written iff the gradient loop writes it);
 * mode 1: the reference as written -- flux.c tests the class with the README numbering
 *         (`ftype != 3` adds to p0, `ftype != 2` subtracts from p1, src/flux.c:177-188)
 *         while classes are numbered 1 = p1, 2 = p0, 3 = both (src/rangelist.c:719-736).
 *         Well defined for 1 thread only (it writes foreign/ghost points otherwise).      */
static void flux_colour(const thread_dom_t *T, const colour_t *col, int mode,
                        const double (*grad)[NGRAD][3], double (*flux)[NFLUX]) {
  for (int i = 0; i < col->nfirst; i++) {
    int pnt = col->first[i];
    flux[pnt][0] = flux[pnt][1] = flux[pnt][2] = 0.0;
  }
  int w0, w1;
  if (mode == 1) { w0 = col->ftype != 3; w1 = col->ftype != 2; }
  else { w0 = (col->ftype & 2) != 0; w1 = (col->ftype & 1) != 0; }
  const double mue_eff = 1.0, lambda = -2.0 / 3.0 * mue_eff;
  for (int f = col->start; f < col->stop; f++) {
    const int p0 = T->fpoint[f][0], p1 = T->fpoint[f][1];
    const double nx = T->fnormal[f][0], ny = T->fnormal[f][1], nz = T->fnormal[f][2];
    double d[3][3];
    for (int v = 0; v < 3; v++)
      for (int k = 0; k < 3; k++) d[v][k] = 0.5 * (grad[p0][v][k] + grad[p1][v][k]);
    const double sts_xx = lambda * (d[1][1] + d[2][2] - 2.0 * d[0][0]);
    const double sts_yy = lambda * (d[0][0] + d[2][2] - 2.0 * d[1][1]);
    const double sts_zz = lambda * (d[0][0] + d[1][1] - 2.0 * d[2][2]);
    const double sts_xy = mue_eff * (d[0][1] + d[1][0]);
    const double sts_xz = mue_eff * (d[0][2] + d[2][0]);
    const double sts_yz = mue_eff * (d[1][2] + d[2][1]);
    const double fx = -(sts_xx * nx + sts_xy * ny + sts_xz * nz);
    const double fy = -(sts_xy * nx + sts_yy * ny + sts_yz * nz);
    const double fz = -(sts_xz * nx + sts_yz * ny + sts_zz * nz);
    if (w0) { flux[p0][0] += fx; flux[p0][1] += fy; flux[p0][2] += fz; }
    if (w1) { flux[p1][0] -= fx; flux[p1][1] -= fy; flux[p1][2] -= fz; }
  }
}

void oracle_gradients(const oracle_plan *P, const double *var, double *grad) {
#pragma omp parallel num_threads(P->nthreads)
  {
    const thread_dom_t *T = &P->td[omp_get_thread_num()];
    for (int c = 0; c < T->ncolours; c++)
      gradient_colour(T, &T->colours[c], P->pvolume, (const double(*)[NGRAD])var,
                      (double(*)[NGRAD][3])grad);
  }
}

void oracle_flux(const oracle_plan *P, const double *grad, double *flux, int mode) {
#pragma omp parallel num_threads(P->nthreads)
  {
    const thread_dom_t *T = &P->td[omp_get_thread_num()];
    for (int c = 0; c < T->ncolours; c++)
      flux_colour(T, &T->colours[c], mode, (const double(*)[NGRAD][3])grad, (double(*)[NFLUX])flux);
  }
}

/* halo pack / unpack: 21 doubles per point, message element j <-> index[j] */
void oracle_pack(const int *sendindex, int count, const double *data, int dim2, double *sbuf) {
  for (int j = 0; j < count; j++)
    memcpy(&sbuf[(size_t)dim2 * j], &data[(size_t)dim2 * sendindex[j]], (size_t)dim2 * sizeof(double));
}
void oracle_unpack(const int *recvindex, int count, double *data, int dim2, const double *rbuf) {
  for (int j = 0; j < count; j++)
    memcpy(&data[(size_t)dim2 * recvindex[j]], &rbuf[(size_t)dim2 * j], (size_t)dim2 * sizeof(double));
}

static double now(void) {
  struct timeval tp;
  gettimeofday(&tp, NULL);
  return (double)tp.tv_sec + (double)tp.tv_usec * 1.e-6;
}

/* the comm_free timing loop of the harness: all threads run `niter` iterations of
 * gradients (+ flux) with one barrier per iteration.  Returns seconds; with_flux = 0
 * times the gradient loop alone.                                                         */
double oracle_timed_iterations(const oracle_plan *P, const double *var, double *grad, double *flux,
                               int niter, int with_flux, int flux_mode) {
  double t = -now();
#pragma omp parallel num_threads(P->nthreads)
  {
    const thread_dom_t *T = &P->td[omp_get_thread_num()];
    for (int it = 0; it < niter; it++) {
      for (int c = 0; c < T->ncolours; c++)
        gradient_colour(T, &T->colours[c], P->pvolume, (const double(*)[NGRAD])var,
                        (double(*)[NGRAD][3])grad);
      if (with_flux) {
        /* the flux of a face reads the finished gradient of BOTH ends, which may belong
         * to another thread: the reference orders this with stage counters
         * (wait_for_local_neighbours, src/rangelist.c:779-795); a barrier is the safe
         * restatement                                                                   */
#pragma omp barrier
        for (int c = 0; c < T->ncolours; c++)
          flux_colour(T, &T->colours[c], flux_mode, (const double(*)[NGRAD][3])grad,
                      (double(*)[NFLUX])flux);
      }
#pragma omp barrier
    }
  }
  t += now();
  return t;
}

int oracle_max_threads(void) { return omp_get_max_threads(); }
