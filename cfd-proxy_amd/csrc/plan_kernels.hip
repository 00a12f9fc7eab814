// plan_kernels.hip -- the preprocessing of init_threads() on the device.
//
// The reference prepares its face loop on the host, once per run: every thread copies the faces touching its
// points, classifies and sorts them, cuts them into colours and builds first/last-touch and per-colour
// send/receive lists (init_thread_rangelist, src/rangelist.c:500-764; gather_sendcount/recvcount,
// src/thread_comm.c:27-432) -- O(F log F), 1.9 s at 1.8 M faces.  The GPU analogue is the "plan"
// (host/tiling.c).  Its two heavy stages are data-parallel and run here as HIP kernels:
//
//   stage 1  point -> incident faces, CSR over the owned points in file face order
//            (count with atomics -> exclusive scan -> fill with atomics -> sort every point's short list by
//            face number, which restores the file order the host stage produces by streaming)
//   stage 5  per-tile blobs: one workgroup per tile numbers the tile's faces and halo points by FIRST TOUCH
//            along the tile's walk (its points in tile order, each point's faces in file order) -- a
//            sequential notion, done in parallel: a face is numbered at exactly one incidence (a cut face
//            at its only one, an internal face at its p0 end), so its number is a prefix sum of flags over
//            the walk; a halo point is numbered at the smallest walk position that names it (LDS hash
//            table with atomicMin), again ranked by a prefix sum
//
// Tile growth (a sequential BFS), the tile order, the renumbering and the pack lists stay on the host
// (host/tiling.c, cfdp_plan_build_with).  The plan built this way is bit-identical to the host's
// (tests/test_gpu_parity.py::test_device_built_plan_equals_host_plan).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cfdproxy_hip.h"
#include "gg_kernels.h"

int cfdp_set_error(const char *fmt, ...);  // gpu_abi.hip

#define PK_TRY(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return cfdp_set_error("%s failed: %s [%s:%d]", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

namespace {

// ------------------------------------------------------------------------------- stage 1
__global__ void k_degree(const int2 *__restrict__ fpoint, int nf, int nown, int *__restrict__ deg,
                         unsigned long long *__restrict__ used) {
  unsigned long long mine = 0;
  for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += gridDim.x * blockDim.x) {
    const int2 ab = fpoint[f];
    if (ab.x < nown) atomicAdd(&deg[ab.x], 1);
    if (ab.y < nown) atomicAdd(&deg[ab.y], 1);
    mine += (ab.x < nown || ab.y < nown) ? 1 : 0;
  }
  for (int m = 32; m >= 1; m >>= 1) mine += __shfl_xor(mine, m, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(used, mine);
}

// exclusive scan of n ints in three small kernels (block sums -> scan of block sums -> add)
constexpr int SCAN_B = 1024;
__device__ int block_exclusive_scan(int v, int *tmp /*[blockDim.x]*/, int *total) {
  const int tid = threadIdx.x;
  tmp[tid] = v;
  __syncthreads();
  for (int off = 1; off < (int)blockDim.x; off <<= 1) {
    const int add = tid >= off ? tmp[tid - off] : 0;
    __syncthreads();
    tmp[tid] += add;
    __syncthreads();
  }
  const int incl = tmp[tid];
  if (total) *total = tmp[blockDim.x - 1];
  __syncthreads();
  return incl - v;
}
__global__ __launch_bounds__(SCAN_B) void k_scan_blocks(const int *__restrict__ in, int n, int *__restrict__ out,
                                                        int *__restrict__ block_sum) {
  __shared__ int tmp[SCAN_B];
  const int i = blockIdx.x * SCAN_B + threadIdx.x;
  const int v = i < n ? in[i] : 0;
  int total = 0;
  const int ex = block_exclusive_scan(v, tmp, &total);
  if (i < n) out[i] = ex;
  if (threadIdx.x == 0) block_sum[blockIdx.x] = total;
}
__global__ void k_scan_top(int *block_sum, int nb) {  // nb is small (n / 1024): one thread
  if (threadIdx.x || blockIdx.x) return;
  int run = 0;
  for (int b = 0; b < nb; b++) {
    const int v = block_sum[b];
    block_sum[b] = run;
    run += v;
  }
  block_sum[nb] = run;
}
__global__ __launch_bounds__(SCAN_B) void k_scan_add(int *__restrict__ out, int n, const int *__restrict__ block_sum, int nb) {
  const int i = blockIdx.x * SCAN_B + threadIdx.x;
  if (i < n) out[i] += block_sum[blockIdx.x];
  if (i == 0) out[n] = block_sum[nb];
}

__global__ void k_fill(const int2 *__restrict__ fpoint, int nf, int nown, const int *__restrict__ xadj,
                       int *__restrict__ cursor, int *__restrict__ adj_face, int *__restrict__ adj_other) {
  for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += gridDim.x * blockDim.x) {
    const int2 ab = fpoint[f];
    if (ab.x < nown) {
      const int at = xadj[ab.x] + atomicAdd(&cursor[ab.x], 1);
      adj_face[at] = f;
      adj_other[at] = ab.y;
    }
    if (ab.y < nown) {
      const int at = xadj[ab.y] + atomicAdd(&cursor[ab.y], 1);
      adj_face[at] = (int)((unsigned)f | 0x80000000u);
      adj_other[at] = ab.x;
    }
  }
}

// file order of a point's list = ascending (face, p1-flag): the host stage streams the faces in order and lists
// the p0 end before the p1 end
__device__ __forceinline__ unsigned long long adj_key(int face_word) {
  return ((unsigned long long)((unsigned)face_word & 0x7FFFFFFFu) << 1) | ((unsigned)face_word >> 31);
}
constexpr int SORT_SMALL = 48;
__global__ void k_sort_small(int nown, const int *__restrict__ xadj, int *__restrict__ adj_face, int *__restrict__ adj_other) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nown) return;
  const int s = xadj[p], n = xadj[p + 1] - s;
  if (n < 2 || n > SORT_SMALL) return;
  for (int i = 1; i < n; i++) {  // insertion sort: the lists are ~14 long
    const int fw = adj_face[s + i], ow = adj_other[s + i];
    const unsigned long long k = adj_key(fw);
    int j = i - 1;
    while (j >= 0 && adj_key(adj_face[s + j]) > k) {
      adj_face[s + j + 1] = adj_face[s + j];
      adj_other[s + j + 1] = adj_other[s + j];
      j--;
    }
    adj_face[s + j + 1] = fw;
    adj_other[s + j + 1] = ow;
  }
}
// hub points: one workgroup per point, rank sort through a scratch copy
__global__ void k_sort_big(const int *__restrict__ big, const int *__restrict__ xadj, int *__restrict__ adj_face,
                           int *__restrict__ adj_other, int *__restrict__ scratch_face, int *__restrict__ scratch_other) {
  const int p = big[blockIdx.x];
  const int s = xadj[p], n = xadj[p + 1] - s;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    scratch_face[s + i] = adj_face[s + i];
    scratch_other[s + i] = adj_other[s + i];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const unsigned long long k = adj_key(scratch_face[s + i]);
    int rank = 0;
    for (int j = 0; j < n; j++) rank += adj_key(scratch_face[s + j]) < k ? 1 : 0;
    adj_face[s + rank] = scratch_face[s + i];
    adj_other[s + rank] = scratch_other[s + i];
  }
}

// ------------------------------------------------------------------------------- stage 5
struct blob_args {
  const int *xadj, *adj_face, *adj_other;  // stage 1 (file ids)
  const int *order, *tile_first, *tile_of, *old2new;
  const double *fnormal;                   // [nf][3]
  int nown;
  // pass A out / pass B in
  int *cntE, *cntH, *cntI;                 // [ntiles]
  const long *boff;                        // [ntiles+1] blob byte offsets (pass B)
  const long *hoff;                        // [ntiles+1] halo entry offsets (pass B)
  unsigned char *blob;
  int *halo_idx;
  int max_inc, cap;                        // LDS sizing: incidences per tile, hash capacity (power of two)
  int *bad;
  // long incidence lists cut into chunks (cfdproxy_host.h): helper lane groups per tile (0: no list of the tile is cut) and
  // the plan's tile_points (the cap on chunks per list)
  const int *nhelp;                        // [ntiles] (pass B)
  int tile_points, long_list, list_chunk;  // (cfdp_long_list(), cfdp_list_chunk())
};

// cfdp_list_chunks (cfdproxy_host.h), for the device
__device__ __forceinline__ int dev_list_chunks(int deg, int tile_points, int long_list, int list_chunk) {
  if (deg <= long_list) return 1;
  int cap = tile_points / 4 < CFDP_MAX_CHUNKS ? tile_points / 4 : CFDP_MAX_CHUNKS;
  if (cap < 1) cap = 1;
  const int n = (deg + list_chunk - 1) / list_chunk;
  return n > cap ? cap : n;
}

constexpr int BLOB_T = 256;
constexpr int EMPTY_KEY = -1;

// Dynamic LDS: ioff[np_max + 2] | lf[max_inc] | slot[max_inc] | hkey[cap] | hmin[cap] | hval[cap] | scanA[BLOB_T] | scanB[BLOB_T]
template <bool FILL>
__global__ __launch_bounds__(BLOB_T) void k_tile_blobs(blob_args a, int np_max) {
  extern __shared__ int lds[];
  int *ioff = lds;                       // [np + 1] walk position of each point's first incidence
  int *lf = ioff + np_max + 2;           // [I] local face number (numbering incidences) / scratch
  int *slot = lf + a.max_inc;            // [I] hash slot of the incidence's halo point, or -1
  int *hkey = slot + a.max_inc;          // [cap]
  int *hmin = hkey + a.cap;              // [cap] smallest walk position naming the key
  int *hval = hmin + a.cap;              // [cap] halo number of the key
  int *scanA = hval + a.cap;             // [BLOB_T]
  int *scanB = scanA + BLOB_T;           // [BLOB_T]
  const int t = blockIdx.x, tid = threadIdx.x;
  const int ts = a.tile_first[t], np = a.tile_first[t + 1] - ts;

  // walk offsets: exclusive scan of the degrees of the tile's points (np <= 1024: 4 per thread)
  {
    int d[4], sum = 0;
    for (int r = 0; r < 4; r++) {
      const int li = tid * 4 + r;
      d[r] = 0;
      if (li < np) {
        const int p = a.order[ts + li];
        d[r] = a.xadj[p + 1] - a.xadj[p];
      }
      sum += d[r];
    }
    int total = 0;
    int ex = block_exclusive_scan(sum, scanA, &total);
    for (int r = 0; r < 4; r++) {
      const int li = tid * 4 + r;
      if (li < np) ioff[li] = ex;
      ex += d[r];
    }
    if (tid == 0) ioff[np] = total;
  }
  for (int i = tid; i < a.cap; i += BLOB_T) { hkey[i] = EMPTY_KEY; hmin[i] = 0x7FFFFFFF; }
  __syncthreads();
  const int I = ioff[np];
  if (I > a.max_inc) {  // cannot happen: max_inc is the maximum over the tiles
    if (tid == 0) *a.bad = 1;
    return;
  }
  // contiguous chunk of the walk per thread: ranks along the walk need only a scan of per-thread counts
  const int per = (I + BLOB_T - 1) / BLOB_T, k0 = tid * per < I ? tid * per : I, k1 = k0 + per < I ? k0 + per : I;
  auto point_of = [&](int k) {  // tile-local point whose list holds walk position k
    int lo = 0, hi = np - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (ioff[mid] <= k) lo = mid; else hi = mid - 1;
    }
    return lo;
  };
  // pass 1 over my chunk: numbering flags of faces, hash inserts of halo points
  int nE = 0;
  {
    int li = k0 < k1 ? point_of(k0) : 0;
    for (int k = k0; k < k1; k++) {
      while (k >= ioff[li + 1]) li++;
      const int p = a.order[ts + li];
      const int e = a.xadj[p] + (k - ioff[li]);
      const int q = a.adj_other[e];
      const unsigned sgn = (unsigned)a.adj_face[e] >> 31;
      const bool in_tile = q < a.nown && a.tile_of[q] == t;
      const bool numbering = !in_tile || sgn == 0;  // an internal face is numbered at its p0 end
      lf[k] = numbering ? 1 : 0;
      nE += numbering ? 1 : 0;
      int sl = -1;
      if (!in_tile) {
        unsigned h = ((unsigned)q * 2654435761u) >> 7;
        for (;;) {
          h &= (unsigned)(a.cap - 1);
          const int old = atomicCAS(&hkey[h], EMPTY_KEY, q);
          if (old == EMPTY_KEY || old == q) break;
          h++;
        }
        sl = (int)h;
        atomicMin(&hmin[sl], k);
      }
      slot[k] = sl;
    }
  }
  __syncthreads();
  int nH = 0;
  for (int k = k0; k < k1; k++) nH += (slot[k] >= 0 && hmin[slot[k]] == k) ? 1 : 0;
  int totE = 0, totH = 0;
  int exE = block_exclusive_scan(nE, scanA, &totE);
  int exH = block_exclusive_scan(nH, scanB, &totH);
  if (!FILL) {
    if (tid == 0) {
      a.cntE[t] = totE;
      a.cntH[t] = totH;
      a.cntI[t] = I;
      if (np + totH > 65535 || totE > 32767) *a.bad = 1;  // 16-bit neighbour / 15-bit face slots
    }
    return;
  }
  // pass 2: ranks along the walk
  for (int k = k0; k < k1; k++) {
    if (lf[k]) lf[k] = exE++; else lf[k] = -1;
    if (slot[k] >= 0 && hmin[slot[k]] == k) hval[slot[k]] = exH++;
  }
  __syncthreads();
  // pass 3: write the blob
  const int E = totE;
  unsigned char *bp = a.blob + a.boff[t];
  const long plane = (((long)E * 8 + 15) & ~15L) / 8;
  const long b_fn = 3 * plane * 8, b_inc = ((long)I * 4 + 15) & ~15L;
  double *fn = reinterpret_cast<double *>(bp);
  uint32_t *inc = reinterpret_cast<uint32_t *>(bp + b_fn);
  uint32_t *io = reinterpret_cast<uint32_t *>(bp + b_fn + b_inc);
  int *hp = a.halo_idx + a.hoff[t];
  const int nh = a.nhelp[t];  // helper lane groups of this tile: > 0 = its long lists are cut into chunks
  for (int li = tid; li <= np; li += BLOB_T) {
    const int nch = nh && li < np ? dev_list_chunks(ioff[li + 1] - ioff[li], a.tile_points, a.long_list, a.list_chunk) : 1;
    io[li] = (uint32_t)ioff[li] | ((uint32_t)(nch - 1) << 24);
  }
  if (nh && tid == 0) {  // the helper table behind the offsets, in (point, chunk) order (a few entries in a tile in a hundred)
    uint32_t *htab = reinterpret_cast<uint32_t *>(bp + b_fn + b_inc + ((((long)np + 1) * 4 + 15) & ~15L));
    htab[0] = (uint32_t)nh;
    int fill = 0;
    for (int li = 0; li < np; li++) {
      const int nch = dev_list_chunks(ioff[li + 1] - ioff[li], a.tile_points, a.long_list, a.list_chunk);
      for (int c = 1; c < nch; c++) htab[1 + fill++] = (uint32_t)li | ((uint32_t)c << 16);
    }
    if (fill != nh) atomicExch(a.bad, 1);
  }
  {
    int li = k0 < k1 ? point_of(k0) : 0;
    for (int k = k0; k < k1; k++) {
      while (k >= ioff[li + 1]) li++;
      const int p = a.order[ts + li];
      const int e = a.xadj[p] + (k - ioff[li]);
      const int q = a.adj_other[e];
      const int fw = a.adj_face[e];
      const int f = fw & 0x7FFFFFFF;
      const unsigned sgn = (unsigned)fw >> 31;
      int face = lf[k];
      unsigned nbr;
      if (slot[k] >= 0) {  // halo point
        const int hv = hval[slot[k]];
        nbr = (unsigned)(np + hv);
        if (hmin[slot[k]] == k) hp[hv] = a.old2new[q];
      } else {
        const int lq = a.old2new[q] - ts;  // points are renumbered tile-major: tile-local index of q
        nbr = (unsigned)lq;
        if (face < 0) {  // internal face met at its p1 end: numbered at q's incidence of the same face (sign 0)
          int lo = a.xadj[q], hi = a.xadj[q + 1] - 1;
          while (lo < hi) {  // q's list is sorted by (face, sign)
            const int mid = lo + ((hi - lo) >> 1);  // (lo + hi overflows an int once the mesh has more than 2^29 faces)
            if (adj_key(a.adj_face[mid]) < ((unsigned long long)(unsigned)f << 1)) lo = mid + 1; else hi = mid;
          }
          face = lf[ioff[lq] + (lo - a.xadj[q])];
        }
      }
      if (lf[k] >= 0) {  // the numbering incidence stores the normal (once per tile face)
        fn[face] = a.fnormal[(size_t)f * 3 + 0];
        fn[plane + face] = a.fnormal[(size_t)f * 3 + 1];
        fn[2 * plane + face] = a.fnormal[(size_t)f * 3 + 2];
      }
      inc[k] = nbr | ((unsigned)face << 16) | (sgn << 31);
    }
  }
}

struct stage_ctx {
  int device = 0;
  int nown = 0, nf = 0, nadj = 0;
  int *d_xadj = nullptr, *d_adj_face = nullptr, *d_adj_other = nullptr;
  double seconds[2] = {0.0, 0.0};
  bool blobs_on_host = false;  // stage 5 fell back (a tile too big for the LDS hash)
  ~stage_ctx() {
    (void)hipFree(d_xadj);
    (void)hipFree(d_adj_face);
    (void)hipFree(d_adj_other);
  }
};

double wall() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

template <typename T> struct host_buf {  // malloc'd; freed unless released
  T *p = nullptr;
  ~host_buf() { free(p); }
  T *release() { T *q = p; p = nullptr; return q; }
};

template <typename T> struct dev_buf {  // freed on every return path
  T *p = nullptr;
  ~dev_buf() { (void)hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc(&p, sizeof(T) * (n ? n : 1)); }
};

int exclusive_scan(const int *d_in, int n, int *d_out /*[n+1]*/) {
  const int nb = (n + SCAN_B - 1) / SCAN_B;
  dev_buf<int> sums;
  PK_TRY(sums.alloc((size_t)nb + 1));
  hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(SCAN_B), 0, 0, d_in, n, d_out, sums.p);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(64), 0, 0, sums.p, nb);
  hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(SCAN_B), 0, 0, d_out, n, sums.p, nb);
  PK_TRY(hipGetLastError());
  PK_TRY(hipDeviceSynchronize());
  return 0;
}

// CFDP_PLAN_FAIL_STAGE=1|5 (tests): the stage fails the way an out-of-memory hipMalloc would
bool fail_injected(int stage) {
  const char *e = cfdp_experiment_getenv("CFDP_PLAN_FAIL_STAGE");  // honoured only with CFDP_EXPERIMENTS=1
  return e && atoi(e) == stage;
}

int stage_csr(const solver_data *sd, int **xadj_out, int **adj_face_out, int **adj_other_out, long *used_out, void *vctx) {
  stage_ctx *c = static_cast<stage_ctx *>(vctx);
  const double t0 = wall();
  if (fail_injected(1)) return cfdp_set_error("hipMalloc failed: out of memory (injected by CFDP_PLAN_FAIL_STAGE)");
  const int nown = sd->nownpoints, nf = sd->nfaces;
  c->nown = nown;
  c->nf = nf;
  PK_TRY(hipSetDevice(c->device));
  dev_buf<int2> fpoint;
  dev_buf<int> deg, cursor;
  dev_buf<unsigned long long> used;
  PK_TRY(fpoint.alloc((size_t)nf));
  PK_TRY(deg.alloc((size_t)nown));
  PK_TRY(cursor.alloc((size_t)nown));
  PK_TRY(used.alloc(1));
  PK_TRY(hipMalloc(&c->d_xadj, sizeof(int) * ((size_t)nown + 2)));
  PK_TRY(hipMemcpy(fpoint.p, &sd->fpoint[0][0], sizeof(int2) * (size_t)nf, hipMemcpyHostToDevice));
  PK_TRY(hipMemset(deg.p, 0, sizeof(int) * (size_t)nown));
  PK_TRY(hipMemset(cursor.p, 0, sizeof(int) * (size_t)nown));
  PK_TRY(hipMemset(used.p, 0, sizeof(unsigned long long)));
  const int blocks = 4096;
  hipLaunchKernelGGL(k_degree, dim3(blocks), dim3(256), 0, 0, fpoint.p, nf, nown, deg.p, used.p);
  PK_TRY(hipGetLastError());
  if (exclusive_scan(deg.p, nown, c->d_xadj)) return 1;  // (the scan has set the error text)
  host_buf<int> hx, hf, ho;  // freed on every failing return path, released to the caller on success
  int *xadj = hx.p = static_cast<int *>(calloc((size_t)nown + 2, sizeof(int)));
  if (!xadj) return cfdp_set_error("out of memory");
  PK_TRY(hipMemcpy(xadj, c->d_xadj, sizeof(int) * ((size_t)nown + 1), hipMemcpyDeviceToHost));
  const int nadj = xadj[nown];
  c->nadj = nadj;
  PK_TRY(hipMalloc(&c->d_adj_face, sizeof(int) * (size_t)(nadj ? nadj : 1)));
  PK_TRY(hipMalloc(&c->d_adj_other, sizeof(int) * (size_t)(nadj ? nadj : 1)));
  hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, 0, fpoint.p, nf, nown, c->d_xadj, cursor.p, c->d_adj_face, c->d_adj_other);
  hipLaunchKernelGGL(k_sort_small, dim3((nown + 255) / 256), dim3(256), 0, 0, nown, c->d_xadj, c->d_adj_face, c->d_adj_other);
  PK_TRY(hipGetLastError());
  std::vector<int> big;
  for (int p = 0; p < nown; p++)
    if (xadj[p + 1] - xadj[p] > SORT_SMALL) big.push_back(p);
  if (!big.empty()) {
    dev_buf<int> d_big, sf, so;
    PK_TRY(d_big.alloc(big.size()));
    PK_TRY(sf.alloc((size_t)nadj));
    PK_TRY(so.alloc((size_t)nadj));
    PK_TRY(hipMemcpy(d_big.p, big.data(), sizeof(int) * big.size(), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_sort_big, dim3((unsigned)big.size()), dim3(256), 0, 0, d_big.p, c->d_xadj, c->d_adj_face, c->d_adj_other,
                       sf.p, so.p);
    PK_TRY(hipGetLastError());
    PK_TRY(hipDeviceSynchronize());
  }
  int *adj_face = hf.p = static_cast<int *>(malloc(sizeof(int) * (size_t)(nadj ? nadj : 1)));
  int *adj_other = ho.p = static_cast<int *>(malloc(sizeof(int) * (size_t)(nadj ? nadj : 1)));
  if (!adj_face || !adj_other) return cfdp_set_error("out of memory");
  PK_TRY(hipMemcpy(adj_face, c->d_adj_face, sizeof(int) * (size_t)nadj, hipMemcpyDeviceToHost));
  PK_TRY(hipMemcpy(adj_other, c->d_adj_other, sizeof(int) * (size_t)nadj, hipMemcpyDeviceToHost));
  unsigned long long u = 0;
  PK_TRY(hipMemcpy(&u, used.p, sizeof u, hipMemcpyDeviceToHost));
  *xadj_out = hx.release();
  *adj_face_out = hf.release();
  *adj_other_out = ho.release();
  *used_out = (long)u;
  c->seconds[0] = wall() - t0;
  return 0;
}

int stage_blobs(const solver_data *sd, const cfdp_tiling *tl, cfdp_plan *P, void *vctx) {
  stage_ctx *c = static_cast<stage_ctx *>(vctx);
  const double t0 = wall();
  const int nown = P->nown, nall = P->nall, nt = P->ntiles;
  PK_TRY(hipSetDevice(c->device));
  // LDS sizing from the tiling: incidences and points per tile
  int max_inc = 1, np_max = 1;
  for (int t = 0; t < nt; t++) {
    int n = 0;
    for (int i = tl->tile_first[t]; i < tl->tile_first[t + 1]; i++) n += tl->xadj[tl->order[i] + 1] - tl->xadj[tl->order[i]];
    if (n > max_inc) max_inc = n;
    if (tl->tile_first[t + 1] - tl->tile_first[t] > np_max) np_max = tl->tile_first[t + 1] - tl->tile_first[t];
  }
  int cap = 64;
  while (cap < 2 * max_inc) cap *= 2;
  const size_t lds = sizeof(int) * ((size_t)np_max + 2 + 2 * (size_t)max_inc + 3 * (size_t)cap + 2 * BLOB_T);
  if (fail_injected(5)) {
    P->tiles = static_cast<cfdp_tile_desc *>(calloc((size_t)nt, sizeof(cfdp_tile_desc)));  // a partial output, as a late failure leaves
    return cfdp_set_error("hipMalloc failed: out of memory (injected by CFDP_PLAN_FAIL_STAGE)");
  }
  if (lds > 160 * 1024 || np_max > 4 * BLOB_T) return 2;  // a tile too big for the LDS hash: the host stage takes over
  dev_buf<int> order, tile_first, tile_of, old2new, cntE, cntH, cntI, bad, halo, nhelp;
  dev_buf<double> fnormal;
  dev_buf<long> boff, hoff;
  dev_buf<unsigned char> blob;
  PK_TRY(order.alloc((size_t)nown));
  PK_TRY(tile_first.alloc((size_t)nt + 1));
  PK_TRY(tile_of.alloc((size_t)nown));
  PK_TRY(old2new.alloc((size_t)nall));
  PK_TRY(cntE.alloc((size_t)nt));
  PK_TRY(cntH.alloc((size_t)nt));
  PK_TRY(cntI.alloc((size_t)nt));
  PK_TRY(bad.alloc(1));
  PK_TRY(fnormal.alloc((size_t)sd->nfaces * 3));
  PK_TRY(boff.alloc((size_t)nt + 1));
  PK_TRY(hoff.alloc((size_t)nt + 1));
  PK_TRY(hipMemcpy(order.p, tl->order, sizeof(int) * (size_t)nown, hipMemcpyHostToDevice));
  PK_TRY(hipMemcpy(tile_first.p, tl->tile_first, sizeof(int) * ((size_t)nt + 1), hipMemcpyHostToDevice));
  PK_TRY(hipMemcpy(tile_of.p, tl->tile_of, sizeof(int) * (size_t)nown, hipMemcpyHostToDevice));
  PK_TRY(hipMemcpy(old2new.p, P->old2new, sizeof(int) * (size_t)nall, hipMemcpyHostToDevice));
  PK_TRY(hipMemcpy(fnormal.p, &sd->fnormal[0][0], sizeof(double) * 3 * (size_t)sd->nfaces, hipMemcpyHostToDevice));
  PK_TRY(hipMemset(bad.p, 0, sizeof(int)));
  PK_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile_blobs<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  PK_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile_blobs<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  blob_args a;
  memset(&a, 0, sizeof a);
  a.xadj = c->d_xadj; a.adj_face = c->d_adj_face; a.adj_other = c->d_adj_other;
  dev_buf<int> own_xadj, own_face, own_other;  // stage 1 ran on the host: its arrays are uploaded here
  if (!c->d_xadj) {
    const int nadj = tl->xadj[nown];
    PK_TRY(own_xadj.alloc((size_t)nown + 1));
    PK_TRY(own_face.alloc((size_t)nadj));
    PK_TRY(own_other.alloc((size_t)nadj));
    PK_TRY(hipMemcpy(own_xadj.p, tl->xadj, sizeof(int) * ((size_t)nown + 1), hipMemcpyHostToDevice));
    PK_TRY(hipMemcpy(own_face.p, tl->adj_face, sizeof(int) * (size_t)nadj, hipMemcpyHostToDevice));
    PK_TRY(hipMemcpy(own_other.p, tl->adj_other, sizeof(int) * (size_t)nadj, hipMemcpyHostToDevice));
    a.xadj = own_xadj.p; a.adj_face = own_face.p; a.adj_other = own_other.p;
  }
  a.order = order.p; a.tile_first = tile_first.p; a.tile_of = tile_of.p; a.old2new = old2new.p;
  a.fnormal = fnormal.p; a.nown = nown;
  a.cntE = cntE.p; a.cntH = cntH.p; a.cntI = cntI.p; a.max_inc = max_inc; a.cap = cap; a.bad = bad.p;
  hipLaunchKernelGGL(k_tile_blobs<false>, dim3(nt), dim3(BLOB_T), lds, 0, a, np_max);
  PK_TRY(hipGetLastError());
  std::vector<int> E(nt), H(nt), I(nt);
  PK_TRY(hipMemcpy(E.data(), cntE.p, sizeof(int) * (size_t)nt, hipMemcpyDeviceToHost));
  PK_TRY(hipMemcpy(H.data(), cntH.p, sizeof(int) * (size_t)nt, hipMemcpyDeviceToHost));
  PK_TRY(hipMemcpy(I.data(), cntI.p, sizeof(int) * (size_t)nt, hipMemcpyDeviceToHost));
  int isbad = 0;
  PK_TRY(hipMemcpy(&isbad, bad.p, sizeof(int), hipMemcpyDeviceToHost));
  if (isbad) return cfdp_set_error("device plan stage 5: a tile overflowed its LDS hash table or its 16-bit neighbour / 15-bit face slots (counting pass)");
  // sizes -> descriptors, offsets, LDS classes (as the host stage's pass A)
  P->tiles = static_cast<cfdp_tile_desc *>(calloc((size_t)nt, sizeof(cfdp_tile_desc)));
  if (!P->tiles) return cfdp_set_error("out of memory");
  std::vector<long> h_boff((size_t)nt + 1, 0), h_hoff((size_t)nt + 1, 0);
  std::vector<int> h_nhelp((size_t)(nt ? nt : 1), 0);
  const bool split_lists = !(getenv("CFDP_SPLIT_LISTS") && atoi(getenv("CFDP_SPLIT_LISTS")) == 0);
  long lds_g[2] = {0, 0}, lds_f[2] = {0, 0}, dup_total = 0, inc_total = 0;
  for (int t = 0; t < nt; t++) {
    cfdp_tile_desc *td = &P->tiles[t];
    const int np = tl->tile_first[t + 1] - tl->tile_first[t];
    td->pstart = tl->tile_first[t]; td->npts = np;
    td->nhalo = H[t]; td->nfaces = E[t]; td->ninc = I[t];
    {  // helper lane groups of the tile's long lists -- or none where points and helpers do not fit its lane groups (as the host stage)
      int nh = 0;
      if (split_lists)
        for (int i = tl->tile_first[t]; i < tl->tile_first[t + 1]; i++)
          nh += cfdp_list_chunks(tl->xadj[tl->order[i] + 1] - tl->xadj[tl->order[i]], P->tile_points) - 1;
      h_nhelp[t] = np + nh <= P->tile_points ? nh : 0;
    }
    const long bytes = cfdp_blob_fn_bytes(E[t]) + cfdp_blob_inc_bytes(I[t]) + cfdp_blob_off_bytes(np) + cfdp_blob_help_bytes(h_nhelp[t]);
    td->blob_qw = (int)(bytes / 16);
    td->blob_off = (int)(h_boff[t] / 16);
    td->halo_off = (int)h_hoff[t];
    h_boff[t + 1] = h_boff[t] + bytes;
    h_hoff[t + 1] = h_hoff[t] + H[t];
    dup_total += E[t];
    inc_total += I[t];
    const int cls = t < P->nbtiles ? 0 : 1;
    const long lg = (long)td->blob_qw * 16 + (long)(np + H[t]) * 64, lf2 = (long)td->blob_qw * 16 + (long)(np + H[t]) * 80;
    if (lg > lds_g[cls]) lds_g[cls] = lg;
    if (lf2 > lds_f[cls]) lds_f[cls] = lf2;
  }
  P->blob_bytes = h_boff[nt];
  P->nhalo_total = h_hoff[nt];
  if (P->blob_bytes % 16 != 0 || P->blob_bytes / 16 >= 0x7FFFFFFF)
    return cfdp_set_error("device plan stage 5: %ld bytes of tile blobs do not fit 32-bit 16-byte offsets", (long)P->blob_bytes);
  PK_TRY(blob.alloc((size_t)P->blob_bytes));
  PK_TRY(halo.alloc((size_t)P->nhalo_total));
  PK_TRY(hipMemset(blob.p, 0, (size_t)(P->blob_bytes ? P->blob_bytes : 1)));  // alignment padding is defined
  PK_TRY(hipMemcpy(boff.p, h_boff.data(), sizeof(long) * ((size_t)nt + 1), hipMemcpyHostToDevice));
  PK_TRY(hipMemcpy(hoff.p, h_hoff.data(), sizeof(long) * ((size_t)nt + 1), hipMemcpyHostToDevice));
  PK_TRY(nhelp.alloc((size_t)(nt ? nt : 1)));
  PK_TRY(hipMemcpy(nhelp.p, h_nhelp.data(), sizeof(int) * (size_t)(nt ? nt : 1), hipMemcpyHostToDevice));
  a.boff = boff.p; a.hoff = hoff.p; a.blob = blob.p; a.halo_idx = halo.p; a.nhelp = nhelp.p; a.tile_points = P->tile_points; a.long_list = cfdp_long_list(); a.list_chunk = cfdp_list_chunk();
  hipLaunchKernelGGL(k_tile_blobs<true>, dim3(nt), dim3(BLOB_T), lds, 0, a, np_max);
  PK_TRY(hipGetLastError());
  P->blob = static_cast<unsigned char *>(malloc((size_t)(P->blob_bytes ? P->blob_bytes : 16)));
  P->halo_idx = static_cast<int *>(malloc(sizeof(int) * (size_t)(P->nhalo_total ? P->nhalo_total : 1)));
  if (!P->blob || !P->halo_idx) return cfdp_set_error("out of memory");
  PK_TRY(hipMemcpy(P->blob, blob.p, (size_t)P->blob_bytes, hipMemcpyDeviceToHost));
  if (P->nhalo_total) PK_TRY(hipMemcpy(P->halo_idx, halo.p, sizeof(int) * (size_t)P->nhalo_total, hipMemcpyDeviceToHost));
  PK_TRY(hipMemcpy(&isbad, bad.p, sizeof(int), hipMemcpyDeviceToHost));
  if (isbad) return cfdp_set_error("device plan stage 5: a tile overflowed its LDS hash table or its index slots (fill pass)");
  P->nfaces_dup = dup_total;
  P->ninc_total = inc_total;
  for (int cidx = 0; cidx < 2; cidx++) { P->lds_grad_cls[cidx] = lds_g[cidx]; P->lds_flux_cls[cidx] = lds_f[cidx]; }
  P->lds_grad = lds_g[0] > lds_g[1] ? lds_g[0] : lds_g[1];
  P->lds_flux = lds_f[0] > lds_f[1] ? lds_f[0] : lds_f[1];
  c->seconds[1] = wall() - t0;
  return 0;
}

// A stage that fails says why on stderr and returns non-zero: cfdp_plan_build_with then runs the host stage, which
// produces the same arrays bit for bit (out of device memory, a failed attribute call, a tile the kernel refuses).
int stage_csr_logged(const solver_data *sd, int **xadj, int **adj_face, int **adj_other, long *used, void *vctx) {
  const int rc = stage_csr(sd, xadj, adj_face, adj_other, used, vctx);
  if (rc) {
    fprintf(stderr, "cfdp_plan (device stage 1): %s\n", cfdp_gpu_last_error());
    stage_ctx *c = static_cast<stage_ctx *>(vctx);  // stage 5 uploads the host's arrays instead
    (void)hipFree(c->d_xadj); (void)hipFree(c->d_adj_face); (void)hipFree(c->d_adj_other);
    c->d_xadj = c->d_adj_face = c->d_adj_other = nullptr;
    c->seconds[0] = -1.0;
    (void)hipGetLastError();
  }
  return rc;
}

// stage 5 with the host's code when a tile is too big for the LDS hash (rc 2 above)
int stage_blobs_or_host(const solver_data *sd, const cfdp_tiling *tl, cfdp_plan *P, void *vctx) {
  const int rc = stage_blobs(sd, tl, P, vctx);
  if (rc == 0) return 0;
  static_cast<stage_ctx *>(vctx)->blobs_on_host = true;
  if (rc != 2) {
    fprintf(stderr, "cfdp_plan (device stage 5): %s\n", cfdp_gpu_last_error());
    (void)hipGetLastError();
    return rc;  // cfdp_plan_build_with frees the partial outputs and runs the host stage
  }
  return cfdp_plan_host_blobs(sd, tl, P);
}

}  // namespace

extern "C" {

// which: bit 0 = stage 1 (CSR) on the device, bit 1 = stage 5 (blobs) on the device; 3 = both
int cfdp_plan_build_gpu(const solver_data *sd, const comm_data *cd, const cfdp_plan_opts *opts, int device, int which,
                        cfdp_plan **out, double *stage_seconds) {
  if (!sd || !out) return cfdp_set_error("null argument");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return cfdp_set_error("no HIP device: the device-side plan stages have no CPU fallback");
  if (device < 0 || device >= n) return cfdp_set_error("device %d out of range [0,%d)", device, n);
  if (!(which & 3)) return cfdp_set_error("nothing to do on the device (which = %d)", which);
  stage_ctx ctx;
  ctx.device = device;
  cfdp_plan_stages st;
  st.csr = (which & 1) ? stage_csr_logged : nullptr;
  st.blobs = (which & 2) ? stage_blobs_or_host : nullptr;
  st.ctx = &ctx;
  cfdp_plan *P = cfdp_plan_build_with(sd, cd, opts, &st);
  if (!P) return cfdp_set_error("plan build failed");
  if (stage_seconds) {
    stage_seconds[0] = ctx.seconds[0];  // -1: the stage ran on the host
    stage_seconds[1] = ctx.blobs_on_host ? -1.0 : ctx.seconds[1];
  }
  *out = P;
  return 0;
}

}  // extern "C"
