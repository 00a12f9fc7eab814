// gg_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the CFD-Proxy hot path.
//
//   gg_gradient_kernel : Green-Gauss gradient face loop
//                        (reference private_compute_gradients_gg, src/gradients.c:25-147)
//   gg_flux_kernel     : pseudo viscous-flux face loop
//                        (reference private_compute_psd_flux, src/flux.c:111-190)
//   gg_pack_kernel     : halo gather into the send arena
//                        (reference exchange_dbl_copy_in, src/threads.c:791-813)
//   gg_unpack_kernel   : halo scatter from a staging buffer (only for transports that cannot
//                        deliver straight into the ghost rows)
//                        (reference exchange_dbl_copy_out, src/threads.c:816-839)
//
// Design (bandwidth-bound fp64 gather/scatter; no MFMA):
//   * one workgroup = one point TILE (see host/tiling.c).  The tile's static data (face
//     normals, incidence lists) is one contiguous, 16-byte aligned "blob" in HBM and is
//     streamed into LDS with 16 B/lane coalesced loads; the tile's own `var` rows are
//     contiguous too (points are renumbered tile-major); only the halo rows are gathered
//     (64-byte rows, 4 lanes per row) and those mostly hit the XCD's L2 because
//     neighbouring tiles run on the same XCD (blockIdx -> tile map below).
//   * race-free scatter without atomics or colouring: the reference lets every thread
//     process all faces touching its points and write only its own end
//     (src/rangelist.c:513-523,567-608).  Here a lane owns (point, equation group),
//     walks the point's incidence list and accumulates in REGISTERS; cross-tile faces are
//     simply stored in both tiles.  Zero-init ("first points") and the 1/volume scaling
//     ("last points", src/gradients.c:54-63,135-145) collapse into acc=0 and one multiply
//     before the single coalesced store of the finished 168-byte row.
//   * deterministic: a point's faces are added in file order, independent of the launch.
//
// Algorithmic bytes per launch: 32*F + 232*P_own + 56*P_add.
//
// grad in HBM (one allocation of nall*21 doubles, see gg_grad_view in gg_kernels.h): the 21 doubles of an OWNED row are
// split into A1 (6 doubles: the diagonal and the three symmetric sums of the 3x3 velocity-gradient block -- all the flux
// loop reads), A2 (4 doubles: the rest of the block and the row's tenth double) and B (doubles 10..20), stored as
//   [A1: nown x 6][ghost rows: nghost x 21, message order][A2: nown x 4][B: nown x 11]
// so that the flux loop streams contiguous, 16-byte aligned 48-byte rows, while the halo exchange still delivers whole
// 168-byte rows [A1 | A2 | B] straight into the ghost block.
#include "gg_device.h"

#include <dlfcn.h>

// --------------------------------------------------------------------------- pack/unpack
__global__ __launch_bounds__(256) void gg_pack_kernel(const int *__restrict__ send_idx, int nsend,
                                                      const double *__restrict__ gradA,
                                                      const double *__restrict__ gradA2,
                                                      const double *__restrict__ gradB,
                                                      double *__restrict__ sendbuf) {
  const int n = nsend * 21;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int j = i / 21, c = i - 21 * j;
    const size_t p = (size_t)send_idx[j];  // send points are owned points
    // whole 168-byte rows on the wire, in their stored form [A1 | A2 | B]
    sendbuf[i] = c < 6 ? gradA[p * 6 + c] : (c < 10 ? gradA2[p * 4 + c - 6] : gradB[p * 11 + c - 10]);
  }
}

__global__ __launch_bounds__(256) void gg_unpack_kernel(const double *__restrict__ recvbuf, int nrecv,
                                                        double *__restrict__ ghost) {
  const int n = nrecv * 21;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    ghost[i] = recvbuf[i];  // ghost rows are in message order (host/tiling.c)
}

// ------------------------------------------------------------- xGMI write + notify exchange
// The analogue of the reference's best variant, gaspi_write_notify + gaspi_notify_waitsome
// (src/exchange_data_gaspi.c:105-151,190-305), between processes on one node: the packing kernel
// writes the rows of every partner straight into that partner's landing arena (its memory,
// mapped here through a HIP IPC handle -- stores over xGMI), a second kernel then raises the
// iteration counter in each partner's flag word (system-scope release), and the receiver's
// stream runs a one-wave kernel that polls its own flag words (system-scope acquire) before the
// kernels that read the ghost rows.  All plain kernels: a whole iteration is hipGraph-capturable.
// dst[s] = partner s's landing rows for this rank (parity chosen by the host: two arenas,
// alternating by iteration, so a push never overwrites rows the partner may still be reading).
__global__ __launch_bounds__(256) void gg_push_kernel(const int *__restrict__ send_idx, int nsend,
                                                      const int *__restrict__ slot_of_row,
                                                      const int *__restrict__ send_off,
                                                      const double *__restrict__ gradA,
                                                      const double *__restrict__ gradA2,
                                                      const double *__restrict__ gradB,
                                                      double *const *__restrict__ dst) {
  const int n = nsend * 21;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int j = i / 21, c = i - 21 * j;
    const size_t p = (size_t)send_idx[j];
    const int s = slot_of_row[j];
    const double v = c < 6 ? gradA[p * 6 + c] : (c < 10 ? gradA2[p * 4 + c - 6] : gradB[p * 11 + c - 10]);
    dst[s][(size_t)(j - send_off[s]) * 21 + c] = v;
  }
}

// hdr: the layout of gg_kernels.h (one cache line per partner slot, then [GG_IPC_ITER], [GG_IPC_ERR ..])
// need != nullptr: counter notification -- the whole exchange at once: need[s] tiles' worth to partner s's counter,
// one exchange more for every boundary tile (the separate push kernel has stored every row before this kernel starts)
__global__ void gg_notify_kernel(int *__restrict__ hdr, int *const *__restrict__ remote_flag, int nslots,
                                 const int *__restrict__ need, int *__restrict__ tile_iter, int nbtiles) {
  if (need) {
    if ((int)threadIdx.x < nslots)
      (void)__hip_atomic_fetch_add(remote_flag[threadIdx.x], need[threadIdx.x], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    for (int t = threadIdx.x; t < nbtiles; t += blockDim.x) tile_iter[t] += 1;
    return;
  }
  const int it = hdr[GG_IPC_ITER] + 1;
  __syncthreads();
  if ((int)threadIdx.x < nslots)
    __hip_atomic_store(remote_flag[threadIdx.x], it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (threadIdx.x == 0) hdr[GG_IPC_ITER] = it;
}

__global__ void gg_wait_kernel(int *__restrict__ hdr, int nslots, long max_polls, const int *__restrict__ tile_iter) {
  if ((int)threadIdx.x >= nslots) return;
  if (hdr[GG_IPC_ERR]) return;  // a wait has given up before: the run is void anyway, do not stall every step
  int *slot = hdr + threadIdx.x * GG_IPC_SLOT_STRIDE;
  const int need = (tile_iter ? tile_iter[0] : hdr[GG_IPC_ITER]) * __hip_atomic_load(&slot[GG_IPC_NEED_IN], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  for (long k = 0; k < max_polls; k++) {
    if ((int)((unsigned)__hip_atomic_load(slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - (unsigned)need) >= 0) return;
    __builtin_amdgcn_s_sleep(32);  // ~1 us between polls: the flag line is not hammered
  }
  // bounded: a lost partner must not hang the device.  Leave what was seen for the post-mortem.
  hdr[GG_IPC_ERR] = 1;
  hdr[GG_IPC_ERR + 1] = (int)threadIdx.x;
  hdr[GG_IPC_ERR + 2] = need;
  hdr[GG_IPC_ERR + 3] = __hip_atomic_load(slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
  atomicAdd(&hdr[GG_IPC_ERR + 4], 1);
}

// ------------------------------------------------- scaled-field validation of an exchange
// The benchmark's field is constant in time, so a ghost row read one exchange too early -- the landing arenas
// alternate, the row of two exchanges ago sits at the same address -- has the same value and no comparison of final
// states can see it.  The reference asserts flag / stage lock-step at every receive (src/exchange_data_mpi.c:189,439,
// src/exchange_data_gaspi.c:389-416); the analogue here makes the VALUES carry the iteration number.  Everything is
// linear in var: after every iteration's gradient launch this kernel multiplies var by 2, 2, 1/4, 2, 2, 1/4, ...
// (exact in fp64), so the gradients, ghost rows and flux of iteration k are those of the first iteration times
// 2^((k-1) mod 3) EXACTLY, and the exponents of iterations k and k-2 always differ: a flux computed from a ghost row
// of two exchanges ago is off by a factor 2 or 4 in that row's terms.  The same kernel compares the flux the step
// has just produced with reference * 2^e, bit for bit.  All state lives on the device (state[GG_V_ITER] counts the
// gradient launches), so the kernel replays from a hipGraph like the steps it rides with.
//   lag  : d_flux holds the flux of iteration ITER + 1 - lag (1: fused mode, the pass computed the PREVIOUS
//          iteration's flux; 0: the step's own flux kernel has run; < 0: no flux to compare)
//   skip : optional, per owned point: its flux row is never written (a point without faces)
__global__ __launch_bounds__(256) void gg_validate_kernel(double *__restrict__ var, int nvar,
                                                          const double *__restrict__ flux, const double *__restrict__ fref,
                                                          const unsigned char *__restrict__ skip, int nflux, int lag,
                                                          int do_scale, int *__restrict__ state) {
  const int c = state[GG_V_ITER];  // read by every thread before its block takes a ticket: the last ticket moves it
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
  const bool check = lag >= 0 && c - lag >= 0;
  if (check) {
    const int e = (c - lag) % 3;
    const double s = e == 0 ? 1.0 : (e == 1 ? 2.0 : 4.0);
    int bad = 0;
    for (int i = gtid; i < nflux; i += gsz) {
      if (skip && skip[i / 3]) continue;
      const double want = fref[i] * s, got = flux[i];
      if (!(got == want)) {
        bad++;
        if (atomicCAS(&state[GG_V_CLAIM], 0, 1) == 0) {  // the first one found is kept for the report
          state[GG_V_FIRST_ITER] = c - lag + 1;
          state[GG_V_FIRST_IDX] = i;
          *reinterpret_cast<double *>(&state[GG_V_SEEN]) = got;
          *reinterpret_cast<double *>(&state[GG_V_EXPECT]) = want;
        }
      }
    }
    if (bad) atomicAdd(reinterpret_cast<unsigned long long *>(&state[GG_V_BAD]), (unsigned long long)bad);
  }
  if (do_scale) {
    const double f = (c % 3) == 2 ? 0.25 : 2.0;
    for (int i = gtid; i < nvar; i += gsz)
      if ((i & 7) != 7) var[i] *= f;  // slot 7 of a var row is the dual volume
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(&state[GG_V_TICKET], 1) == (int)gridDim.x - 1) {
    state[GG_V_TICKET] = 0;
    if (check) state[GG_V_CHECKS]++;
    if (do_scale) state[GG_V_ITER] = c + 1;
  }
}

__global__ __launch_bounds__(256) void gg_scale_var_kernel(double *__restrict__ var, int nvar, double f) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nvar; i += gridDim.x * blockDim.x)
    if ((i & 7) != 7) var[i] *= f;
}

hipError_t gg_launch_validate(double *var, int nall, const double *flux, const double *fref, const unsigned char *skip,
                              int nown, int lag, bool do_scale, int *state, hipStream_t stream) {
  const int nvar = nall * 8, nflux = nown * 3;
  int blocks = ((nvar > nflux ? nvar : nflux) + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  hipLaunchKernelGGL(gg_validate_kernel, dim3(blocks), dim3(256), 0, stream, var, nvar, flux, fref, skip, nflux, lag,
                     do_scale ? 1 : 0, state);
  return hipGetLastError();
}

// diagnostics of the validation itself: how many elements of var are NOT var0 * f (the mode's own bookkeeping went wrong)
__global__ __launch_bounds__(256) void gg_var_check_kernel(const double *__restrict__ var, const double *__restrict__ var0, int nvar,
                                                           double f, unsigned long long *__restrict__ bad) {
  int n = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nvar; i += gridDim.x * blockDim.x)
    if ((i & 7) != 7 && !(var[i] == var0[i] * f)) n++;
  if (n) atomicAdd(bad, (unsigned long long)n);
}
hipError_t gg_launch_var_check(const double *var, const double *var0, int nall, double factor, unsigned long long *bad, hipStream_t stream) {
  int blocks = (nall * 8 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  hipLaunchKernelGGL(gg_var_check_kernel, dim3(blocks), dim3(256), 0, stream, var, var0, nall * 8, factor, bad);
  return hipGetLastError();
}

hipError_t gg_launch_scale_var(double *var, int nall, double factor, hipStream_t stream) {
  int blocks = (nall * 8 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  hipLaunchKernelGGL(gg_scale_var_kernel, dim3(blocks), dim3(256), 0, stream, var, nall * 8, factor);
  return hipGetLastError();
}

// TESTS ONLY (CFDP_IPC_JITTER_US): one thread idles for a pseudo-random time in [0, max_us) in front of a step.  The
// generator's state lives on the device, so every replay of a hipGraph draws new delays: ranks drift against each other
// step by step, and a hole in the exchange protocol that runs in lockstep never hit shows up in the scaled-field check.
__global__ void gg_jitter_kernel(unsigned *__restrict__ rng, int max_us) {
  unsigned x = *rng;
  x ^= x << 13; x ^= x >> 17; x ^= x << 5;  // xorshift32
  *rng = x;
  const long long ticks = (long long)(x % (unsigned)(max_us > 0 ? max_us : 1)) * 100;  // s_memrealtime: 100 MHz
  const unsigned long long start = __builtin_amdgcn_s_memrealtime();
  while ((long long)(__builtin_amdgcn_s_memrealtime() - start) < ticks) __builtin_amdgcn_s_sleep(16);
}
// set-up: words[i] -> *dst[i] with system-scope stores -- how a rank tells its partners something once (how many of its
// boundary tiles count per exchange): the same kind of store, to the same mappings, as the exchange itself relies on
// (a host-side copy into another device's IPC-mapped memory is one more thing a machine could refuse)
__global__ void gg_poke_kernel(int *const *__restrict__ dst, const int *__restrict__ words, int offset, int n) {
  const int i = threadIdx.x;
  if (i < n) __hip_atomic_store(dst[i] + offset, words[i], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t gg_launch_poke(int *const *dst, const int *words, int offset, int n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(gg_poke_kernel, dim3(1), dim3(64), 0, stream, dst, words, offset, n);
  return hipGetLastError();
}
hipError_t gg_launch_jitter(unsigned *rng, int max_us, hipStream_t stream) {
  hipLaunchKernelGGL(gg_jitter_kernel, dim3(1), dim3(1), 0, stream, rng, max_us);
  return hipGetLastError();
}

hipError_t gg_launch_push(const int *send_idx, int nsend, const int *slot_of_row, const int *send_off,
                          const gg_grad_view &grad, double *const *dst, hipStream_t stream) {
  if (nsend <= 0) return hipSuccess;
  int blocks = (nsend * 21 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gg_push_kernel, dim3(blocks), dim3(256), 0, stream, send_idx, nsend, slot_of_row, send_off,
                     grad.a, grad.a2, grad.b, dst);
  return hipGetLastError();
}
hipError_t gg_launch_notify(int *hdr, int *const *remote_flag, int nslots, const int *need, int *tile_iter, int nbtiles, hipStream_t stream) {
  hipLaunchKernelGGL(gg_notify_kernel, dim3(1), dim3(64), 0, stream, hdr, remote_flag, nslots, need, tile_iter, nbtiles);
  return hipGetLastError();
}
hipError_t gg_launch_wait(int *hdr, int nslots, long max_polls, const int *tile_iter, hipStream_t stream) {
  hipLaunchKernelGGL(gg_wait_kernel, dim3(1), dim3(64), 0, stream, hdr, nslots, max_polls, tile_iter);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------ launchers
// Kernel forms that are instantiated (everything else was measured, lost and removed -- DESIGN.md appendix):
//   gradient  gg_gradient_dma_kernel<4, NT, CB, KV>   fixed-count LDS-DMA staging, 4 lanes per point (default)
//             gg_gradient_kernel<LPP, NT>             register-staged, any tile shape, LPP in {1,2,4,8} (fallback)
//   flux      gg_flux_dma_kernel<8, REF, NT, CB, KV>  fixed-count LDS-DMA staging, 8 lanes per point (default)
//             gg_flux_kernel<LPP, REF, NT>            register-staged, any tile shape (fallback)
//   fused     gg_fused_split_kernel<REF, NT, 5,3,3,3> one shared row region, 5 workgroups per CU (default); <6,4,3,4>: the
//                                                     large image (full tiles of unstructured meshes), 4 per CU
//             gg_fused_dma_kernel<REF, NT, CB,KV,KG>  everything staged up front (beside an RCCL kernel; larger tiles)
// (the diagnostic instantiations of the fused pass: gg_diag.hip, a library of their own)
// The diagnostic instantiations of the fused pass (phase stamps, data movement only, skip-pre) live in a library of their
// own, lib/libcfdproxy_diag.so (csrc/gg_diag.hip), looked for beside this one and loaded at the first diagnostic.
namespace {
typedef hipError_t (*diag_fused_fn)(int, int, int, const gg_args *, const gg_grad_view *, int, int, int, hipStream_t, int, const gg_push_args *);
typedef hipError_t (*diag_stamp_fn)(unsigned long long *);
struct diag_lib {
  void *h = nullptr;
  diag_fused_fn fused = nullptr;
  diag_stamp_fn stamp = nullptr;
  char why[512] = "";
};
diag_lib &diag() {
  static diag_lib D;
  static bool tried = false;
  if (tried) return D;
  tried = true;
  char path[4096] = "libcfdproxy_diag.so";
  Dl_info info;
  if (dladdr(reinterpret_cast<const void *>(&gg_diag_available), &info) && info.dli_fname) {
    const char *slash = strrchr(info.dli_fname, '/');
    if (slash) snprintf(path, sizeof path, "%.*s/libcfdproxy_diag.so", (int)(slash - info.dli_fname), info.dli_fname);
  }
  D.h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!D.h) {
    snprintf(D.why, sizeof D.why, "the diagnostic kernels are not built: %s (make -C cfd-proxy_amd lib/libcfdproxy_diag.so)", dlerror());
    return D;
  }
  D.fused = reinterpret_cast<diag_fused_fn>(dlsym(D.h, "gg_diag_launch_fused"));
  D.stamp = reinterpret_cast<diag_stamp_fn>(dlsym(D.h, "gg_diag_set_stamp_buffer"));
  if (!D.fused || !D.stamp) snprintf(D.why, sizeof D.why, "%s does not export the diagnostic entry points", path);
  return D;
}
}  // namespace
const char *gg_diag_available() {  // nullptr: available; else why not
  diag_lib &D = diag();
  return D.fused && D.stamp ? nullptr : D.why;
}
hipError_t gg_set_stamp_buffer(unsigned long long *dev) {
  diag_lib &D = diag();
  return D.stamp ? D.stamp(dev) : hipErrorSharedObjectInitFailed;
}
// fused pass: 0 everything staged up front; 1 the phase-split form at its large capacity only (36 KiB, 4 workgroups per
// CU); 2 (default) its small capacity where the tiles allow it (32 KiB, 5 workgroups per CU)
int gg_fused_split = 2;
// the gradient kernel's store slab ON its var rows (one more workgroup per CU): bit 0 at the small capacity (measured +-0 on the
// lattice stand-ins: off), bit 1 at the large one (<6,4>: four workgroups per CU instead of three: -8.5 % on the irregular
// stand-in: on).  CFDP_GRAD_ALIAS overrides (development).
int gg_grad_alias = 2;
int gg_debug_flags = 0;  // 16: register-staged kernels only; 64: per-lane row stores; GG_DBG_STAMP: phase stamps

namespace {

constexpr size_t LDS_MAX = 160 * 1024;

// which form ran (cfdp_gpu_kernel_forms): bounded, per thread, only while somebody has asked once
thread_local char g_forms[1024];
thread_local int g_forms_len = 0, g_forms_n = 0;
thread_local bool g_forms_on = false;
void note_form(const char *name, int a, int b, int c, int d, const char *suffix, int tile_begin, int ntiles) {
  if (!g_forms_on) return;
  g_forms_n++;
  char one[96];
  int n;
  if (d >= 0) n = snprintf(one, sizeof one, "%s<%d,%d,%d,%d>%s@%d+%d", name, a, b, c, d, suffix, tile_begin, ntiles);
  else if (c >= 0) n = snprintf(one, sizeof one, "%s<%d,%d,%d>%s@%d+%d", name, a, b, c, suffix, tile_begin, ntiles);
  else if (b >= 0) n = snprintf(one, sizeof one, "%s<%d,%d>%s@%d+%d", name, a, b, suffix, tile_begin, ntiles);
  else n = snprintf(one, sizeof one, "%s<%d>%s@%d+%d", name, a, suffix, tile_begin, ntiles);
  if (n <= 0 || g_forms_len + n + 2 > (int)sizeof g_forms) return;
  if (g_forms_len) {  // (back-to-back repeats of one launch are logged once)
    const char *last = g_forms + g_forms_len - n;
    if (g_forms_len >= n && !memcmp(last, one, (size_t)n) && (g_forms_len == n || last[-1] == ' ')) return;
    g_forms[g_forms_len++] = ' ';
  }
  memcpy(g_forms + g_forms_len, one, (size_t)n + 1);
  g_forms_len += n;
}

// raise the dynamic-LDS limit of a kernel to the full 160 KiB, once per kernel and device
template <typename K> hipError_t allow_lds(K *kernel) {
  static bool done[64] = {false};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < 64 && done[dev]) return hipSuccess;
  e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);
  if (e == hipSuccess && dev >= 0 && dev < 64) done[dev] = true;
  return e;
}

template <typename K, typename... A>
hipError_t launch(K *kernel, int grid, int block, size_t lds, hipStream_t stream, A... args) {
  if (lds > LDS_MAX) return hipErrorInvalidConfiguration;
  if (lds > 64 * 1024) {
    const hipError_t e = allow_lds(kernel);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds, stream, args...);
  return hipGetLastError();
}

template <int L> hipError_t launch_grad_generic(const gg_args &a, bool nt, int tile_begin, int ntiles, int block, size_t lds,
                                                hipStream_t stream) {
  note_form("gradient_generic", L, -1, -1, -1, "", tile_begin, ntiles);
  if (nt) return launch(gg_gradient_kernel<L, true>, ntiles, block, lds, stream, a.tiles, tile_begin, a.blob, a.halo_idx, a.var, a.grad);
  return launch(gg_gradient_kernel<L, false>, ntiles, block, lds, stream, a.tiles, tile_begin, a.blob, a.halo_idx, a.var, a.grad);
}

template <int CB, int KV, bool ALIAS = false> hipError_t launch_grad_dma(const gg_args &a, bool nt, int tile_begin, int ntiles, int block,
                                                                         size_t stage_bytes, hipStream_t stream) {
  const size_t lds = (size_t)(CB + KV) * block * 16 + (ALIAS ? 0 : stage_bytes);
  note_form("gradient_dma", CB, KV, -1, -1, ALIAS ? "alias" : "", tile_begin, ntiles);
  if (nt) return launch(gg_gradient_dma_kernel<4, true, CB, KV, ALIAS>, ntiles, block, lds, stream, a.tiles, tile_begin, a.blob, a.halo_idx, a.var, a.grad, gg_debug_flags);
  return launch(gg_gradient_dma_kernel<4, false, CB, KV, ALIAS>, ntiles, block, lds, stream, a.tiles, tile_begin, a.blob, a.halo_idx, a.var, a.grad, gg_debug_flags);
}

template <int L, bool R> hipError_t launch_flux_generic(const gg_args &a, bool nt, int tile_begin, int ntiles, int block, size_t lds,
                                                        hipStream_t stream) {
  note_form("flux_generic", L, -1, -1, -1, R ? "ref" : "", tile_begin, ntiles);
  if (nt) return launch(gg_flux_kernel<L, R, true>, ntiles, block, lds, stream, a.tiles, tile_begin, a.blob, a.halo_idx, a.grad.a, a.grad.ghost, a.flux, a.nown);
  return launch(gg_flux_kernel<L, R, false>, ntiles, block, lds, stream, a.tiles, tile_begin, a.blob, a.halo_idx, a.grad.a, a.grad.ghost, a.flux, a.nown);
}

template <bool R, int CB, int KV> hipError_t launch_flux_dma(const gg_args &a, bool nt, int tile_begin, int ntiles, int block,
                                                             hipStream_t stream, const gg_push_args *wait) {
  const size_t lds = (size_t)(CB + KV) * block * 16;
  note_form("flux_dma", CB, KV, -1, -1, wait ? "wait" : "", tile_begin, ntiles);
  gg_push_args pa;
  memset(&pa, 0, sizeof pa);
  if (wait) {
    pa = *wait;
    if (nt) return launch(gg_flux_dma_kernel<8, R, true, CB, KV, true>, ntiles, block, lds, stream, a.tiles, tile_begin, a.blob, a.halo_idx, a.grad.a, a.grad.ghost, a.flux, a.nown, pa);
    return launch(gg_flux_dma_kernel<8, R, false, CB, KV, true>, ntiles, block, lds, stream, a.tiles, tile_begin, a.blob, a.halo_idx, a.grad.a, a.grad.ghost, a.flux, a.nown, pa);
  }
  if (nt) return launch(gg_flux_dma_kernel<8, R, true, CB, KV>, ntiles, block, lds, stream, a.tiles, tile_begin, a.blob, a.halo_idx, a.grad.a, a.grad.ghost, a.flux, a.nown, pa);
  return launch(gg_flux_dma_kernel<8, R, false, CB, KV>, ntiles, block, lds, stream, a.tiles, tile_begin, a.blob, a.halo_idx, a.grad.a, a.grad.ghost, a.flux, a.nown, pa);
}

template <bool R, bool N, bool L, bool P>
hipError_t launch_split(const gg_args &a, const gg_grad_view &gnew, int tile_begin, int ntiles, int block, hipStream_t stream,
                        int dbgf, const gg_push_args &pa) {
  note_form("fused_split", 6, 4, 3, 4, L ? (P ? "listed+push" : "listed") : (P ? "push" : ""), tile_begin, ntiles);
  return launch(gg_fused_split_kernel<R, N, 6, 4, 3, 4, 0, L, P>, ntiles, block, (size_t)(6 + 4) * block * 16, stream, a.tiles, tile_begin,
                a.blob, a.halo_idx, a.rowlist, a.rowlist_stride, a.var, a.grad.a, a.grad.ghost, a.flux, a.nown, gnew, dbgf, pa);
}
template <bool R, bool N>
hipError_t launch_split_lp(const gg_args &a, const gg_grad_view &gnew, int tile_begin, int ntiles, int block, hipStream_t stream,
                           int dbgf, const gg_push_args &pa) {
  const bool listed = a.rowlist != nullptr, pushing = pa.tile_off != nullptr;
  if (listed) return pushing ? launch_split<R, N, true, true>(a, gnew, tile_begin, ntiles, block, stream, dbgf, pa)
                             : launch_split<R, N, true, false>(a, gnew, tile_begin, ntiles, block, stream, dbgf, pa);
  return pushing ? launch_split<R, N, false, true>(a, gnew, tile_begin, ntiles, block, stream, dbgf, pa)
                 : launch_split<R, N, false, false>(a, gnew, tile_begin, ntiles, block, stream, dbgf, pa);
}

// the small capacity of gg_fused_split_kernel: 5 blob + 3 row pieces per thread = a 32-KiB image, five workgroups per CU
template <bool R, bool N, bool L, bool P>
hipError_t launch_preg(const gg_args &a, const gg_grad_view &gnew, int tile_begin, int ntiles, int block, hipStream_t stream,
                       int dbgf, const gg_push_args &pa) {
  note_form("fused_split", 5, 3, 3, 3, L ? (P ? "listed+push" : "listed") : (P ? "push" : ""), tile_begin, ntiles);
  return launch(gg_fused_split_kernel<R, N, 5, 3, 3, 3, 0, L, P>, ntiles, block, (size_t)(5 + 3) * block * 16, stream, a.tiles, tile_begin,
                a.blob, a.halo_idx, a.rowlist, a.rowlist_stride, a.var, a.grad.a, a.grad.ghost, a.flux, a.nown, gnew, dbgf, pa);
}
template <bool R, bool N>
hipError_t launch_preg_lp(const gg_args &a, const gg_grad_view &gnew, int tile_begin, int ntiles, int block, hipStream_t stream,
                          int dbgf, const gg_push_args &pa) {
  const bool listed = a.rowlist != nullptr, pushing = pa.tile_off != nullptr;
  if (listed) return pushing ? launch_preg<R, N, true, true>(a, gnew, tile_begin, ntiles, block, stream, dbgf, pa)
                             : launch_preg<R, N, true, false>(a, gnew, tile_begin, ntiles, block, stream, dbgf, pa);
  return pushing ? launch_preg<R, N, false, true>(a, gnew, tile_begin, ntiles, block, stream, dbgf, pa)
                 : launch_preg<R, N, false, false>(a, gnew, tile_begin, ntiles, block, stream, dbgf, pa);
}

template <int CB, int KV, int KG>
hipError_t launch_fused_upfront(const gg_args &a, const gg_grad_view &gnew, bool refmode, bool nt, int tile_begin, int ntiles, int block,
                                hipStream_t stream, int dbgf, const gg_push_args &pa) {
  const size_t lds = (size_t)(CB + KV + KG) * block * 16;
  note_form("fused_upfront", CB, KV, KG, -1, pa.tile_off ? "push" : "", tile_begin, ntiles);
#define FUSED_ARGS ntiles, block, lds, stream, a.tiles, tile_begin, a.blob, a.halo_idx, a.var, a.grad.a, a.grad.ghost, a.flux, a.nown, gnew, dbgf, pa
  if (refmode) return nt ? launch(gg_fused_dma_kernel<true, true, CB, KV, KG>, FUSED_ARGS) : launch(gg_fused_dma_kernel<true, false, CB, KV, KG>, FUSED_ARGS);
  return nt ? launch(gg_fused_dma_kernel<false, true, CB, KV, KG>, FUSED_ARGS) : launch(gg_fused_dma_kernel<false, false, CB, KV, KG>, FUSED_ARGS);
#undef FUSED_ARGS
}

}  // namespace

int gg_forms_take(char *buf, size_t len) {
  g_forms_on = buf != nullptr;  // (a null buffer switches the log off again: launchers then note nothing)
  const int n = g_forms_n;
  if (buf && len) {
    const size_t m = (size_t)g_forms_len < len - 1 ? (size_t)g_forms_len : len - 1;
    memcpy(buf, g_forms, m);
    buf[m] = 0;
  }
  g_forms_len = g_forms_n = 0;
  g_forms[0] = 0;
  return n;
}

hipError_t gg_launch_gradient(const gg_args &a, int lanes, int tile_begin, int ntiles, int tile_points, size_t lds,
                              int max_halo, int max_blob_qw, bool nt, hipStream_t stream) {
  if (ntiles <= 0) return hipSuccess;
  const int block = ((tile_points * lanes + 63) / 64) * 64;
  if (block > 1024) return hipErrorInvalidConfiguration;
  lds = (lds + 15) & ~(size_t)15;
  const size_t stage_bytes = (size_t)(block / 64) * 8 * 21 * 8;  // 8 rows of 168 bytes per wave
  if (lanes == 4 && !(gg_debug_flags & 16)) {  // fixed-count LDS-DMA staging; two equations per lane
    const int cb = (max_blob_qw + block - 1) / block;                    // blob pieces per thread
    const int kv = ((tile_points + max_halo) * 4 + block - 1) / block;  // var-row pieces per thread
    if (cb >= 1 && kv >= 1) {
      const int alias = gg_grad_alias;  // bit 0: the small capacity, bit 1: the large one
      // (the slab must fit the var rows it lies on: 8 rows of 168 bytes per wave <= KV pieces per thread)
      if (cb <= 5 && kv <= 3 && (alias & 1) && stage_bytes <= (size_t)3 * block * 16) return launch_grad_dma<5, 3, true>(a, nt, tile_begin, ntiles, block, stage_bytes, stream);
      if (cb <= 5 && kv <= 3) return launch_grad_dma<5, 3>(a, nt, tile_begin, ntiles, block, stage_bytes, stream);
      if (cb <= 6 && kv <= 4 && (alias & 2) && stage_bytes <= (size_t)4 * block * 16) return launch_grad_dma<6, 4, true>(a, nt, tile_begin, ntiles, block, stage_bytes, stream);
      if (cb <= 5 && kv <= 4) return launch_grad_dma<5, 4>(a, nt, tile_begin, ntiles, block, stage_bytes, stream);
      if (cb <= 6 && kv <= 5) return launch_grad_dma<6, 5>(a, nt, tile_begin, ntiles, block, stage_bytes, stream);
      if (cb <= 8 && kv <= 6 && (size_t)(8 + 6) * block * 16 + stage_bytes <= LDS_MAX)
        return launch_grad_dma<8, 6>(a, nt, tile_begin, ntiles, block, stage_bytes, stream);
    }
  }
  switch (lanes) {
    case 1: return launch_grad_generic<1>(a, nt, tile_begin, ntiles, block, lds + stage_bytes, stream);
    case 2: return launch_grad_generic<2>(a, nt, tile_begin, ntiles, block, lds + stage_bytes, stream);
    case 4: return launch_grad_generic<4>(a, nt, tile_begin, ntiles, block, lds + stage_bytes, stream);
    case 8: return launch_grad_generic<8>(a, nt, tile_begin, ntiles, block, lds + stage_bytes, stream);
    default: return hipErrorInvalidValue;
  }
}

// can the flux kernel these tile sizes select wait for an exchange in its boundary tiles (gg_launch_flux's `wait`)?
bool gg_flux_can_wait(int lanes, int tile_points, int max_halo, int max_blob_qw) {
  const int block = ((tile_points * lanes + 63) / 64) * 64;
  if (block > 1024 || lanes != 8 || (gg_debug_flags & 16)) return false;
  const int cb = (max_blob_qw + block - 1) / block, kv = ((tile_points + max_halo) * 3 + block - 1) / block;
  return cb >= 1 && kv >= 1 && cb <= 6 && kv <= 4 && (size_t)(6 + 4) * block * 16 <= LDS_MAX;
}

hipError_t gg_launch_flux(const gg_args &a, int lanes, bool refmode, int tile_begin, int ntiles, int tile_points, size_t lds,
                          int max_halo, int max_blob_qw, bool nt, hipStream_t stream, const gg_push_args *wait) {
  if (ntiles <= 0) return hipSuccess;
  if (wait && !gg_flux_can_wait(lanes, tile_points, max_halo, max_blob_qw)) return hipErrorNotSupported;
  const int block = ((tile_points * lanes + 63) / 64) * 64;
  if (block > 1024) return hipErrorInvalidConfiguration;
  if (lanes == 8 && !(gg_debug_flags & 16)) {  // fixed-count LDS-DMA staging
    const int cb = (max_blob_qw + block - 1) / block;
    const int kv = ((tile_points + max_halo) * 3 + block - 1) / block;
#define FLUX_DMA(CB, KV)                                                                                 \
  if (cb <= CB && kv <= KV && (size_t)(CB + KV) * block * 16 <= LDS_MAX)                                   \
    return refmode ? launch_flux_dma<true, CB, KV>(a, nt, tile_begin, ntiles, block, stream, wait)         \
                   : launch_flux_dma<false, CB, KV>(a, nt, tile_begin, ntiles, block, stream, wait)
    if (cb >= 1 && kv >= 1) {
      FLUX_DMA(3, 2);
      FLUX_DMA(4, 3);
      FLUX_DMA(6, 4);
    }
#undef FLUX_DMA
  }
#define FLUX_GENERIC(L) \
  return refmode ? launch_flux_generic<L, true>(a, nt, tile_begin, ntiles, block, lds, stream) \
                 : launch_flux_generic<L, false>(a, nt, tile_begin, ntiles, block, lds, stream)
  switch (lanes) {
    case 1: FLUX_GENERIC(1);
    case 2: FLUX_GENERIC(2);
    case 4: FLUX_GENERIC(4);
    case 8: FLUX_GENERIC(8);
    default: return hipErrorInvalidValue;
  }
#undef FLUX_GENERIC
}

// flux(i) from `a.grad`, gradients(i+1) into `gnew`.  hipErrorNotSupported: no instantiated
// capacity fits this launch -- the caller runs the two separate kernels instead.
bool gg_fused_fits(int tile_points, int max_halo, int max_blob_qw) {
  const int block = ((tile_points * 4 + 63) / 64) * 64;
  if (block > 1024 || (gg_debug_flags & 16)) return false;
  const int cb = (max_blob_qw + block - 1) / block;
  const int kv = ((tile_points + max_halo) * 4 + block - 1) / block;
  const int kg = ((tile_points + max_halo) * 3 + block - 1) / block;
  return cb >= 1 && kv >= 1 && kg >= 1 && cb <= 8 && kv <= 6 && kg <= 8 &&
         (size_t)(8 + 6 + 8) * block * 16 <= LDS_MAX;  // the largest instantiated capacity
}

hipError_t gg_launch_fused(const gg_args &a, const gg_grad_view &gnew, bool refmode, int tile_begin,
                           int ntiles, int tile_points, int max_halo, int max_blob_qw, bool nt,
                           bool allow_split, hipStream_t stream, const gg_push_args *push, bool reverse) {
  if (ntiles <= 0) return hipSuccess;
  const int dbgf = gg_debug_flags | (reverse ? GG_DBG_REVERSE : 0);
  gg_push_args pa;
  memset(&pa, 0, sizeof pa);
  if (push) pa = *push;
  const int block = ((tile_points * 4 + 63) / 64) * 64;
  if (block > 1024 || (gg_debug_flags & 16)) return hipErrorNotSupported;
  const int cb = (max_blob_qw + block - 1) / block;
  const int kv = ((tile_points + max_halo) * 4 + block - 1) / block;
  const int kg = ((tile_points + max_halo) * 3 + block - 1) / block;
  if (cb < 1 || kv < 1 || kg < 1) return hipErrorNotSupported;
  if (gg_fused_split && allow_split && cb <= 6 && kv <= 4 && kg <= 3) {
    // the diagnostic instantiations (phase stamps, data movement only, the skip-pre timing experiment) are not in this library:
    // lib/libcfdproxy_diag.so (csrc/gg_diag.hip), loaded here at the first request, at the capacity the real pass of these tiles runs at
    {
      const int diag_kind = (gg_debug_flags & GG_DBG_STAMP) ? 1 : (gg_debug_flags & GG_DBG_MOVE) ? 2
                            : ((gg_debug_flags & 0x80000) && cb <= 5 && kv <= 3 && a.rowlist && !push && !refmode) ? 3 : 0;  // 3: CFDP_EXP_SKIP_PRE (EXPERIMENTS.md D.2)
      if (diag_kind) {
        if (!a.rowlist) return hipErrorNotSupported;  // the diagnostic instantiations read the fixed-stride row lists
        const int large = !(gg_fused_split >= 2 && cb <= 5 && kv <= 3);
        diag_lib &D = diag();
        if (!D.fused) return hipErrorSharedObjectInitFailed;
        note_form("fused_split", large ? 6 : 5, large ? 4 : 3, 3, large ? 4 : 3, diag_kind == 1 ? "stamp" : diag_kind == 2 ? "move" : "skip-pre", tile_begin, ntiles);
        return D.fused(diag_kind, large, nt ? 1 : 0, &a, &gnew, tile_begin, ntiles, block, stream, dbgf, &pa);
      }
    }
    // tiles of at most 192 staged rows (3 var pieces per thread): the 32-KiB capacity
    if (gg_fused_split >= 2 && cb <= 5 && kv <= 3) {
      if (refmode) return nt ? launch_preg_lp<true, true>(a, gnew, tile_begin, ntiles, block, stream, dbgf, pa)
                             : launch_preg_lp<true, false>(a, gnew, tile_begin, ntiles, block, stream, dbgf, pa);
      return nt ? launch_preg_lp<false, true>(a, gnew, tile_begin, ntiles, block, stream, dbgf, pa)
                : launch_preg_lp<false, false>(a, gnew, tile_begin, ntiles, block, stream, dbgf, pa);
    }
    if (refmode) return nt ? launch_split_lp<true, true>(a, gnew, tile_begin, ntiles, block, stream, dbgf, pa)
                           : launch_split_lp<true, false>(a, gnew, tile_begin, ntiles, block, stream, dbgf, pa);
    return nt ? launch_split_lp<false, true>(a, gnew, tile_begin, ntiles, block, stream, dbgf, pa)
              : launch_split_lp<false, false>(a, gnew, tile_begin, ntiles, block, stream, dbgf, pa);
  }
  if (gg_debug_flags & (GG_DBG_MOVE | GG_DBG_STAMP)) return hipErrorNotSupported;  // diagnostics exist for the phase-split form only
  // everything staged up front; the store slab (8 rows of 168 bytes per wave) must fit the gradient-row region
  const size_t slab = (size_t)(block / 64) * 8 * 21 * 8;
#define FUSED_UPFRONT(CB, KV, KG)                                                                                   \
  if (cb <= CB && kv <= KV && kg <= KG && (size_t)(CB + KV + KG) * block * 16 <= LDS_MAX && slab <= (size_t)(KG) * block * 16) \
    return launch_fused_upfront<CB, KV, KG>(a, gnew, refmode, nt, tile_begin, ntiles, block, stream, dbgf, pa)
  FUSED_UPFRONT(5, 3, 4);
  FUSED_UPFRONT(5, 4, 5);  // 56 KiB at 64-point tiles: two workgroups per CU
  FUSED_UPFRONT(8, 6, 8);
#undef FUSED_UPFRONT
  return hipErrorNotSupported;
}

hipError_t gg_launch_pack(const int *send_idx, int nsend, const gg_grad_view &grad, double *sendbuf,
                          hipStream_t stream) {
  if (nsend <= 0) return hipSuccess;
  const int n = nsend * 21;
  int blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gg_pack_kernel, dim3(blocks), dim3(256), 0, stream, send_idx, nsend, grad.a, grad.a2, grad.b, sendbuf);
  return hipGetLastError();
}

hipError_t gg_launch_unpack(const double *recvbuf, int nrecv, const gg_grad_view &grad,
                            hipStream_t stream) {
  if (nrecv <= 0) return hipSuccess;
  const int n = nrecv * 21;
  int blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gg_unpack_kernel, dim3(blocks), dim3(256), 0, stream, recvbuf, nrecv, grad.ghost);
  return hipGetLastError();
}

