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
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gg_kernels.h"

namespace {

__device__ __forceinline__ int xcd_tile(int b, int nb) {
  // Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one).  Give each
  // XCD a CONTIGUOUS run of tiles: tiles are numbered in growth order, so neighbours in
  // space share halo rows and duplicated faces through the same 4 MiB L2.  Speed only.
  const int x = b & 7, i = b >> 3, base = nb >> 3, rem = nb & 7;
  return x * base + (x < rem ? x : rem) + i;
}

template <int LPP> struct grad_cfg;
template <> struct grad_cfg<1> { static constexpr int NE = 7; };
template <> struct grad_cfg<2> { static constexpr int NE = 4; };
template <> struct grad_cfg<4> { static constexpr int NE = 2; };
template <> struct grad_cfg<8> { static constexpr int NE = 1; };

}  // namespace

// ------------------------------------------------------------------------------ gradient
template <int LPP>
__global__ __launch_bounds__(1024) void gg_gradient_kernel(
    const cfdp_tile_desc *__restrict__ tiles, int tile_begin, const uint4 *__restrict__ blob,
    const int *__restrict__ halo_idx, const double *__restrict__ var /*[nall][8]*/,
    const double *__restrict__ vol /*[nown]*/, double *__restrict__ grad /*[nall][21]*/) {
  constexpr int NE = grad_cfg<LPP>::NE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int t = tile_begin + xcd_tile(blockIdx.x, gridDim.x);
  const cfdp_tile_desc td = tiles[t];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int npts = td.npts, nhalo = td.nhalo;

  // ---- stage: blob (normals | incidences | offsets), own var rows, halo var rows ----
  uint4 *s4 = reinterpret_cast<uint4 *>(smem);
  const uint4 *b4 = blob + td.blob_off;
  for (int q = tid; q < td.blob_qw; q += nthr) s4[q] = b4[q];
  uint4 *v4 = s4 + td.blob_qw;  // var_l[(npts+nhalo)][8 doubles] = 4 x uint4 per row
  const uint4 *gv4 = reinterpret_cast<const uint4 *>(var);
  {
    const uint4 *own = gv4 + (size_t)td.pstart * 4;
    for (int q = tid; q < npts * 4; q += nthr) v4[q] = own[q];
    const int *hid = halo_idx + td.halo_off;
    for (int q = tid; q < nhalo * 4; q += nthr) {
      const int row = hid[q >> 2];
      v4[npts * 4 + q] = gv4[(size_t)row * 4 + (q & 3)];
    }
  }
  __syncthreads();

  const int li = tid / LPP, sub = tid % LPP;
  if (li >= npts) return;
  const int fn_bytes = (td.nfaces * 24 + 15) & ~15;
  const int inc_bytes = (td.ninc * 4 + 15) & ~15;
  const double *fn = reinterpret_cast<const double *>(smem);
  const uint32_t *inc = reinterpret_cast<const uint32_t *>(smem + fn_bytes);
  const uint32_t *ioff = reinterpret_cast<const uint32_t *>(smem + fn_bytes + inc_bytes);
  const double *var_l = reinterpret_cast<const double *>(v4);

  const int eq0 = sub * NE;
  double vs[NE], acc[NE][3];
#pragma unroll
  for (int j = 0; j < NE; j++) {
    vs[j] = var_l[li * 8 + eq0 + j];
    acc[j][0] = acc[j][1] = acc[j][2] = 0.0;
  }
  const int ks = (int)ioff[li], ke = (int)ioff[li + 1];
  for (int k = ks; k < ke; k++) {
    const uint32_t w = inc[k];
    const int nbr = (int)(w & 0xFFFFu), f = (int)((w >> 16) & 0x7FFFu);
    const double sg = (w >> 31) ? -0.5 : 0.5;  // owned end is p1: contribution is subtracted
    const double anx = fn[3 * f + 0], any = fn[3 * f + 1], anz = fn[3 * f + 2];
    const double *vn = var_l + nbr * 8 + eq0;
#pragma unroll
    for (int j = 0; j < NE; j++) {
      // val = 0.5*(var[p0][eq] + var[p1][eq])  (src/gradients.c:77,99,121); the sign of
      // the p1 side is folded into the exact factor +-0.5
      const double val = sg * (vs[j] + vn[j]);
      acc[j][0] += anx * val;
      acc[j][1] += any * val;
      acc[j][2] += anz * val;
    }
  }
  if (ke > ks) {  // a point without faces is in no colour list: the reference leaves it alone
    const double tmp = 1.0 / vol[td.pstart + li];  // src/gradients.c:138
    double *g = grad + (size_t)(td.pstart + li) * 21 + eq0 * 3;
#pragma unroll
    for (int j = 0; j < NE; j++) {
      if (eq0 + j < 7) {
        g[3 * j + 0] = acc[j][0] * tmp;
        g[3 * j + 1] = acc[j][1] * tmp;
        g[3 * j + 2] = acc[j][2] * tmp;
      }
    }
  }
}

// ---------------------------------------------------------------------------------- flux
// LPP lanes share a point and split its incidence list; partial sums are combined with
// wave shuffles in a fixed order (deterministic).
template <int LPP, bool REFMODE>
__global__ __launch_bounds__(1024) void gg_flux_kernel(
    const cfdp_tile_desc *__restrict__ tiles, int tile_begin, const uint4 *__restrict__ blob,
    const int *__restrict__ halo_idx, const double *__restrict__ grad /*[nall][21]*/,
    double *__restrict__ flux /*[nown][3]*/, int nown) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = tile_begin + xcd_tile(blockIdx.x, gridDim.x);
  const cfdp_tile_desc td = tiles[t];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int npts = td.npts, nhalo = td.nhalo;

  uint4 *s4 = reinterpret_cast<uint4 *>(smem);
  const uint4 *b4 = blob + td.blob_off;
  for (int q = tid; q < td.blob_qw; q += nthr) s4[q] = b4[q];
  // velocity-gradient block grad[p][IVX..IVZ][0..2] = first 9 doubles of each 21-double row
  double *g_l = reinterpret_cast<double *>(s4 + td.blob_qw);  // [(npts+nhalo)][10]
  const int *hid = halo_idx + td.halo_off;
  for (int q = tid; q < (npts + nhalo) * 9; q += nthr) {
    const int r = q / 9, c = q - 9 * r;
    const int row = r < npts ? td.pstart + r : hid[r - npts];
    g_l[r * 10 + c] = grad[(size_t)row * 21 + c];
  }
  __syncthreads();

  const int li = tid / LPP, sub = tid % LPP;
  const bool active = li < npts;
  const int fn_bytes = (td.nfaces * 24 + 15) & ~15;
  const int inc_bytes = (td.ninc * 4 + 15) & ~15;
  const double *fn = reinterpret_cast<const double *>(smem);
  const uint32_t *inc = reinterpret_cast<const uint32_t *>(smem + fn_bytes);
  const uint32_t *ioff = reinterpret_cast<const uint32_t *>(smem + fn_bytes + inc_bytes);

  double f0 = 0.0, f1 = 0.0, f2 = 0.0;
  int ks = 0, ke = 0;
  if (active) {
    ks = (int)ioff[li];
    ke = (int)ioff[li + 1];
    double gs[9];
#pragma unroll
    for (int c = 0; c < 9; c++) gs[c] = g_l[li * 10 + c];
    for (int k = ks + sub; k < ke; k += LPP) {
      const uint32_t w = inc[k];
      const int nbr = (int)(w & 0xFFFFu), f = (int)((w >> 16) & 0x7FFFu);
      const bool is_p1 = (w >> 31) != 0;
      if (REFMODE && !is_p1) {
        // reference 1-thread semantics (src/flux.c:177-182 with the class numbering of
        // src/rangelist.c:719-736): the p0 end only receives +flux when p1 is a ghost
        const bool nbr_ghost = nbr >= npts && hid[nbr - npts] >= nown;
        if (!nbr_ghost) continue;
      }
      const double nx = fn[3 * f + 0], ny = fn[3 * f + 1], nz = fn[3 * f + 2];
      const double *gn = g_l + nbr * 10;
      const double dvx_dx = 0.5 * (gs[0] + gn[0]), dvx_dy = 0.5 * (gs[1] + gn[1]),
                   dvx_dz = 0.5 * (gs[2] + gn[2]);
      const double dvy_dx = 0.5 * (gs[3] + gn[3]), dvy_dy = 0.5 * (gs[4] + gn[4]),
                   dvy_dz = 0.5 * (gs[5] + gn[5]);
      const double dvz_dx = 0.5 * (gs[6] + gn[6]), dvz_dy = 0.5 * (gs[7] + gn[7]),
                   dvz_dz = 0.5 * (gs[8] + gn[8]);
      const double mue_eff = 1.0, lambda = -2.0 / 3.0 * mue_eff;  // src/flux.c:125,163
      const double sts_xx = lambda * (dvy_dy + dvz_dz - 2.0 * dvx_dx);
      const double sts_yy = lambda * (dvx_dx + dvz_dz - 2.0 * dvy_dy);
      const double sts_zz = lambda * (dvx_dx + dvy_dy - 2.0 * dvz_dz);
      const double sts_xy = mue_eff * (dvx_dy + dvy_dx);
      const double sts_xz = mue_eff * (dvx_dz + dvz_dx);
      const double sts_yz = mue_eff * (dvy_dz + dvz_dy);
      const double fl0 = -(sts_xx * nx + sts_xy * ny + sts_xz * nz);
      const double fl1 = -(sts_xy * nx + sts_yy * ny + sts_yz * nz);
      const double fl2 = -(sts_xz * nx + sts_yz * ny + sts_zz * nz);
      if (is_p1) { f0 -= fl0; f1 -= fl1; f2 -= fl2; }
      else       { f0 += fl0; f1 += fl1; f2 += fl2; }
    }
  }
  // combine the LPP partial sums (lanes of one point are adjacent, LPP divides 64)
#pragma unroll
  for (int m = 1; m < LPP; m <<= 1) {
    f0 += __shfl_xor(f0, m, 64);
    f1 += __shfl_xor(f1, m, 64);
    f2 += __shfl_xor(f2, m, 64);
  }
  if (active && sub == 0 && ke > ks) {
    double *o = flux + (size_t)(td.pstart + li) * 3;
    o[0] = f0; o[1] = f1; o[2] = f2;
  }
}

// --------------------------------------------------------------------------- pack/unpack
__global__ __launch_bounds__(256) void gg_pack_kernel(const int *__restrict__ send_idx, int nsend,
                                                      const double *__restrict__ grad,
                                                      double *__restrict__ sendbuf) {
  const int n = nsend * 21;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int j = i / 21, c = i - 21 * j;
    sendbuf[i] = grad[(size_t)send_idx[j] * 21 + c];
  }
}

__global__ __launch_bounds__(256) void gg_unpack_kernel(const double *__restrict__ recvbuf, int nrecv,
                                                        int nown, double *__restrict__ grad) {
  const int n = nrecv * 21;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    grad[(size_t)nown * 21 + i] = recvbuf[i];  // ghost rows are in message order (host/tiling.c)
}

// ------------------------------------------------------------------------------ launchers
#define LAUNCH_GRAD(L)                                                                         \
  hipLaunchKernelGGL((gg_gradient_kernel<L>), dim3(ntiles), dim3(block), lds, stream, a.tiles, \
                     tile_begin, a.blob, a.halo_idx, a.var, a.vol, a.grad)

hipError_t gg_launch_gradient(const gg_args &a, int lanes, int tile_begin, int ntiles,
                              int tile_points, size_t lds, hipStream_t stream) {
  if (ntiles <= 0) return hipSuccess;
  const int block = ((tile_points * lanes + 63) / 64) * 64;
  if (block > 1024) return hipErrorInvalidConfiguration;
  switch (lanes) {
    case 1: LAUNCH_GRAD(1); break;
    case 2: LAUNCH_GRAD(2); break;
    case 4: LAUNCH_GRAD(4); break;
    case 8: LAUNCH_GRAD(8); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

#define LAUNCH_FLUX(L, R)                                                                        \
  hipLaunchKernelGGL((gg_flux_kernel<L, R>), dim3(ntiles), dim3(block), lds, stream, a.tiles,    \
                     tile_begin, a.blob, a.halo_idx, a.grad, a.flux, a.nown)

hipError_t gg_launch_flux(const gg_args &a, int lanes, bool refmode, int tile_begin, int ntiles,
                          int tile_points, size_t lds, hipStream_t stream) {
  if (ntiles <= 0) return hipSuccess;
  const int block = ((tile_points * lanes + 63) / 64) * 64;
  if (block > 1024) return hipErrorInvalidConfiguration;
  if (refmode) {
    switch (lanes) {
      case 1: LAUNCH_FLUX(1, true); break;
      case 2: LAUNCH_FLUX(2, true); break;
      case 4: LAUNCH_FLUX(4, true); break;
      case 8: LAUNCH_FLUX(8, true); break;
      default: return hipErrorInvalidValue;
    }
  } else {
    switch (lanes) {
      case 1: LAUNCH_FLUX(1, false); break;
      case 2: LAUNCH_FLUX(2, false); break;
      case 4: LAUNCH_FLUX(4, false); break;
      case 8: LAUNCH_FLUX(8, false); break;
      default: return hipErrorInvalidValue;
    }
  }
  return hipGetLastError();
}

hipError_t gg_launch_pack(const int *send_idx, int nsend, const double *grad, double *sendbuf,
                          hipStream_t stream) {
  if (nsend <= 0) return hipSuccess;
  const int n = nsend * 21;
  int blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gg_pack_kernel, dim3(blocks), dim3(256), 0, stream, send_idx, nsend, grad, sendbuf);
  return hipGetLastError();
}

hipError_t gg_launch_unpack(const double *recvbuf, int nrecv, int nown, double *grad,
                            hipStream_t stream) {
  if (nrecv <= 0) return hipSuccess;
  const int n = nrecv * 21;
  int blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gg_unpack_kernel, dim3(blocks), dim3(256), 0, stream, recvbuf, nrecv, nown, grad);
  return hipGetLastError();
}

hipError_t gg_set_max_lds(size_t lds_grad, size_t lds_flux) {
  hipError_t e = hipSuccess;
#define SET_LDS(K, B)                                                                          \
  if (e == hipSuccess && (B) > 65536)                                                          \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&K),                                \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(B));
  SET_LDS(gg_gradient_kernel<1>, lds_grad)
  SET_LDS(gg_gradient_kernel<2>, lds_grad)
  SET_LDS(gg_gradient_kernel<4>, lds_grad)
  SET_LDS(gg_gradient_kernel<8>, lds_grad)
  SET_LDS((gg_flux_kernel<1, false>), lds_flux)
  SET_LDS((gg_flux_kernel<2, false>), lds_flux)
  SET_LDS((gg_flux_kernel<4, false>), lds_flux)
  SET_LDS((gg_flux_kernel<8, false>), lds_flux)
  SET_LDS((gg_flux_kernel<1, true>), lds_flux)
  SET_LDS((gg_flux_kernel<2, true>), lds_flux)
  SET_LDS((gg_flux_kernel<4, true>), lds_flux)
  SET_LDS((gg_flux_kernel<8, true>), lds_flux)
#undef SET_LDS
  return e;
}
