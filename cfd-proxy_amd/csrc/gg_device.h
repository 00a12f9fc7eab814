// gg_device.h -- the device side of the face-loop kernels: what a tile does (staging, the two face loops, stores, the exchange
// riding in a pass) and the kernel templates built from it.  Included by gg_kernels.hip -- the product: the forms a timed run
// can execute -- and by gg_diag.hip -- the DIAGNOSTIC instantiations of the phase-split fused pass (phase stamps, data movement
// only, the skip-pre timing experiment), which live in a library of their own (libcfdproxy_diag.so) that the product loads
// only when a diagnostic is asked for.  Everything here is a template or inline: two translation units may hold it.
#ifndef CFDP_GG_DEVICE_H
#define CFDP_GG_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "gg_kernels.h"

namespace {

__device__ __forceinline__ int xcd_tile(int b, int nb, bool rev = false) {
  // Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one).  Give each
  // XCD a CONTIGUOUS run of tiles: tiles are numbered in growth order, so neighbours in
  // space share halo rows and duplicated faces through the same 4 MiB L2.  Speed only.
  // rev: the XCD walks its run backwards.  Passes that alternate the direction start on the
  // tiles the previous pass finished with, whose rows and blobs are the most recent content of
  // the 256 MiB Infinity Cache -- on meshes that stream more than it holds per pass.
  const int x = b & 7, base = nb >> 3, rem = nb & 7;
  int i = b >> 3;
  if (rev) i = base + (x < rem ? 1 : 0) - 1 - i;
  return x * base + (x < rem ? x : rem) + i;
}

// The same for a launch whose first nbt tiles are a rank's boundary tiles (smaller, and the ones
// whose results the partners wait for): those are dealt round-robin over ALL XCDs, first in the
// grid -- one round instead of three on a single XCD, and no XCD left with a chunk of half-size
// tiles -- and each XCD owns a contiguous chunk of the remaining tiles.
__device__ __forceinline__ int xcd_tile_bfirst(int b, int nb, int nbt, bool rev = false) {
  if (nbt <= 0) return xcd_tile(b, nb, rev);
  if (b < nbt) return b;
  const int x = b & 7;
  int start = 0, mine = 0;
#pragma unroll
  for (int y = 0; y < 8; y++) {
    const int b0 = nbt + ((y - nbt) & 7);  // the first non-boundary block that lands on XCD y
    const int cnt = b0 < nb ? (nb - b0 + 7) >> 3 : 0;
    if (y < x) start += cnt;
    if (y == x) mine = (b - b0) >> 3;
  }
  return nbt + start + mine;
}

// non-temporal 16-byte load: for data that is streamed exactly once per launch
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_nt(const uint4 *p) {
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

// NT = the launch streams more than the Infinity Cache holds: blobs are read and rows written
// with the non-temporal policy so that they do not evict the var / grad rows that neighbouring
// tiles re-read as halo rows (measured on 128^3: -8 % kernel time).  For cache-resident meshes
// (64^3: everything stays in the 256 MiB Infinity Cache across iterations) the default policy
// is faster (+20 % with nt), so the host picks per launch.
template <bool NT> __device__ __forceinline__ uint4 ld_blob(const uint4 *p) {
  if constexpr (NT) return ld_nt(p);
  else return *p;
}
template <bool NT> __device__ __forceinline__ void st_row(double v, double *p) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// bit of the kernels' `dbg` argument that is not a timing experiment: walk the tiles backwards
#define GG_DBG_REVERSE 0x10000
// diagnostics (dbg bit GG_DBG_STAMP selects a separate instantiation of the split fused pass, STAMP = true; the
// kernels of a timed run contain none of it -- merely compiled in and switched off it cost 1-2.5 %): thread 0 of
// every workgroup writes shader-clock stamps of its phase boundaries, 8 per tile, to a buffer of its own
#define GG_DBG_STAMP 0x20000
// a third instantiation of the same pass: every load and every store, neither face loop (the data-movement floor)
#define GG_DBG_MOVE 0x40000
#ifndef GG_DEEP_BATCH
#define GG_DEEP_BATCH 7
#endif
#ifndef GG_FLUX_BATCH
#define GG_FLUX_BATCH 1
#endif
#ifndef GG_WAVES_EU
#define GG_WAVES_EU 4
#endif
template <int LPP> struct grad_cfg;
template <> struct grad_cfg<1> { static constexpr int NE = 7; };
template <> struct grad_cfg<2> { static constexpr int NE = 4; };
template <> struct grad_cfg<4> { static constexpr int NE = 2; };
template <> struct grad_cfg<8> { static constexpr int NE = 1; };

}  // namespace

// ------------------------------------------------------------------------------ gradient
// The per-tile arithmetic, shared by the simple and the pipelined kernel.  `buf` is the LDS
// image of one tile: [nx | ny | nz | incidences | offsets | var rows (own, then halo)].
// var rows are 8 doubles: 7 variables + the point's dual volume in slot 7.
//
// U consecutive incidences of a point are processed as one batch: all U incidence words are
// read, then all their operands (normal components, neighbour row), then the FMAs -- so a
// lane pays the LDS round trip once per batch, not twice per incidence (a wave only has its
// own ~14 dependent steps; measured: the un-batched loop spent ~2/3 of its time parked on
// lgkmcnt).  The accumulation order stays the file order of the point's faces.
template <int U, int NE>
__device__ __forceinline__ void grad_batch(const uint32_t *__restrict__ inc, int k,
                                           const double *__restrict__ nx, const double *__restrict__ ny,
                                           const double *__restrict__ nz,
                                           const double *__restrict__ var_eq0, const double (&vs)[NE],
                                           double (&acc)[NE][3]) {
  uint32_t w[U];
#pragma unroll
  for (int i = 0; i < U; i++) w[i] = inc[k + i];
  double n0[U], n1[U], n2[U], vn[U][NE];
#pragma unroll
  for (int i = 0; i < U; i++) {
    const uint32_t f = (w[i] >> 16) & 0x7FFFu;
    n0[i] = nx[f];
    n1[i] = ny[f];
    n2[i] = nz[f];
    const double *vp = var_eq0 + (w[i] & 0xFFFFu) * 8;
#pragma unroll
    for (int j = 0; j < NE; j++) vn[i][j] = vp[j];
  }
#pragma unroll
  for (int i = 0; i < U; i++) {
    // val = 0.5*(var[p0][eq] + var[p1][eq]) (src/gradients.c:77,99,121); the owned end being p1 means the contribution
    // is subtracted (:103-105,128-130).  Neither costs a multiply: the sign goes onto the sum as bit 31 of the incidence
    // word (= the sign bit of a double), and the 0.5 -- a power of two, it commutes with every rounding -- is applied
    // once, with 1/pvolume, to the finished sums (grad_tile_compute).  Bit for bit the sums of the factor +-0.5 per face.
    const int sgn = (int)(w[i] & 0x80000000u);
#pragma unroll
    for (int j = 0; j < NE; j++) {
      const double sum = vs[j] + vn[i][j];
      const double val = __hiloint2double(__double2hiint(sum) ^ sgn, __double2loint(sum));
      acc[j][0] += n0[i] * val;
      acc[j][1] += n1[i] * val;
      acc[j][2] += n2[i] * val;
    }
  }
}

// write-through system-scope store (sc0 sc1): the bytes leave this device's caches with the store itself.  Inline asm is
// invisible to the compiler's vmcnt bookkeeping: the caller drains with s_waitcnt vmcnt(0) before it signals.
typedef unsigned int gg_u32x4 __attribute__((ext_vector_type(4)));
// s_nop 1: a store of more than 64 bits reads its data VGPRs over several cycles, and the instruction BEHIND an inline-asm
// store is not checked against it by the compiler's hazard recogniser -- a VALU write of those registers in the next slot
// (the address of the next piece, typically) went out as the first 8 bytes of the row piece (found in round 5: ghost rows
// whose doubles 2, 8 and 14 held an address).  Two wait states, as the ISA asks for this write-after-read case.
__device__ __forceinline__ void st16_sys(void *p, gg_u32x4 v) {  // global_store_dwordx4 ... sc0 sc1
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
// xGMI write + notify, a point's first destination, pushed from the REGISTERS of its lanes the moment the row is finished:
// the lane holds the doubles [eq0 * 3, eq0 * 3 + n) of the 168-byte row (n = 6, or 3 for the lane group with the last
// equation) and `row` is where the row lands in the partner's arena -- known long before (the table entry is requested at
// the top of the kernel, the slice pointer behind the staging wait), so nothing but the stores themselves sits on the
// boundary tile's critical path.  Write-through system-scope stores, 16 bytes wherever the arena's alignment allows
// (rows start at 0 or 8 mod 16).  The caller drains them (s_waitcnt vmcnt(0), push_tile_done) before it counts the tile.
// The row leaves in its stored form (gg_a_encode): lane group 0 holds g0..g5 and stores e0..e5 = [g0 g4 g8 g1+g3 g2+g6
// g5+g7], lane group 1 holds g6..g11 and stores [e6..e9 | d10 d11] = [g3 g6 g7 g9 | g10 g11]; ex[] = what each takes from
// the other (group 0: g6 g7 g8 of group 1; group 1: g3 of group 0), shuffled in by the caller.
template <int NE>
__device__ __forceinline__ void push_from_registers(double *row, int eq0, const double (&acc)[NE][3], double tmp,
                                                    const double (&ex)[3]) {
  double *p = row + eq0 * 3;
  auto st8 = [](double *q, double x) { __hip_atomic_store(q, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
  auto st16 = [](double *q, double a, double b) {
    gg_u32x4 u;
    u.x = (unsigned)__double2loint(a); u.y = (unsigned)__double2hiint(a);
    u.z = (unsigned)__double2loint(b); u.w = (unsigned)__double2hiint(b);
    st16_sys(q, u);
  };
  if constexpr (NE == 2) {  // 4 lanes per point (every kernel that pushes): 6 doubles, the last lane group 3
    double v0 = acc[0][0] * tmp, v1 = acc[0][1] * tmp, v2 = acc[0][2] * tmp;
    double v3 = acc[1][0] * tmp, v4 = acc[1][1] * tmp, v5 = acc[1][2] * tmp;
    if (eq0 == 0) {  // g0..g5 here, ex = g6 g7 g8
      const double e0 = v0, e1 = v4, e2 = ex[2], e3 = v1 + v3, e4 = v2 + ex[0], e5 = v5 + ex[1];
      v0 = e0; v1 = e1; v2 = e2; v3 = e3; v4 = e4; v5 = e5;
    } else if (eq0 == 2) {  // g6..g11 here, ex[0] = g3
      const double e6 = ex[0], e7 = v0, e8 = v1, e9 = v3;
      v0 = e6; v1 = e7; v2 = e8; v3 = e9;  // v4, v5 = d10, d11 as they are
    }
    const bool six = eq0 + 2 <= 7;
    if (((uintptr_t)p & 15) == 0) {
      st16(p, v0, v1);
      if (six) { st16(p + 2, v2, v3); st16(p + 4, v4, v5); }
      else st8(p + 2, v2);
    } else {
      st8(p, v0);
      st16(p + 1, v1, v2);
      if (six) { st16(p + 3, v3, v4); st8(p + 5, v5); }
    }
  } else {
#pragma unroll
    for (int j = 0; j < NE; j++)
      if (eq0 + j < 7) {
#pragma unroll
        for (int c = 0; c < 3; c++) st8(p + 3 * j + c, acc[j][c] * tmp);
      }
  }
}

// Long incidence lists cut into chunks (cfdproxy_host.h): the helper table of a tile image, behind its offsets -- present only
// in tiles that have helpers (nhelp = 0: nothing behind the offsets, the blob ends there).  Uniform per workgroup.
struct gg_helpers {
  int n;                  // helper lane groups: tile-local slots npts .. npts + n - 1 (they own no row)
  const uint32_t *tab;    // [n] target li | chunk << 16, in (point, chunk) order
  double *scratch;        // [n][24] partial sums on their way to the point's own lanes
};
__device__ __forceinline__ gg_helpers tile_helpers(const unsigned char *buf, const cfdp_tile_desc &td, int plane, int inc_bytes) {
  gg_helpers h;
  const int base = 3 * plane + inc_bytes + (((td.npts + 1) * 4 + 15) & ~15);
  h.n = td.blob_qw * 16 > base ? (int)*reinterpret_cast<const uint32_t *>(buf + base) : 0;
  h.tab = reinterpret_cast<const uint32_t *>(buf + base) + 1;
  h.scratch = reinterpret_cast<double *>(const_cast<unsigned char *>(buf) + base + (((1 + h.n) * 4 + 15) & ~15));
  return h;
}
// the part [ks, ke) of its list a lane group walks: the whole list, or -- a list cut into nchunks -- chunk c of it
__device__ __forceinline__ void list_chunk(const uint32_t *ioff, int li, int chunk, int &ks, int &ke, int &ks0, int &ke0, int &nchunks) {
  const uint32_t w0 = ioff[li], w1 = ioff[li + 1];
  ks0 = (int)(w0 & 0xFFFFFFu);
  ke0 = (int)(w1 & 0xFFFFFFu);
  nchunks = (int)(w0 >> 24) + 1;
  ks = ks0;
  ke = ke0;
  if (nchunks > 1) {
    const int len = (ke0 - ks0 + nchunks - 1) / nchunks;
    ks = ks0 + chunk * len;
    ke = ks + len < ke0 ? ks + len : ke0;
    if (ks > ke0) ks = ke = ke0;
  }
}

// The finished rows are stored 8 bytes per lane in contiguous runs (NT: non-temporal).
// MOVE (the data-movement floor, a diagnostic instantiation): no incidence is walked -- every row is stored as zeros
// through the same slab and the same store instructions.
// pushing (uniform per workgroup): some lanes of this tile push their rows (push_row != nullptr for those)
template <int LPP, bool NT, bool SYNC = false, bool MOVE = false>
__device__ __forceinline__ void grad_tile_compute(const unsigned char *buf, const cfdp_tile_desc &td,
                                                  int tid, const gg_grad_view &gout,
                                                  double *__restrict__ stage, int dbg = 0,
                                                  int var_off = -1, double *push_row = nullptr, bool pushing = false) {
  double *__restrict__ gradA = gout.a, *__restrict__ gradA2 = gout.a2, *__restrict__ gradB = gout.b;
  constexpr int NE = grad_cfg<LPP>::NE;
  constexpr int PPW = 64 / LPP;  // points per wave
  const int li = tid / LPP, sub = tid % LPP;
  const bool active = li < td.npts;
  const int plane = (td.nfaces * 8 + 15) & ~15;
  const int inc_bytes = (td.ninc * 4 + 15) & ~15;
  const double *nx = reinterpret_cast<const double *>(buf);
  const double *ny = reinterpret_cast<const double *>(buf + plane);
  const double *nz = reinterpret_cast<const double *>(buf + 2 * plane);
  const uint32_t *inc = reinterpret_cast<const uint32_t *>(buf + 3 * plane);
  const uint32_t *ioff = reinterpret_cast<const uint32_t *>(buf + 3 * plane + inc_bytes);
  // var rows follow the blob (packed image) or sit at a fixed offset (fixed-capacity image)
  const double *var_l = reinterpret_cast<const double *>(buf + (var_off >= 0 ? (size_t)var_off : (size_t)td.blob_qw * 16));

  const int eq0 = sub * NE;
  double vs[NE], acc[NE][3];
#pragma unroll
  for (int j = 0; j < NE; j++) acc[j][0] = acc[j][1] = acc[j][2] = 0.0;
  // (a long list is walked in chunks: this lane group takes the first -- or, a HELPER group in a slot behind the tile's points,
  // another chunk of some point's list, with that point's var row; its sums join the point's below)
  const gg_helpers hp = tile_helpers(buf, td, plane, inc_bytes);
  const bool helper = !active && li < td.npts + hp.n;
  int src = li, chunk = 0;
  if (helper) {
    const uint32_t hw = hp.tab[li - td.npts];
    src = (int)(hw & 0xFFFFu);
    chunk = (int)(hw >> 16);
  }
  int ks = 0, ke0 = 0, nchunks = 1;  // [ks, ke0): the point's WHOLE list (faceless / push decisions below)
  double tmp = 0.0;
  if (active || helper) {
#pragma unroll
    for (int j = 0; j < NE; j++) vs[j] = var_l[src * 8 + eq0 + j];
    int k, kend;
    list_chunk(ioff, src, chunk, k, kend, ks, ke0, nchunks);
    const int ke = MOVE ? k : kend;
    const double *var_eq0 = var_l + eq0;
    // (each lane runs the chain "sevens, a four, a two, a one" over its OWN list; a wave issues a batch for as long as any
    // lane has one, but lanes without it are switched off and read nothing from LDS, which is what this loop is short of:
    // one schedule per wave with lanes sitting out masked batches -- built and measured in round 6 -- was 4-8 % slower)
    if constexpr (NE <= 2)  // few registers per incidence: deeper batches (a point has ~14 incidences)
      for (; k + GG_DEEP_BATCH <= ke; k += GG_DEEP_BATCH) grad_batch<GG_DEEP_BATCH, NE>(inc, k, nx, ny, nz, var_eq0, vs, acc);
    for (; k + 4 <= ke; k += 4) grad_batch<4, NE>(inc, k, nx, ny, nz, var_eq0, vs, acc);
    if (k + 2 <= ke) {
      grad_batch<2, NE>(inc, k, nx, ny, nz, var_eq0, vs, acc);
      k += 2;
    }
    if (k < ke) grad_batch<1, NE>(inc, k, nx, ny, nz, var_eq0, vs, acc);
    if (active) tmp = 0.5 / var_l[li * 8 + 7];  // 1/pvolume, src/gradients.c:138, and the face value's 0.5 (grad_batch)
  }
  if (hp.n) {  // (uniform per workgroup) the helpers' sums join their points': through LDS, in chunk order
    if (helper) {
#pragma unroll
      for (int j = 0; j < NE; j++)
        if (eq0 + j < 8) {
#pragma unroll
          for (int c = 0; c < 3; c++) hp.scratch[(li - td.npts) * 24 + (eq0 + j) * 3 + c] = acc[j][c];
        }
    }
    __syncthreads();
    if (active && nchunks > 1)
      for (int h = 0; h < hp.n; h++)
        if ((int)(hp.tab[h] & 0xFFFFu) == li) {
#pragma unroll
          for (int j = 0; j < NE; j++)
            if (eq0 + j < 8) {
#pragma unroll
              for (int c = 0; c < 3; c++) acc[j][c] += hp.scratch[h * 24 + (eq0 + j) * 3 + c];
            }
        }
  }
  // this point's row goes to a partner: out of the registers, now (before the tile's own stores: the acknowledgement of a
  // remote store takes longest).  A point without faces is pushed by nobody (such partitions keep the push kernel)
  if constexpr (NE == 2) {
    if (pushing) {  // (every lane takes part in the shuffles: lanes 4p and 4p + 1 swap what the stored form mixes)
      double ex[3];
      const double up = __shfl_up(acc[1][0] * tmp, 1, 64);  // g3 of lane group 0, for group 1
      ex[0] = __shfl_down(acc[0][0] * tmp, 1, 64);          // g6 g7 g8 of lane group 1, for group 0
      ex[1] = __shfl_down(acc[0][1] * tmp, 1, 64);
      ex[2] = __shfl_down(acc[0][2] * tmp, 1, 64);
      if (sub == 1) ex[0] = up;
      if (push_row && active && ke0 > ks) push_from_registers<NE>(push_row, eq0, acc, tmp, ex);
    }
  }
  // SYNC: `stage` aliases a region of the tile image other waves may still be reading
  if constexpr (SYNC) __syncthreads();
  // ---- write the finished rows.  A lane holds NE*3 doubles of a 168-byte row; storing them
  // directly is 8 bytes per lane at a 24..168-byte stride (measured: the stores alone then
  // take longer than streaming the whole tile in).  Instead each wave transposes its PPW rows
  // through a private LDS slab and writes them as contiguous runs, 8 bytes per lane x 64
  // lanes per instruction.  Wave-private slab => no workgroup barrier.
  const int wave = tid >> 6, lane = tid & 63;
  const int wp = wave * PPW;                                    // first point of this wave
  const int nv = td.npts - wp < PPW ? td.npts - wp : PPW;       // its valid points (may be <= 0)
  // a point without faces is in no colour list: the reference leaves its row alone (rare: its row is skipped below)
  const unsigned long long faceless = __ballot(active && ke0 == ks);
  // 8 points (1344 bytes) per pass: the slab stays small enough for five workgroups per CU.
  // Slab image of a pass: [A1: 8 x 6][A2: 8 x 4][B: 8 x 11], the three runs it is stored as.
  constexpr int SPP = 8, NPASS = PPW / SPP;
  double *slab = stage + wave * (SPP * 21);
  // where this lane's values go in the slab of the pass its point belongs to: the first ten doubles of a row in their stored
  // ORDER (gg_a_encode, gg_kernels.h: raw double d -> slot, one nibble each) -- A1 [g0 g4 g8 | g1 g2 g5], A2 [g3 g6 g7 g9];
  // the three sums are formed on the way out.  Computed once per lane, not per value and pass.
  int sidx[NE][3];
  {
    const int lpp = (li - wp) & (SPP - 1);
    const int a1 = lpp * 6, a2 = SPP * 6 + lpp * 4 - 6, bb = SPP * 10 + lpp * 11 - 10;
#pragma unroll
    for (int j = 0; j < NE; j++)
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const int d = (eq0 + j) * 3 + c;
        const int slot = (int)((0x9287516430ull >> (4 * (d < 10 ? d : 0))) & 15ull);
        sidx[j][c] = d >= 10 ? bb + d : (slot < 6 ? a1 : a2) + slot;
      }
  }
  // round 0 of a full pass: lanes 0..47 hold A1, slot k = lane % 6 of row lane / 6; slots 3..5 take A2's 0..2 on board
  const bool sum0 = lane < 48 && (unsigned)lane % 6u >= 3u;
  const int sidx0 = sum0 ? SPP * 6 + (lane / 6) * 4 + (lane % 6 - 3) : 0;
#pragma unroll
  for (int h = 0; h < NPASS; h++) {
    const int lp = li - wp - h * SPP;  // this lane's point within the pass
    if (active && lp >= 0 && lp < SPP) {
#pragma unroll
      for (int j = 0; j < NE; j++)
        if (eq0 + j < 7) {
#pragma unroll
          for (int c = 0; c < 3; c++) slab[sidx[j][c]] = acc[j][c] * tmp;
        }
    }
    __builtin_amdgcn_wave_barrier();  // LDS executes a wave's accesses in order; keep the compiler in order too
    int nvh = nv - h * SPP;
    nvh = nvh < 0 ? 0 : (nvh > SPP ? SPP : nvh);
    const size_t p0 = (size_t)(td.pstart + wp + h * SPP);
    double *ga1 = gradA + p0 * 6, *ga2 = gradA2 + p0 * 4, *gb = gradB + p0 * 11;
    const int na1 = nvh * 6, na2 = nvh * 4, nd = nvh * 21;
    // three rounds of 64 doubles cover the pass's 8 x 21; the symmetric sums of the stored form (g1 + g3, g2 + g6, g5 + g7:
    // A1's slots 3..5 take A2's 0..2 on board) are formed here, on the way out
    auto out = [&](int c, bool skip_rows) {
      if (c >= nd) return;
      const int row = c < na1 ? c / 6 : (c < na1 + na2 ? (c - na1) / 4 : (c - na1 - na2) / 11);
      if (skip_rows && ((faceless >> ((h * SPP + row) * LPP)) & 1ull)) return;  // a point without faces: its row stays
      if (c < na1) {
        double v = slab[c];
        const int k = c - 6 * row;
        if (k >= 3) v += slab[SPP * 6 + row * 4 + k - 3];
        st_row<NT>(v, &ga1[c]);
      } else if (c < na1 + na2) {
        st_row<NT>(slab[SPP * 6 + c - na1], &ga2[c - na1]);
      } else {
        st_row<NT>(slab[SPP * 10 + c - na1 - na2], &gb[c - na1 - na2]);
      }
    };
    if (nvh == SPP && !faceless) {
      // the common case, a full pass (uniform per wave): the slab is one run [A1: 48][A2: 32][B: 88] and so is what
      // leaves -- no index arithmetic, no branch; which lanes hold a sum slot is known per lane (sum0, sidx0)
      {
        const double v = slab[lane], t = slab[sidx0];
        st_row<NT>(sum0 ? v + t : v, lane < 48 ? &ga1[lane] : &ga2[lane - 48]);
      }
      st_row<NT>(slab[lane + 64], lane < 16 ? &ga2[lane + 16] : &gb[lane - 16]);
      if (lane < 40) st_row<NT>(slab[lane + 128], &gb[lane + 48]);
    } else {  // a partial pass (the tile's last points) or a point without faces in this wave: rare
      for (int i = 0; i < 3; i++) out(lane + 64 * i, faceless != 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// xGMI write + notify, pushed by the tile itself: once the tile's rows are stored, ALL its threads copy the rows of its send
// points into the partners' landing arenas -- value i of the tile's (entries x 21) doubles goes to thread i mod nthr, so a
// 168-byte row leaves as one contiguous run of stores and every load (entry word, slice pointer, row number, the value
// itself out of L2) is independent of every other.  (Round 4, tools/loopback_probe.py: the first form walked the entries
// one after the other, every thread testing every entry, with three dependent loads each and the 4 lanes of the point
// storing 8 bytes at a 24-byte stride from their registers: 40-80 entries x ~1 us on the critical path of EVERY boundary
// tile -- a pass with exchange took 99 us where the same pass without took 39.)
__device__ __forceinline__ void push_tile_rows(const gg_push_args &pa, int tile, int tid, int nthr, const cfdp_tile_desc &td,
                                               const gg_grad_view &gout) {
  const double *gradA = gout.a, *gradA2 = gout.a2, *gradB = gout.b;
  if (!pa.tile_off || tile >= pa.nbtiles) return;  // uniform per workgroup
  // a 168-byte row leaves as ELEVEN stores: ten of 16 bytes and one of 8 (rows start at 0 or 8 mod 16 in the arena; narrow
  // write-through stores are one fabric write each and cost 2.7x a 16-byte store per byte, MI355X_MICROARCH.md)
  // (only the FURTHER destinations of points sent to several partners take this road -- a point's first destination is
  // pushed from the registers of its lanes, push_from_registers; most boundary tiles have none)
  const int e0 = pa.tile_xoff[tile], n = (pa.tile_off[tile + 1] - e0) * 11;
  if (n <= 0) return;  // uniform per workgroup
  __syncthreads();  // s_waitcnt vmcnt(0) + barrier: every wave's row stores have been acknowledged by L2
  for (int i = tid; i < n; i += nthr) {
    const int e = e0 + i / 11, q = i % 11;
    const int w = pa.ent[e];
    const size_t p = (size_t)td.pstart + (size_t)(w & 0xFFFF);
    double *row = pa.dst[w >> 16] + (size_t)pa.ent_row[e] * 21;
    const bool odd = ((uintptr_t)row & 15) != 0;        // the row starts 8 mod 16: one double first, then ten pairs
    const int c0 = odd ? (q == 0 ? 0 : 2 * q - 1) : 2 * q;  // first double of this piece
    const bool pair = odd ? q != 0 : q != 10;
    // past this CU's L1: the values were written a moment ago by other waves of this workgroup
    auto ld = [&](int c) {  // (rows travel in their stored form: [A1 | A2 | B])
      const double *src = c < 6 ? gradA + p * 6 + c : (c < 10 ? gradA2 + p * 4 + (c - 6) : gradB + p * 11 + (c - 10));
      return __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // write-through at system scope: the bytes leave this device's caches with the store itself, so NO cache write-back
    // is needed before the flag (push_tile_done) -- a system-scope release fence there would be a buffer_wbl2 of an L2
    // full of the interior tiles' freshly written rows, once per boundary tile
    if (pair) {
      const double v0 = ld(c0), v1 = ld(c0 + 1);
      gg_u32x4 v;
      v.x = (unsigned)__double2loint(v0); v.y = (unsigned)__double2hiint(v0);
      v.z = (unsigned)__double2loint(v1); v.w = (unsigned)__double2hiint(v1);
      st16_sys(row + c0, v);
    } else {
      __hip_atomic_store(row + c0, ld(c0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// exchanges this rank has announced so far, read once at the top of a boundary tile (flag form: the word only moves when
// the LAST boundary tile of a launch has finished, i.e. after every tile's read here)
__device__ __forceinline__ int exchanges_so_far(const gg_push_args &pa, int tile) {
  if (pa.need && pa.counters) return pa.tile_iter[tile];  // this tile's own word: it stores it at its end, nobody else does
  return pa.hdr[GG_IPC_ITER];
}

// The wait for the previous exchange, at the top of a boundary tile of the next pass (see gg_push_args::
// wait_polls; the protocol is gg_wait_kernel's).  iter0 = hdr[GG_IPC_ITER] = exchanges this rank has announced so
// far, read once at the top of the pass: it only changes when the LAST boundary tile of a launch has finished, i.e.
// after every tile's wait here.
// Per-partner protocol (pa.need != nullptr; the reference's receiving thread waits only for the partners it needs,
// src/exchange_data_gaspi.c:389-416): the tile polls the flags of the partners it SENDS to (pa.tile_mask).  Those
// are the partners whose ghost rows it may read (the host has checked: a tile reads ghost rows only of partners it
// holds send points for), and their flag k also says that they are done reading what this tile is about to overwrite
// in their arena of the same parity (a partner raises its flag k towards this rank only when all its tiles that hold
// send rows for -- hence may read ghost rows from -- this rank have finished pass k).
__device__ __forceinline__ void wait_previous_exchange(const gg_push_args &pa, int tile, int tid, int iter0) {
  if (!pa.tile_off || tile >= pa.nbtiles) return;  // uniform per workgroup
  // (a pass whose wait has been settled by the wait kernel -- ranks sharing a device, CFDP_IPC_WAIT_INKERNEL=0 -- still
  // owes split mode its invalidate below: the wait kernel's acquire touched its own CU and XCD only)
  if (pa.wait_polls <= 0 && !pa.inv_after_flag) return;
  const bool mine = pa.wait_polls > 0 && tid < pa.nslots && (!pa.need || ((pa.tile_mask[tile] >> tid) & 1ull));
  if (mine && !pa.hdr[GG_IPC_ERR]) {
    // flag notification: partner s has stored its exchange number.  Counter notification: s's boundary tiles have each
    // added 1 for every exchange they completed for this rank, NEED_IN of them per exchange (compared wrap-safe)
    // (what the word must reach is the SENDER's statement -- NEED_IN: its boundary tiles that count per exchange when it
    // notifies by counters, 1 when it stores its exchange number -- so a rank need not know which form a neighbour resolved to)
    const int *slot = pa.hdr + tid * GG_IPC_SLOT_STRIDE;
    const int need = iter0 * __hip_atomic_load(&slot[GG_IPC_NEED_IN], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    bool ok = false;
    // RELAXED system-scope polls (global_load_dword ... sc0 sc1, past every cache): an acquire per poll is a cache
    // invalidate per poll (MI355X_MICROARCH.md: polling with acquire loads is 2-3x slower per hop and many pollers cut
    // the chip's bandwidth).  No invalidate is needed behind the flag either: EVERY load of the rows the flag stands
    // for is itself a system-scope load (glds16_sys / the generic flux kernel's atomic loads), issued after this poll
    // has returned (the loop exit depends on its value) and, for the other waves, behind the barrier below.
    for (long k = 0; k < pa.wait_polls && !ok; k++) {
      ok = (int)((unsigned)__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - (unsigned)need) >= 0;
      if (!ok) __builtin_amdgcn_s_sleep(32);
    }
    if (!ok) {  // bounded: a lost partner must not hang the device.  Leave what was seen for the post-mortem.
      pa.hdr[GG_IPC_ERR] = 1;
      pa.hdr[GG_IPC_ERR + 1] = tid;
      pa.hdr[GG_IPC_ERR + 2] = need;
      pa.hdr[GG_IPC_ERR + 3] = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      atomicAdd(&pa.hdr[GG_IPC_ERR + 4], 1);
    }
  }
  // "split" memory mode: rows in a coarse-grained arena behind flags in fine-grained memory -- drop whatever this CU's
  // L1 / this XCD's L2 still hold, by name, once per tile (one wave), before the first row is requested
  if (pa.inv_after_flag && tid < 64) asm volatile("buffer_inv sc0 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// The same wait with its FIRST poll overlapped with the tile's staging (the phase-split pass): a system-scope load goes
// past every cache -- 1-2 us under load -- and with the wait in front of everything a boundary tile paid that round trip
// before it requested a single byte, every pass.  Here wave 0 requests its partners' words at the very top of the tile
// (wait_first_poll: untracked loads, older than every staging load of the wave, so the counted vmcnt in front of the row
// gathers covers them), the descriptor, the row numbers and the blob travel meanwhile, and wait_check -- placed where the
// row numbers have arrived, in front of the first request for a ghost row -- finds the word there in the common case
// (partner not late) and polls on only otherwise.  The other waves meet wave 0 at a raw s_barrier (no memory wait: the
// blob stays in flight) before they request rows.
__device__ __forceinline__ int ld_i32_nowait(const int *p);
__device__ __forceinline__ unsigned long long ld_u64_nowait(const void *p);
__device__ __forceinline__ int ld_i32_sys_nowait(const int *p) {
  int v;  // global_load_dword ... sc0 sc1 (system scope); not tracked by the compiler: the caller waits (counted vmcnt)
  asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ bool wait_wanted(const gg_push_args &pa, int tile) {  // uniform per workgroup
  return pa.tile_off && tile < pa.nbtiles && (pa.wait_polls > 0 || pa.inv_after_flag);
}
struct gg_wait_probe {  // what wave 0 requests at the top of a boundary tile (every field an untracked load)
  int word = 0, nin = 1, err = 0;
  unsigned long long mask = ~0ull;
};
__device__ __forceinline__ void wait_first_poll(const gg_push_args &pa, int tile, int tid, gg_wait_probe &w) {
  if (tid >= 64) return;  // wave 0, every lane (clamped): what the wave issues does not depend on data
  const int *slot = pa.hdr + (tid < pa.nslots ? tid : pa.nslots - 1) * GG_IPC_SLOT_STRIDE;
  w.word = ld_i32_sys_nowait(slot);
  w.nin = ld_i32_sys_nowait(slot + GG_IPC_NEED_IN);
  w.err = ld_i32_nowait(pa.hdr + GG_IPC_ERR);
  if (pa.need) w.mask = ld_u64_nowait(pa.tile_mask + tile);
}
__device__ __forceinline__ void wait_check(const gg_push_args &pa, int tile, int tid, int iter0, gg_wait_probe w) {
  if (tid < 64) {
    asm volatile("" : "+v"(w.word), "+v"(w.nin), "+v"(w.err), "+v"(w.mask));  // (uses stay behind the caller's counted wait)
    const int word = w.word, nin = w.nin;
    const bool mine = pa.wait_polls > 0 && tid < pa.nslots && ((w.mask >> tid) & 1ull);
    if (mine && !w.err) {
      const int *slot = pa.hdr + tid * GG_IPC_SLOT_STRIDE;
      const int need = iter0 * nin;  // nin: what the SENDER says its word advances by per exchange (see wait_previous_exchange)
      bool ok = (int)((unsigned)word - (unsigned)need) >= 0;
      for (long k = 0; k < pa.wait_polls && !ok; k++) {  // the partner IS late: poll on (relaxed, see wait_previous_exchange)
        __builtin_amdgcn_s_sleep(32);
        ok = (int)((unsigned)__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - (unsigned)need) >= 0;
      }
      if (!ok) {
        pa.hdr[GG_IPC_ERR] = 1;
        pa.hdr[GG_IPC_ERR + 1] = tid;
        pa.hdr[GG_IPC_ERR + 2] = need;
        pa.hdr[GG_IPC_ERR + 3] = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        atomicAdd(&pa.hdr[GG_IPC_ERR + 4], 1);
      }
    }
    if (pa.inv_after_flag) asm volatile("buffer_inv sc0 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");  // split mode, see above
  }
  asm volatile("s_barrier" ::: "memory");  // the other waves request ghost rows only behind wave 0's verdict
}

// After the pushes of a boundary tile: count it.  Every wave has waited for the acknowledgement of its (write-through)
// remote stores before the tile is counted.
// Coarse protocol (pa.need == nullptr): ONE counter; the last boundary tile of the launch raises this rank's iteration
// counter in every partner's flag word (gg_notify_kernel's job, done in place).
// Per-partner protocol: one counter per partner slot, need[s] = the boundary tiles that hold send rows for partner s;
// the tile that completes partner s's rows raises s's flag AT ONCE -- the reference's thread that completes partner
// k's buffer fires k's send (src/threads.c:268-311) -- while other boundary tiles are still computing.
__device__ __forceinline__ void push_tile_done(const gg_push_args &pa, int tile, int tid, int iter0, int dbg = 0) {
  if (!pa.tile_off || tile >= pa.nbtiles) return;  // uniform per workgroup
  // drained-flag publish (MI355X_MICROARCH.md, inter-workgroup visibility, the write-through form): the pushes were
  // write-through system-scope stores; every storing wave waits until they are acknowledged, the workgroup meets, and only
  // then ONE lane counts the tile / raises a flag with a relaxed system-scope store.  No fence: nothing sits in a cache.
  if (!(dbg & 0x1000)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const int it = iter0 + 1;
  if (pa.need && pa.counters) {
    // counter notification: nothing comes back to the tile.  Lane s adds 1 to partner s's counter word (a no-return
    // system-scope atomic: fire and forget), lane 0 stores the tile's own exchange count -- its next pass reads it at its
    // top, behind the kernel boundary.  Takes the two dependent device-scope atomics of the flag form (2-3 us each under
    // load) off the boundary tile's critical path; the reference's notification travels with the write as well
    // (gaspi_write_notify, src/exchange_data_gaspi.c:134-145).
    if (!(dbg & 0x800) && tid < pa.nslots && ((pa.tile_mask[tile] >> tid) & 1ull))
      // (the pointer comes out of a table: say that it is global memory, or the add is a flat_ instruction)
      (void)__hip_atomic_fetch_add((__attribute__((address_space(1))) int *)pa.rflag[tid], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (tid == 0) pa.tile_iter[tile] = it;
    return;
  }
  if (pa.need) {  // lane s of wave 0 looks after partner slot s (at most GG_IPC_MAXSLOTS = 48 of them)
    if (!(dbg & 0x800) && tid < pa.nslots && ((pa.tile_mask[tile] >> tid) & 1ull)) {
      // (the flag store depends on the value the add returns: it is issued after every earlier tile's count -- and so
      // after their acknowledged pushes -- has been observed)
      if (atomicAdd(&pa.done[(1 + tid) * GG_DONE_STRIDE], 1) == pa.need[tid] - 1) {
        pa.done[(1 + tid) * GG_DONE_STRIDE] = 0;  // nobody counts this slot again before the next launch
        __hip_atomic_store(pa.rflag[tid], it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // second level: the slots that are complete.  Every boundary tile counts on at least one slot (it holds send
        // points), so "every slot complete" = "every boundary tile has counted": the rank's iteration counter moves.
        // nslots adds on this word per pass instead of one per boundary tile (hundreds of adds on ONE address cost the
        // pass 1 us; the slot counters sit on cache lines of their own for the same reason: 2 us, tools/loopback_ablate.py)
        if (atomicAdd(&pa.done[0], 1) == pa.nslots - 1) {
          pa.done[0] = 0;
          pa.hdr[GG_IPC_ITER] = it;
        }
      }
    }
    return;
  }
  if (tid == 0) {
    const int old = atomicAdd(&pa.done[0], 1);
    if (old == pa.nbtiles - 1) {
      pa.done[0] = 0;  // nobody counts again before the next launch
      for (int s = 0; s < pa.nslots; s++)
        __hip_atomic_store(pa.rflag[s], it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      pa.hdr[GG_IPC_ITER] = it;
    }
  }
}

// Simple form: one workgroup per tile, stage through registers, barrier, compute.
template <int LPP, bool NT>
__global__ __launch_bounds__(1024) void gg_gradient_kernel(
    const cfdp_tile_desc *__restrict__ tiles, int tile_begin, const uint4 *__restrict__ blob,
    const int *__restrict__ halo_idx, const double *__restrict__ var /*[nall][8]*/,
    gg_grad_view gout) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int t = tile_begin + xcd_tile(blockIdx.x, gridDim.x);
  const cfdp_tile_desc td = tiles[t];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int npts = td.npts, nhalo = td.nhalo;

  // ---- stage: blob (normals | incidences | offsets), own var rows, halo var rows ----
  uint4 *s4 = reinterpret_cast<uint4 *>(smem);
  const uint4 *b4 = blob + td.blob_off;
  for (int q = tid; q < td.blob_qw; q += nthr) s4[q] = ld_blob<NT>(&b4[q]);
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
  double *stage = reinterpret_cast<double *>(smem + (size_t)td.blob_qw * 16 + (size_t)(npts + nhalo) * 64);
  grad_tile_compute<LPP, NT>(smem, td, tid, gout, stage);
}

// ---- fixed-count LDS-DMA staging ------------------------------------------------------------
// The register-staged form above serialises ~5 global round trips per tile (the compiler
// waits between the blob, own-row, halo-index and halo-row loops).  Here every wave issues a
// FIXED number of LDS-DMA instructions with full EXEC and clamped addresses (out-of-range
// pieces re-read the last valid 16 bytes into the padding of a fixed-capacity LDS region), so
// that the only true dependency -- halo row numbers -> halo rows -- can be waited for with a
// COUNTED vmcnt while the CB blob pieces issued after the index loads stay in flight.
// LDS image: [blob region: CB*nthr*16 B][var rows, own then halo: KV*nthr*16 B].
__device__ __forceinline__ int ld_i32_nowait(const int *p) {
  int v;  // the compiler does not track this load: the caller waits for it (counted vmcnt)
  asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ unsigned long long ld_u64_nowait(const void *p) {
  unsigned long long v;  // not tracked by the compiler either: issued ahead of the counted wait, used behind it
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void glds16(const uint4 *src, unsigned char *lds_wave_base);
__device__ __forceinline__ void glds16_nt(const uint4 *src, unsigned char *lds_wave_base);
__device__ __forceinline__ void glds16_sys(const uint4 *src, unsigned char *lds_wave_base);

template <bool NT, int CB, int KV>
__device__ __forceinline__ void dma_stage_tile(unsigned char *buf, const cfdp_tile_desc &td,
                                               const uint4 *__restrict__ blob, const uint4 *__restrict__ gv4,
                                               const int *__restrict__ halo_idx, int tid, int nthr) {
  const int lane = tid & 63, w0 = tid & ~63;
  // (1) halo row numbers of the pieces this thread will gather (always issued, clamped)
  int hrow[KV];
  const int *hid = halo_idx + td.halo_off;
  const int hmax = td.nhalo > 0 ? td.nhalo - 1 : 0;
#pragma unroll
  for (int k = 0; k < KV; k++) {
    int h = ((tid + k * nthr) >> 2) - td.npts;
    h = h < 0 ? 0 : (h > hmax ? hmax : h);
    hrow[k] = ld_i32_nowait(hid + h);
  }
  // (2) the blob: CB pieces per wave, source clamped to the last valid 16 bytes
  const uint4 *b4 = blob + td.blob_off;
  const int qmax = td.blob_qw - 1;
#pragma unroll
  for (int i = 0; i < CB; i++) {
    const int q0 = w0 + i * nthr;
    const int q = q0 + lane < qmax ? q0 + lane : qmax;
    if constexpr (NT) glds16_nt(b4 + q, buf + (size_t)q0 * 16);
    else glds16(b4 + q, buf + (size_t)q0 * 16);
  }
  // (3) wait for the index loads only: the CB DMA instructions issued after them stay in flight
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(CB) : "memory");
#pragma unroll
  for (int k = 0; k < KV; k++) asm volatile("" : "+v"(hrow[k]));  // uses stay behind the wait
  // (4) var rows: own rows by position, halo rows by number; 4 lanes per 64-byte row
  unsigned char *vbuf = buf + (size_t)CB * nthr * 16;
#pragma unroll
  for (int k = 0; k < KV; k++) {
    const int q = tid + k * nthr, r = q >> 2;
    const int row = r < td.npts ? td.pstart + r : hrow[k];
    glds16(gv4 + (size_t)row * 4 + (q & 3), vbuf + (size_t)(w0 + k * nthr) * 16);
  }
}

// ALIAS: the store slab of the last phase lies ON the var rows (as in the fused pass: the waves meet once between the
// face loop and the stores) instead of behind them -- 5.25 KiB less per tile, which is a workgroup per CU more at the
// capacities it is used for (<5,3>: 32 KiB, five instead of four; <6,4>: 40 KiB, four instead of three)
template <int LPP, bool NT, int CB, int KV, bool ALIAS = false>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(LPP == 8 ? GG_WAVES_EU : LPP == 4 ? 4 : 2)))
void gg_gradient_dma_kernel(
    const cfdp_tile_desc *__restrict__ tiles, int tile_begin, const uint4 *__restrict__ blob,
    const int *__restrict__ halo_idx, const double *__restrict__ var /*[nall][8]*/,
    gg_grad_view gout, int dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = tile_begin + xcd_tile(blockIdx.x, gridDim.x);
  const cfdp_tile_desc td = tiles[t];
  const int tid = threadIdx.x, nthr = blockDim.x;
  dma_stage_tile<NT, CB, KV>(smem, td, blob, reinterpret_cast<const uint4 *>(var), halo_idx, tid, nthr);
  __syncthreads();  // vmcnt(0) + barrier: every wave's pieces have landed
  const int var_off = CB * nthr * 16;
  double *stage = reinterpret_cast<double *>(smem + (size_t)(ALIAS ? CB : CB + KV) * nthr * 16);
  grad_tile_compute<LPP, NT, ALIAS>(smem, td, tid, gout, stage, dbg, var_off);
}

// LDS-DMA: global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPRs, asynchronous (vmcnt)
__device__ __forceinline__ void glds16(const uint4 *src, unsigned char *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
// the same with the non-temporal policy (aux = 2): for the tile blobs, which are read exactly
// once per launch and would otherwise push the var rows (re-read as halo rows by the
// neighbouring tiles) out of the XCD's L2
// system scope (sc0 sc1): ghost rows may have been written by ANOTHER device straight into this
// device's memory (xGMI write + notify exchange); a line of them left in this XCD's L2 from the
// previous iteration would be stale, so these loads go past the L2
__device__ __forceinline__ void glds16_sys(const uint4 *src, unsigned char *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 17);
}
__device__ __forceinline__ void glds16_nt(const uint4 *src, unsigned char *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 2);
}

// ---------------------------------------------------------------------------------- flux
// LPP lanes share a point and split its incidence list; partial sums are combined with
// wave shuffles in a fixed order (deterministic).
// The per-tile flux arithmetic.  `smem` holds the tile blob, `g_l` the rows of own + halo points, GS doubles
// apart, each starting with the six numbers of its velocity-gradient block the stress needs (A1, gg_kernels.h).
//
// The viscous stress is linear in the velocity gradient, and the face value is the mean of the
// two ends (src/flux.c:139-173): flux = -stress(0.5*(g0+g1)).n = (P(g0) + P(g1)).n with
// P(g) = -0.5*stress(g).  P (6 doubles) is computed ONCE per staged row, in place, instead of
// once per incidence: 6 adds + 9 FMAs per incidence instead of ~48 fp64 operations -- the fp64
// vector rate, not LDS or HBM, bounded this loop (measured: the flux phase cost 64 us of the
// 331-us fused pass on the 128^3 mesh).  Differs from the reference's association by round-off.
// P(g) = -stress(g) / 2 of one row's 3x3 velocity-gradient block (src/flux.c:125,139-173): the ONE statement of it, so
// that every kernel form produces the same bits
// Input: the six numbers the stress needs -- the diagonal of the velocity-gradient block and its three symmetric sums,
// s = [dvx_dx, dvy_dy, dvz_dz, dvx_dy + dvy_dx, dvx_dz + dvz_dx, dvy_dz + dvz_dy] -- which is what the first 48 bytes
// of every row on the device hold (gg_a_encode: owned rows and, sent that way, ghost rows alike).
__device__ __forceinline__ void half_stress(const double (&s)[6], double (&p)[6]) {
  const double mue_eff = 1.0, lambda = -2.0 / 3.0 * mue_eff;  // src/flux.c:125,163
  const double dvx_dx = s[0], dvy_dy = s[1], dvz_dz = s[2];
  const double sts_xx = lambda * (dvy_dy + dvz_dz - 2.0 * dvx_dx);
  const double sts_yy = lambda * (dvx_dx + dvz_dz - 2.0 * dvy_dy);
  const double sts_zz = lambda * (dvx_dx + dvy_dy - 2.0 * dvz_dz);
  const double sts_xy = mue_eff * s[3];
  const double sts_xz = mue_eff * s[4];
  const double sts_yz = mue_eff * s[5];
  p[0] = -0.5 * sts_xx; p[1] = -0.5 * sts_xy; p[2] = -0.5 * sts_xz;
  p[3] = -0.5 * sts_yy; p[4] = -0.5 * sts_yz; p[5] = -0.5 * sts_zz;
}


// GS: the staged rows are GS doubles apart (10: whole 80-byte rows; 6: their first 48 bytes) -- either way a row starts
// with the six numbers half_stress takes.  PRE = false (the timing experiment of EXPERIMENTS.md D.2 only): the rows are
// taken as P as they stand
template <int LPP, bool REFMODE, int GS = 10, bool PRE = true>
__device__ __forceinline__ void flux_tile_compute(const unsigned char *smem, double *g_l,
                                                  const cfdp_tile_desc &td, const int *__restrict__ hid,
                                                  int tid, int nthr, double *__restrict__ flux, int nown) {
  const int npts = td.npts;
  if constexpr (PRE) {
    const int nrows = npts + td.nhalo;
    for (int r = tid; r < nrows; r += nthr) {
      double *g = g_l + r * GS;
      double s[6], p[6];
#pragma unroll
      for (int c = 0; c < 6; c++) s[c] = g[c];
      half_stress(s, p);
#pragma unroll
      for (int c = 0; c < 6; c++) g[c] = p[c];
    }
    __syncthreads();
  }
  const int li = tid / LPP, sub = tid % LPP;
  const bool active = li < npts;
  const int plane = (td.nfaces * 8 + 15) & ~15;
  const int inc_bytes = (td.ninc * 4 + 15) & ~15;
  const double *fnx = reinterpret_cast<const double *>(smem);
  const double *fny = reinterpret_cast<const double *>(smem + plane);
  const double *fnz = reinterpret_cast<const double *>(smem + 2 * plane);
  const uint32_t *inc = reinterpret_cast<const uint32_t *>(smem + 3 * plane);
  const uint32_t *ioff = reinterpret_cast<const uint32_t *>(smem + 3 * plane + inc_bytes);

  double f0 = 0.0, f1 = 0.0, f2 = 0.0;
  // (long lists in chunks, as in grad_tile_compute: helper lane groups behind the tile's points take the further chunks)
  const gg_helpers hp = tile_helpers(smem, td, plane, inc_bytes);
  const bool helper = !active && li < npts + hp.n;
  int src = li, chunk = 0;
  if (helper) {
    const uint32_t hw = hp.tab[li - npts];
    src = (int)(hw & 0xFFFFu);
    chunk = (int)(hw >> 16);
  }
  int ks = 0, ke = 0, ks0 = 0, ke0 = 0, nchunks = 1;
  if (active || helper) {
    list_chunk(ioff, src, chunk, ks, ke, ks0, ke0, nchunks);
    double ps[6];
#pragma unroll
    for (int c = 0; c < 6; c++) ps[c] = g_l[src * GS + c];
    // U of this lane's incidences per batch: all incidence words, then all operands, then the
    // FMAs, so the lane pays the LDS round trips once per batch (as in grad_batch).  With few
    // lanes per point a lane has ~4 incidences and runs in a workgroup of few waves: U = 4.
    constexpr int U = LPP <= 4 ? GG_FLUX_BATCH : 1;
    for (int k = ks + sub; k < ke; k += U * LPP) {
      uint32_t w[U];
      bool on[U];
#pragma unroll
      for (int i = 0; i < U; i++) {
        on[i] = k + i * LPP < ke;
        w[i] = inc[on[i] ? k + i * LPP : k];
      }
      double sx[U], sy[U], sz[U], pn[U][6];
#pragma unroll
      for (int i = 0; i < U; i++) {
        const int nbr = (int)(w[i] & 0xFFFFu), f = (int)((w[i] >> 16) & 0x7FFFu);
        if (REFMODE && !(w[i] >> 31)) {
          // reference 1-thread semantics (src/flux.c:177-182 with the class numbering of
          // src/rangelist.c:719-736): the p0 end only receives +flux when p1 is a ghost
          const bool nbr_ghost = nbr >= npts && hid[nbr - npts] >= nown;
          on[i] = on[i] && nbr_ghost;
        }
        sx[i] = fnx[f]; sy[i] = fny[f]; sz[i] = fnz[f];
        const double *p = g_l + nbr * GS;
#pragma unroll
        for (int c = 0; c < 6; c++) pn[i][c] = p[c];
      }
#pragma unroll
      for (int i = 0; i < U; i++) {
        if (!on[i]) continue;
        // the p1 end subtracts (src/flux.c:184-188): flip the normal (bit 31 of w = sign bit)
        const int sgn = (int)(w[i] & 0x80000000u);
        const double nx = __hiloint2double(__double2hiint(sx[i]) ^ sgn, __double2loint(sx[i]));
        const double ny = __hiloint2double(__double2hiint(sy[i]) ^ sgn, __double2loint(sy[i]));
        const double nz = __hiloint2double(__double2hiint(sz[i]) ^ sgn, __double2loint(sz[i]));
        const double txx = ps[0] + pn[i][0], txy = ps[1] + pn[i][1], txz = ps[2] + pn[i][2];
        const double tyy = ps[3] + pn[i][3], tyz = ps[4] + pn[i][4], tzz = ps[5] + pn[i][5];
        f0 = fma(txx, nx, fma(txy, ny, fma(txz, nz, f0)));
        f1 = fma(txy, nx, fma(tyy, ny, fma(tyz, nz, f1)));
        f2 = fma(txz, nx, fma(tyz, ny, fma(tzz, nz, f2)));
      }
    }
  }
  // combine the LPP partial sums (lanes of one point are adjacent, LPP divides 64)
#pragma unroll
  for (int m = 1; m < LPP; m <<= 1) {
    f0 += __shfl_xor(f0, m, 64);
    f1 += __shfl_xor(f1, m, 64);
    f2 += __shfl_xor(f2, m, 64);
  }
  if (hp.n) {  // (uniform per workgroup)
    if (helper && sub == 0) {
      double *sc = hp.scratch + (li - npts) * 24;
      sc[0] = f0; sc[1] = f1; sc[2] = f2;
    }
    __syncthreads();
    if (active && sub == 0 && nchunks > 1)
      for (int h = 0; h < hp.n; h++)
        if ((int)(hp.tab[h] & 0xFFFFu) == li) {
          const double *sc = hp.scratch + h * 24;
          f0 += sc[0]; f1 += sc[1]; f2 += sc[2];
        }
    __syncthreads();  // (the gradient phase of a fused pass uses the same scratch)
  }
  if (active && sub == 0 && ke0 > ks0) {
    double *o = flux + (size_t)(td.pstart + li) * 3;
    o[0] = f0; o[1] = f1; o[2] = f2;
  }
}

template <int LPP, bool REFMODE, bool NT>
__global__ __launch_bounds__(1024) void gg_flux_kernel(
    const cfdp_tile_desc *__restrict__ tiles, int tile_begin, const uint4 *__restrict__ blob,
    const int *__restrict__ halo_idx, const double *__restrict__ gradA /*[nown][10]*/,
    const double *__restrict__ ghost /*[nghost][21]*/, double *__restrict__ flux /*[nown][3]*/, int nown) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = tile_begin + xcd_tile(blockIdx.x, gridDim.x);
  const cfdp_tile_desc td = tiles[t];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int npts = td.npts, nhalo = td.nhalo;

  uint4 *s4 = reinterpret_cast<uint4 *>(smem);
  const uint4 *b4 = blob + td.blob_off;
  for (int q = tid; q < td.blob_qw; q += nthr) s4[q] = ld_blob<NT>(&b4[q]);
  // the six numbers of the velocity-gradient block the stress needs = A1 of an owned row, the first 6 doubles of a ghost row
  double *g_l = reinterpret_cast<double *>(s4 + td.blob_qw);  // [(npts+nhalo)][10]
  const int *hid = halo_idx + td.halo_off;
  for (int q = tid; q < (npts + nhalo) * 6; q += nthr) {
    const int r = q / 6, c = q - 6 * r;
    const int row = r < npts ? td.pstart + r : hid[r - npts];
    g_l[r * 10 + c] = row < nown ? gradA[(size_t)row * 6 + c]
                                 : __hip_atomic_load(&ghost[(size_t)(row - nown) * 21 + c], __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_SYSTEM);  // see glds16_sys
  }
  __syncthreads();

  flux_tile_compute<LPP, REFMODE>(smem, g_l, td, hid, tid, nthr, flux, nown);
}

// one workgroup per tile, fixed-count LDS-DMA staging (see gg_gradient_dma_kernel): the blob as CB
// pieces per wave, the gradient rows as KV pieces per wave -- 3 pieces (48 bytes: A1) per row: own
// rows by position and owned halo rows by number from A1, ghost halo rows = the first 48
// bytes of their 168-byte row in the ghost block
// WAIT: the flux that closes a batch of exchanging iterations -- its boundary tiles wait for the rows of the last exchange
// themselves (as the boundary tiles of a pushing pass do), so that no wait kernel stands between the last pass and it
template <int LPP, bool REFMODE, bool NT, int CB, int KV, bool WAIT = false>
__global__ __launch_bounds__(1024) void gg_flux_dma_kernel(
    const cfdp_tile_desc *__restrict__ tiles, int tile_begin, const uint4 *__restrict__ blob,
    const int *__restrict__ halo_idx, const double *__restrict__ gradA /*[nown][10]*/,
    const double *__restrict__ ghost /*[nghost][21]*/, double *__restrict__ flux /*[nown][3]*/, int nown, gg_push_args pa) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = tile_begin + xcd_tile(blockIdx.x, gridDim.x);
  if constexpr (WAIT)
    if (pa.tile_off && t < pa.nbtiles) wait_previous_exchange(pa, t, (int)threadIdx.x, exchanges_so_far(pa, t));
  const cfdp_tile_desc td = tiles[t];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, w0 = tid & ~63;
  const int *hid = halo_idx + td.halo_off;
  // (1) halo row numbers for this thread's pieces (always issued, clamped)
  int hrow[KV], part[KV], rloc[KV];
  const int hmax = td.nhalo > 0 ? td.nhalo - 1 : 0;
#pragma unroll
  for (int k = 0; k < KV; k++) {
    const int q = tid + k * nthr;
    rloc[k] = q / 3;
    part[k] = q - 3 * rloc[k];
    int h = rloc[k] - td.npts;
    h = h < 0 ? 0 : (h > hmax ? hmax : h);
    hrow[k] = ld_i32_nowait(hid + h);
  }
  // (2) blob pieces
  const uint4 *b4 = blob + td.blob_off;
  const int qmax = td.blob_qw - 1;
#pragma unroll
  for (int i = 0; i < CB; i++) {
    const int q0 = w0 + i * nthr;
    const int q = q0 + lane < qmax ? q0 + lane : qmax;
    if constexpr (NT) glds16_nt(b4 + q, smem + (size_t)q0 * 16);
    else glds16(b4 + q, smem + (size_t)q0 * 16);
  }
  // (3) only the index loads are awaited; the CB blob pieces stay in flight
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(CB) : "memory");
#pragma unroll
  for (int k = 0; k < KV; k++) asm volatile("" : "+v"(hrow[k]));  // uses stay behind the wait
  // (4) gradient rows, 48 bytes each
  unsigned char *gbuf = smem + (size_t)CB * nthr * 16;
  const unsigned char *abytes = reinterpret_cast<const unsigned char *>(gradA);
  const unsigned char *hbytes = reinterpret_cast<const unsigned char *>(ghost);
#pragma unroll
  for (int k = 0; k < KV; k++) {
    const int row = rloc[k] < td.npts ? td.pstart + rloc[k] : hrow[k];
    if (row < nown)
      glds16(reinterpret_cast<const uint4 *>(abytes + (size_t)row * 48 + part[k] * 16), gbuf + (size_t)(w0 + k * nthr) * 16);
    else
      glds16_sys(reinterpret_cast<const uint4 *>(hbytes + (size_t)(row - nown) * 168 + part[k] * 16),
                 gbuf + (size_t)(w0 + k * nthr) * 16);
  }
  __syncthreads();
  flux_tile_compute<LPP, REFMODE, 6>(smem, reinterpret_cast<double *>(gbuf), td, hid, tid, nthr, flux, nown);
}

// ------------------------------------------------------------------- fused iteration kernel
// flux(i) and gradients(i+1) of a tile in ONE pass: both face loops read the same tile blob
// (normals + incidence lists = more than half of either kernel's HBM traffic), so a run of
// iterations streams it once per iteration instead of twice.  grad is double-buffered: the flux
// phase reads A1 / the ghost block of the buffer iteration i wrote (its halo exchange has
// completed), the gradient phase writes the other buffer.  The results are those of the two
// separate kernels, bit for bit (same per-tile arithmetic, flux_tile_compute /
// grad_tile_compute).  4 lanes per point in both phases; fixed-count LDS-DMA staging as above:
// LDS image [blob: CB][var rows: KV][gradient rows (their first 48 bytes each): KG] x nthr x 16 bytes; the
// store slab of the gradient phase reuses the gradient-row region once the flux phase is done.
template <bool REFMODE, bool NT, int CB, int KV, int KG>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4)))
void gg_fused_dma_kernel(
    const cfdp_tile_desc *__restrict__ tiles, int tile_begin, const uint4 *__restrict__ blob,
    const int *__restrict__ halo_idx, const double *__restrict__ var /*[nall][8]*/,
    const double *__restrict__ gradA_old /*[nown][10]*/, const double *__restrict__ ghost_old /*[nghost][21]*/,
    double *__restrict__ flux /*[nown][3]*/, int nown,
    gg_grad_view gnew, int dbg,
    gg_push_args pa) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int LPP = 4;
  const int t = tile_begin + xcd_tile_bfirst(blockIdx.x, gridDim.x, pa.tile_off && tile_begin == 0 ? pa.nbtiles : 0,
                                            (dbg & GG_DBG_REVERSE) != 0);
  const cfdp_tile_desc td = tiles[t];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, w0 = tid & ~63;
  const int iter0 = pa.tile_off && t < pa.nbtiles ? exchanges_so_far(pa, t) : 0;  // uniform: scalar loads
  wait_previous_exchange(pa, t, tid, iter0);  // before any ghost row is requested
  unsigned long long pfirst = ~0ull;  // see gg_fused_split_kernel
  if (pa.tile_off && t < pa.nbtiles) pfirst = ld_u64_nowait(pa.pt_first + (size_t)t * pa.pt_stride + (tid >> 2));
  const int *hid = halo_idx + td.halo_off;
  const int hmax = td.nhalo > 0 ? td.nhalo - 1 : 0;
  // (1) halo row numbers of this thread's var pieces (4 per row) and gradient pieces (5 per row)
  int hv[KV], hg[KG], part[KG], rloc[KG];
#pragma unroll
  for (int k = 0; k < KV; k++) {
    int h = ((tid + k * nthr) >> 2) - td.npts;
    h = h < 0 ? 0 : (h > hmax ? hmax : h);
    hv[k] = ld_i32_nowait(hid + h);
  }
#pragma unroll
  for (int k = 0; k < KG; k++) {
    const int q = tid + k * nthr;
    rloc[k] = q / 3;
    part[k] = q - 3 * rloc[k];
    int h = rloc[k] - td.npts;
    h = h < 0 ? 0 : (h > hmax ? hmax : h);
    hg[k] = ld_i32_nowait(hid + h);
  }
  // (2) the blob, once for both face loops
  const uint4 *b4 = blob + td.blob_off;
  const int qmax = td.blob_qw - 1;
#pragma unroll
  for (int i = 0; i < CB; i++) {
    const int q0 = w0 + i * nthr;
    const int q = q0 + lane < qmax ? q0 + lane : qmax;
    if constexpr (NT) glds16_nt(b4 + q, smem + (size_t)q0 * 16);
    else glds16(b4 + q, smem + (size_t)q0 * 16);
  }
  // (3) only the index loads are awaited; the CB blob pieces stay in flight
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(CB) : "memory");
#pragma unroll
  for (int k = 0; k < KV; k++) asm volatile("" : "+v"(hv[k]));
#pragma unroll
  for (int k = 0; k < KG; k++) asm volatile("" : "+v"(hg[k]));
  // (4) var rows and gradient rows
  unsigned char *vbuf = smem + (size_t)CB * nthr * 16;
  unsigned char *gbuf = vbuf + (size_t)KV * nthr * 16;
  const uint4 *gv4 = reinterpret_cast<const uint4 *>(var);
#pragma unroll
  for (int k = 0; k < KV; k++) {
    const int q = tid + k * nthr, r = q >> 2;
    const int row = r < td.npts ? td.pstart + r : hv[k];
    glds16(gv4 + (size_t)row * 4 + (q & 3), vbuf + (size_t)(w0 + k * nthr) * 16);
  }
  const unsigned char *abytes = reinterpret_cast<const unsigned char *>(gradA_old);
  const unsigned char *hbytes = reinterpret_cast<const unsigned char *>(ghost_old);
#pragma unroll
  for (int k = 0; k < KG; k++) {
    const int row = rloc[k] < td.npts ? td.pstart + rloc[k] : hg[k];
    if (row < nown)
      glds16(reinterpret_cast<const uint4 *>(abytes + (size_t)row * 48 + part[k] * 16), gbuf + (size_t)(w0 + k * nthr) * 16);
    else
      glds16_sys(reinterpret_cast<const uint4 *>(hbytes + (size_t)(row - nown) * 168 + part[k] * 16),
                 gbuf + (size_t)(w0 + k * nthr) * 16);
  }
  __syncthreads();
  double *push_row = nullptr;
  {
    asm volatile("" : "+v"(pfirst));
    const int pslot = (int)(unsigned)pfirst, prow = (int)(pfirst >> 32);
    if (pslot >= 0) push_row = pa.dst[pslot] + (size_t)prow * 21;
  }
  flux_tile_compute<LPP, REFMODE, 6>(smem, reinterpret_cast<double *>(gbuf), td, hid, tid, nthr, flux, nown);
  grad_tile_compute<LPP, NT, true>(smem, td, tid, gnew, reinterpret_cast<double *>(gbuf), dbg,
                                   CB * nthr * 16, push_row, pa.tile_off && t < pa.nbtiles);
  push_tile_rows(pa, t, tid, nthr, td, gnew);
  push_tile_done(pa, t, tid, iter0);
}

// (one per translation unit: only the diagnostic library ever stamps, and it sets its own -- gg_diag.hip)
static __device__ unsigned long long *gg_stamp_buf = nullptr;
__device__ __forceinline__ void gg_stamp(int dbg, int tile, int slot) {
  if ((dbg & GG_DBG_STAMP) && threadIdx.x == 0 && gg_stamp_buf) gg_stamp_buf[(size_t)tile * 8 + slot] = __builtin_amdgcn_s_memtime();
}
// the same per WAVE (lane 0 of each of the 4 waves), behind the per-tile stamps: [ntiles*8 + (tile*4 + wave)*4 + slot]
__device__ __forceinline__ void gg_stamp_wave(int dbg, int ntiles, int tile, int slot) {
  if ((dbg & GG_DBG_STAMP) && (threadIdx.x & 63) == 0 && gg_stamp_buf)
    gg_stamp_buf[(size_t)ntiles * 8 + ((size_t)tile * 4 + (threadIdx.x >> 6)) * 4 + slot] = __builtin_amdgcn_s_memtime();
}

// Phase-split form of the fused pass: ONE row region of LDS holds the gradient rows during the flux
// phase and the var rows during the gradient phase, so a tile occupies CB + KX pieces per thread instead
// of CB + KV + KG: 32 KiB instead of 44 KiB for 64-point tiles = FIVE workgroups per CU.  The var rows
// are requested together with the gradient rows, into registers, and wait out the flux phase there.
// LISTED: the fixed-stride row lists exist (gg_args::rowlist); PUSH: an exchange rides in the pass (the boundary
// tiles wait for the previous exchange, push their rows, notify) -- both compile-time, so the pass that runs one
// partition on one GPU carries neither the other path's code nor its registers
// DIAG: 0 = the timed kernel; 1 = phase stamps (tools/phase_stamps.py); 2 = data movement only: every load and every
// store of the pass, neither face loop (cfdp_gpu_time_fused_movement: the floor bench.py reports beside the pass)
// Two capacities are instantiated -- the two budget levels of the tiler (cfdp_tile_class, cfdproxy_host.h): CB = 5, KV = KG =
// KX = 3 (blob <= 20 KiB, up to 192 staged rows -- every tile of the lattice plans: 32 KiB, FIVE workgroups per CU) and CB = 6,
// KV = 4, KG = 3, KX = 4 (blob <= 24 KiB, up to 256 staged rows -- the full tiles of an unstructured mesh: 40 KiB, four per
// CU).  The flux phase stages the first 48 bytes of every row (A1: the six numbers the stress needs).
template <bool REFMODE, bool NT, int CB, int KV, int KG, int KX, int DIAG = 0, bool LISTED = true, bool PUSH = true>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4)))
void gg_fused_split_kernel(
    const cfdp_tile_desc *__restrict__ tiles, int tile_begin, const uint4 *__restrict__ blob,
    const int *__restrict__ halo_idx, const int *__restrict__ rowlist, int rl_stride, const double *__restrict__ var /*[nall][8]*/,
    const double *__restrict__ gradA_old /*[nown][10]*/, const double *__restrict__ ghost_old /*[nghost][21]*/,
    double *__restrict__ flux /*[nown][3]*/, int nown,
    gg_grad_view gnew, int dbg,
    gg_push_args pa) {
  static_assert(KX >= KV && KX >= KG, "the shared row region must hold either set of rows");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int LPP = 4;
  constexpr bool STAMP = DIAG == 1;
  const int t = tile_begin + xcd_tile_bfirst(blockIdx.x, gridDim.x, PUSH && pa.tile_off && tile_begin == 0 ? pa.nbtiles : 0,
                                            (dbg & GG_DBG_REVERSE) != 0);
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, w0 = tid & ~63;
  if constexpr (STAMP) gg_stamp(dbg, t, 0);
  if constexpr (STAMP)
  if ((dbg & GG_DBG_STAMP) && tid == 0 && gg_stamp_buf)  // which CU: HW_ID (cu, sh, se) and XCC_ID
    gg_stamp_buf[(size_t)t * 8 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                                     (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
  int iter0 = 0;
  gg_wait_probe wprobe;
  bool waiting = false;  // uniform: this boundary tile owes the previous exchange a wait (or split mode its invalidate)
  unsigned long long pfirst = ~0ull;  // {partner slot or -1, row}: where this lane's point goes first (boundary tiles)
  if constexpr (PUSH) {
    if (pa.tile_off && t < pa.nbtiles) iter0 = exchanges_so_far(pa, t);  // uniform: scalar loads
    waiting = !(dbg & 0x100) && wait_wanted(pa, t);
    if (waiting) wait_first_poll(pa, t, tid, wprobe);  // the partners' words travel with the staging loads below
    if (pa.tile_off && t < pa.nbtiles) pfirst = ld_u64_nowait(pa.pt_first + (size_t)t * pa.pt_stride + (tid >> 2));
  }
  int hv[KV], hg[KG], part[KG], rloc[KG];
  constexpr int PPR = 3;  // 16-byte pieces staged per gradient row: its first 48 bytes (A1)
  // with a fixed-stride row list the row numbers are requested before the descriptor is even here:
  // descriptor -> blob and row list -> rows are two chains of two round trips instead of one of three
  constexpr bool listed = LISTED;
  if (listed) {
    const int *rl = rowlist + (size_t)t * rl_stride;
    const int rlast = rl_stride - 1;
#pragma unroll
    for (int k = 0; k < KV; k++) {
      const int r = (tid + k * nthr) >> 2;
      hv[k] = ld_i32_nowait(rl + (r < rlast ? r : rlast));
    }
#pragma unroll
    for (int k = 0; k < KG; k++) {
      const int q = tid + k * nthr;
      rloc[k] = q / PPR;
      part[k] = q - PPR * rloc[k];
      hg[k] = ld_i32_nowait(rl + (rloc[k] < rlast ? rloc[k] : rlast));
    }
  }
  const cfdp_tile_desc td = tiles[t];
  const int *hid = halo_idx + td.halo_off;
  const int hmax = td.nhalo > 0 ? td.nhalo - 1 : 0;
  if (!listed) {
#pragma unroll
    for (int k = 0; k < KV; k++) {
      int h = ((tid + k * nthr) >> 2) - td.npts;
      h = h < 0 ? 0 : (h > hmax ? hmax : h);
      hv[k] = ld_i32_nowait(hid + h);
    }
#pragma unroll
    for (int k = 0; k < KG; k++) {
      const int q = tid + k * nthr;
      rloc[k] = q / PPR;
      part[k] = q - PPR * rloc[k];
      int h = rloc[k] - td.npts;
      h = h < 0 ? 0 : (h > hmax ? hmax : h);
      hg[k] = ld_i32_nowait(hid + h);
    }
  }
  const uint4 *b4 = blob + td.blob_off;
  const int qmax = td.blob_qw - 1;
#pragma unroll
  for (int i = 0; i < CB; i++) {
    const int q0 = w0 + i * nthr;
    const int q = q0 + lane < qmax ? q0 + lane : qmax;
    if constexpr (NT) glds16_nt(b4 + q, smem + (size_t)q0 * 16);
    else glds16(b4 + q, smem + (size_t)q0 * 16);
  }
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(CB) : "memory");
  if constexpr (STAMP) gg_stamp(dbg, t, 1);  // descriptor + row numbers are here
#pragma unroll
  for (int k = 0; k < KV; k++) asm volatile("" : "+v"(hv[k]));
#pragma unroll
  for (int k = 0; k < KG; k++) asm volatile("" : "+v"(hg[k]));
  if constexpr (PUSH)
    if (waiting) wait_check(pa, t, tid, iter0, wprobe);  // before any ghost row is requested
  double *push_row = nullptr;  // this lane's point, in its first partner's arena (the slice pointer travels with the rows)
  if constexpr (PUSH) {
    asm volatile("" : "+v"(pfirst));
    const int pslot = (int)(unsigned)pfirst, prow = (int)(pfirst >> 32);
    if (pslot >= 0 && !(dbg & 0x200)) push_row = pa.dst[pslot] + (size_t)prow * 21;
  }
  unsigned char *xbuf = smem + (size_t)CB * nthr * 16;  // the shared row region
  const unsigned char *abytes = reinterpret_cast<const unsigned char *>(gradA_old);
  const unsigned char *hbytes = reinterpret_cast<const unsigned char *>(ghost_old);
#pragma unroll
  for (int k = 0; k < KG; k++) {
    const int row = !listed && rloc[k] < td.npts ? td.pstart + rloc[k] : hg[k];
    if (row < nown)
      glds16(reinterpret_cast<const uint4 *>(abytes + (size_t)row * 48 + part[k] * 16), xbuf + (size_t)(w0 + k * nthr) * 16);
    else
      glds16_sys(reinterpret_cast<const uint4 *>(hbytes + (size_t)(row - nown) * 168 + part[k] * 16),
                 xbuf + (size_t)(w0 + k * nthr) * 16);
  }
  // the var rows travel with the gradient rows, into registers (4 VGPRs per piece): they wait out the
  // flux phase there and drop into the row region behind it, so the second gather costs no round trip
  const uint4 *gv4 = reinterpret_cast<const uint4 *>(var);
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 vr[KV];
#pragma unroll
  for (int k = 0; k < KV; k++) {
    const int q = tid + k * nthr, r = q >> 2;
    const int row = !listed && r < td.npts ? td.pstart + r : hv[k];
    vr[k] = *reinterpret_cast<const u32x4 *>(gv4 + (size_t)row * 4 + (q & 3));
  }
  if constexpr (STAMP) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    gg_stamp_wave(dbg, (int)gridDim.x, t, 0);  // this wave's own pieces have landed
  }
  __syncthreads();
  if constexpr (STAMP) gg_stamp(dbg, t, 2);  // blob + gradient rows (+ var rows in registers) have landed
  if constexpr (DIAG == 2) {
    if (tid < td.npts * 3) flux[(size_t)td.pstart * 3 + tid] = 0.0;  // the flux rows leave as they do in the real pass
  } else {  // (DIAG == 3: timing experiment, values wrong -- the pass over the staged rows and its barrier skipped)
    flux_tile_compute<LPP, REFMODE, 6, DIAG != 3>(smem, reinterpret_cast<double *>(xbuf), td, hid, tid, nthr, flux, nown);
  }
  if constexpr (STAMP) gg_stamp_wave(dbg, (int)gridDim.x, t, 1);  // this wave is through its flux phase
  __syncthreads();  // every wave is done with the gradient rows: the region takes the var rows
  if constexpr (STAMP) gg_stamp(dbg, t, 3);  // flux phase done
#pragma unroll
  for (int k = 0; k < KV; k++) *reinterpret_cast<u32x4 *>(xbuf + (size_t)(tid + k * nthr) * 16) = vr[k];
  __syncthreads();  // vmcnt(0) + barrier
  if constexpr (STAMP) gg_stamp(dbg, t, 4);  // var rows in place
  grad_tile_compute<LPP, NT, true, DIAG == 2>(smem, td, tid, gnew, reinterpret_cast<double *>(xbuf), dbg,
                                              CB * nthr * 16, push_row, PUSH && pa.tile_off && t < pa.nbtiles);
  if constexpr (STAMP) gg_stamp_wave(dbg, (int)gridDim.x, t, 2);  // this wave is through its gradient phase (stores issued)
  if constexpr (STAMP) gg_stamp(dbg, t, 5);  // gradient arithmetic done, row stores issued (wave 0)
  if constexpr (STAMP) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    gg_stamp(dbg, t, 6);  // wave 0's stores acknowledged
  }
  if constexpr (PUSH) {
    if (!(dbg & 0x200)) push_tile_rows(pa, t, tid, nthr, td, gnew);
    if (!(dbg & 0x400)) push_tile_done(pa, t, tid, iter0, dbg);
  }
  if constexpr (STAMP) gg_stamp_wave(dbg, (int)gridDim.x, t, 3);  // this wave is through the tile (rows pushed, tile counted)
}


#endif
