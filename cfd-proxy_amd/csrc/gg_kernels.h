// gg_kernels.h -- launch wrappers of the gfx950 kernels (internal to libcfdproxy_hip.so).
#ifndef CFDP_GG_KERNELS_H
#define CFDP_GG_KERNELS_H

#include <hip/hip_runtime.h>

#include "cfdproxy_hip.h"   // CFDP_IPC_HEADER_BYTES
#include "cfdproxy_host.h"  // cfdp_tile_desc

// grad in HBM: one allocation of nall*21 doubles (round 5):
//     [A1: nown x 6][ghost rows: nghost x 21, message order][A2: nown x 4][B: nown x 11]
// With g[0..8] the row-major 3x3 block d(vx,vy,vz)/d(x,y,z) of a row and g[9] its tenth double, a row is kept -- on the
// device and on the wire alike -- as
//     e = [ g0  g4  g8  g1+g3  g2+g6  g5+g7 | g3  g6  g7  g9 | d10 .. d20 ]
//            A1: all the flux loop needs      A2              B (untouched)
// The viscous stress needs the diagonal and the three symmetric sums of the velocity-gradient block, nothing else
// (src/flux.c:139-173).  A1 of the owned points is one array of contiguous 48-byte rows: the flux loop stages 3 instead
// of 5 16-byte pieces per row through the CU's memory pipe and into LDS (a 32-KiB tile image instead of 36: five
// workgroups per CU) and streams 48 instead of 80 bytes per owned point from memory (EXPERIMENTS.md D.2).  The sums are
// formed with the very addition the stress formula performs, so the flux is bit for bit what it was.  The three upper
// off-diagonals are handed out as (g1+g3)-g3, (g2+g6)-g6, (g5+g7)-g7: within one rounding of the sum of the two -- 1e-16 of
// the scale the 1e-10 tolerance is written against -- and the SAME value wherever a row is seen: a ghost row (168 bytes,
// whole, [e0..e9 | B]) is a bit copy of its owner's stored row, and rows leave the device through gg_a_decode
// (cfdp_gpu_get_grad / cfdp_sync_fields_to_host), the one statement of it.
__host__ __device__ inline void gg_a_encode(const double *g, double *e) {  // g, e: 10 doubles, may not alias
  e[0] = g[0]; e[1] = g[4]; e[2] = g[8];
  e[3] = g[1] + g[3]; e[4] = g[2] + g[6]; e[5] = g[5] + g[7];
  e[6] = g[3]; e[7] = g[6]; e[8] = g[7];
  e[9] = g[9];
}
__host__ __device__ inline double gg_a_decode(const double *e, int c) {  // component c (0..9) of the stored row e[0..9]
  switch (c) {
    case 0: return e[0];
    case 1: return e[3] - e[6];
    case 2: return e[4] - e[7];
    case 3: return e[6];
    case 4: return e[1];
    case 5: return e[5] - e[8];
    case 6: return e[7];
    case 7: return e[8];
    case 8: return e[2];
    default: return e[9];
  }
}
struct gg_grad_view {
  double *a, *ghost, *a2, *b;  // a = A1
  static gg_grad_view of(double *base, int nown, int nall) {
    gg_grad_view v;
    v.a = base;
    v.ghost = base + (size_t)nown * 6;
    v.a2 = v.ghost + (size_t)(nall - nown) * 21;
    v.b = v.a2 + (size_t)nown * 4;
    return v;
  }
};

// xGMI write + notify done by the tiles themselves (fused pass): per boundary tile t the entries
// [tile_off[t], tile_off[t+1]) name its send rows -- ent = tile-local point | partner slot << 16,
// ent_row = row in that partner's landing slice; dst[slot] = that slice (this iteration's parity).
// done[0] counts finished boundary tiles (the last one moves hdr[GG_IPC_ITER]); tile_off == nullptr: no push.
// Notification, coarse protocol (need == nullptr): the last boundary tile raises the iteration counter in EVERY
// partner's flag word.  Per-partner protocol: done[1 + s] counts the finished tiles that hold send rows for partner
// slot s, need[s] of them exist; the one that completes s raises s's flag at once; tile_mask[t] = the partner slots
// boundary tile t sends to (bit s) = the flags it waits for at the top of the next pass.
struct gg_push_args {
  const int *tile_off;
  const int *ent;
  const int *ent_row;
  double *const *dst;
  int *hdr;
  int *const *rflag;
  int *done;
  const int *need;
  const unsigned long long *tile_mask;
  // register push: pt_first[t * pt_stride + li] = {partner slot or -1, row in that partner's slice} of point li's FIRST
  // destination; the entries [tile_xoff[t], tile_off[t + 1]) are the further destinations of points sent to several partners
  const int2 *pt_first;
  const int *tile_xoff;
  int pt_stride;
  int nbtiles, nslots;
  int inv_after_flag;  // explicit buffer_inv sc0 sc1 once a tile has seen its partners' flags
  // counter notification (per-partner protocol only): a partner's flag word is a COUNTER of the boundary tiles that have
  // completed their rows for it -- raised by a fire-and-forget system-scope atomic add, nothing returns to the tile --
  // and a waiting tile compares it with (tiles of that partner per exchange) x (exchanges so far).  tile_iter[t] = the
  // exchanges boundary tile t has taken part in: read by tile t at its top, stored by tile t at its end -- nobody else
  // touches it, so the count needs no atomic and no "last tile" election
  int counters;
  int *tile_iter;
  // > 0: the boundary tiles (the only ones that read ghost rows) first wait -- bounded -- until every partner's
  // rows of the PREVIOUS exchange have arrived: the job of gg_wait_kernel done at the top of the next pass, so
  // that an iteration is ONE launch (no wait kernel, no kernel boundary behind it)
  long wait_polls;
};

struct gg_args {
  const cfdp_tile_desc *tiles;  // device copies
  const uint4 *blob;
  const int *halo_idx;
  // optional (nullptr = absent): the rows a tile stages (own rows, then halo rows, then padding that
  // repeats the last one), GG_ROW_STRIDE entries per tile -- a list whose address needs nothing but
  // the tile number, so the gathers of the fused pass do not wait for the tile descriptor first
  const int *rowlist;
  int rowlist_stride;           // entries per tile (>= the staged rows of every tile that a listed kernel runs, + 1)
  const double *var;            // [nall][8]: 7 variables + the dual volume in slot 7
  gg_grad_view grad;
  double *flux;                 // [nown][3]
  int nown;
};

hipError_t gg_launch_gradient(const gg_args &a, int lanes, int tile_begin, int ntiles,
                              int tile_points, size_t lds, int max_halo, int max_blob_qw,
                              bool nt, hipStream_t stream);
// wait != nullptr: the boundary tiles wait for the previous exchange themselves (only where gg_flux_can_wait says so)
hipError_t gg_launch_flux(const gg_args &a, int lanes, bool refmode, int tile_begin, int ntiles,
                          int tile_points, size_t lds, int max_halo, int max_blob_qw, bool nt,
                          hipStream_t stream, const gg_push_args *wait = nullptr);
bool gg_flux_can_wait(int lanes, int tile_points, int max_halo, int max_blob_qw);
// flux(i) read from a.grad + gradients(i+1) written to `gnew` in one pass over the tile blobs;
// hipErrorNotSupported when the tile sizes fit no instantiated capacity
// allow_split: the phase-split form (36 instead of 48 KiB of LDS per tile, 4 workgroups per CU) may
// be used -- not while an RCCL kernel has to squeeze in beside the interior tiles: a retiring
// workgroup then frees too little LDS for it (measured: 56 vs 52 us per overlapped iteration)
hipError_t gg_launch_fused(const gg_args &a, const gg_grad_view &gnew, bool refmode, int tile_begin,
                           int ntiles, int tile_points, int max_halo, int max_blob_qw, bool nt,
                           bool allow_split, hipStream_t stream, const gg_push_args *push = nullptr,
                           bool reverse = false);
// would gg_launch_fused run a fused kernel (not hipErrorNotSupported) for these tile sizes?
bool gg_fused_fits(int tile_points, int max_halo, int max_blob_qw);
hipError_t gg_launch_pack(const int *send_idx, int nsend, const gg_grad_view &grad, double *sendbuf,
                          hipStream_t stream);
hipError_t gg_launch_unpack(const double *recvbuf, int nrecv, const gg_grad_view &grad,
                            hipStream_t stream);
// xGMI write + notify exchange (see gg_kernels.hip): header words of a rank's IPC block
// entries per tile of the fixed-stride row lists: 208 where every tile stages <= 204 rows (the small image of the 256-thread
// fused pass: 192), 272 up to 256 rows (its large image); whole 64-byte lines either way
enum { GG_ROW_STRIDE = 208, GG_ROW_STRIDE_LARGE = 272 };
enum { GG_DONE_STRIDE = 32 };  // the completion counters of push_tile_done sit on cache lines of their own (atomics of hundreds of tiles)
// hdr (ints): partner slot s owns the cache line [s * SLOT_STRIDE, (s + 1) * SLOT_STRIDE): word 0 its arrival flag /
// counter (written by partner s only -- seven devices never store to one line), word GG_IPC_NEED_IN what that word
// advances by per exchange: the partner's boundary tiles that count towards this rank when it notifies by counters, 1
// when it stores its exchange number (written once by the partner at set-up) -- a waiting rank compares the word with
// exchanges x NEED_IN whatever ITS OWN form of notification is, so neighbours that resolved to different forms (the
// per-partner protocol depends on a rank's own partition) still understand each other; behind the slot lines [ITER] this
// rank's exchanges so far (flag notification) and [ERR .. ERR + 4] a wait gave up
enum { GG_IPC_MAXSLOTS = 48, GG_IPC_SLOT_STRIDE = 32, GG_IPC_NEED_IN = 1, GG_IPC_ITER = GG_IPC_MAXSLOTS * GG_IPC_SLOT_STRIDE,
       GG_IPC_ERR = GG_IPC_ITER + 1, GG_IPC_HDR_BYTES = 8192 };
static_assert(GG_IPC_HDR_BYTES == CFDP_IPC_HEADER_BYTES, "cfdproxy_hip.h");
static_assert((GG_IPC_ERR + 5) * 4 <= GG_IPC_HDR_BYTES, "header layout");
hipError_t gg_launch_push(const int *send_idx, int nsend, const int *slot_of_row, const int *send_off,
                          const gg_grad_view &grad, double *const *dst, hipStream_t stream);
// need != nullptr: counter notification (every partner's counter += need[s], every boundary tile's exchange count += 1)
hipError_t gg_launch_notify(int *hdr, int *const *remote_flag, int nslots, const int *need, int *tile_iter, int nbtiles, hipStream_t stream);
// tile_iter != nullptr: counter notification (wait for counter >= NEED_IN of the slot x tile_iter[0])
hipError_t gg_launch_wait(int *hdr, int nslots, long max_polls, const int *tile_iter, hipStream_t stream);
hipError_t gg_launch_poke(int *const *dst, const int *words, int offset, int n, hipStream_t stream);  // *(dst[i] + offset) = words[i], n <= 64
hipError_t gg_launch_jitter(unsigned *rng, int max_us, hipStream_t stream);  // tests: a pseudo-random idle time in front of a step
// scaled-field validation of an exchange (gg_validate_kernel): words of its device-side state block (16 ints, 8-byte
// aligned): gradient launches so far, flux fields compared, mismatching values (64 bit), first mismatch (claimed,
// iteration, seen, expected, index into flux[nown][3]), the ticket of the last-block election
enum { GG_V_ITER = 0, GG_V_CHECKS = 1, GG_V_BAD = 2, GG_V_CLAIM = 4, GG_V_FIRST_ITER = 5, GG_V_SEEN = 6, GG_V_EXPECT = 8,
       GG_V_FIRST_IDX = 10, GG_V_TICKET = 11, GG_V_WORDS = 16 };
hipError_t gg_launch_validate(double *var, int nall, const double *flux, const double *fref, const unsigned char *skip,
                              int nown, int lag, bool do_scale, int *state, hipStream_t stream);
hipError_t gg_launch_scale_var(double *var, int nall, double factor, hipStream_t stream);
hipError_t gg_launch_var_check(const double *var, const double *var0, int nall, double factor, unsigned long long *bad, hipStream_t stream);
// the kernel forms the launchers picked (calling thread): "form@first_tile+tiles ..."; see cfdp_gpu_kernel_forms
int gg_forms_take(char *buf, size_t len);
extern int gg_debug_flags;
hipError_t gg_set_stamp_buffer(unsigned long long *dev);  // diagnostics: phase stamps of the split fused pass
// the diagnostic instantiations of the fused pass live in lib/libcfdproxy_diag.so (csrc/gg_diag.hip), loaded on demand:
// nullptr when they are available, else the reason (what the diagnostic entry points of the ABI report)
const char *gg_diag_available();
extern int gg_fused_split;
extern int gg_grad_alias;

#endif
