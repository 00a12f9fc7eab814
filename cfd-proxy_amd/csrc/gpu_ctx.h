// gpu_ctx.h -- internal to libcfdproxy_hip.so: the context behind the opaque cfdp_gpu handle of include/cfdproxy_hip.h and
// the helpers its two translation units share (gpu_abi.hip: context, fields, launches, iterations, measurement;
// gpu_exchange.hip: the exchanges between ranks -- RCCL, xGMI write + notify -- and their validation).
#ifndef CFDP_GPU_CTX_H
#define CFDP_GPU_CTX_H

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <rccl/rccl.h>  // types and prototypes only: the library is resolved at run time (cfdp_rccl_load)

#include "cfdproxy_hip.h"
#include "gg_kernels.h"

namespace cfdp_detail __attribute__((visibility("hidden"))) {
int fail(const char *fmt, ...);  // sets the calling thread's cfdp_gpu_last_error() text, returns 1
}
using cfdp_detail::fail;

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail("%s failed: %s [%s:%d]", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// hipMemset and device-to-device hipMemcpy run on the null stream and return BEFORE the bytes are there; the context's
// streams are non-blocking (no implicit order with the null stream), so work enqueued on them right afterwards may run
// first -- and a late memset / copy then lands on top of live data.  (Found as a once-in-dozens failure on a fresh box,
// where the first use of the copy engine is slow: DESIGN appendix C.5.)  These forms wait.
// (CFDP_EXP_ASYNC_SETUP_COPIES=1, an experiment switch: WITHOUT the wait -- the defect itself, kept so that
// test_setup_copies_are_complete_before_the_first_kernel has something to fail on)
static inline bool cfdp_setup_copies_wait() {
  static const bool wait = !(cfdp_experiment_getenv("CFDP_EXP_ASYNC_SETUP_COPIES") && atoi(cfdp_experiment_getenv("CFDP_EXP_ASYNC_SETUP_COPIES")));
  return wait;
}
static inline hipError_t cfdp_memset_sync(void *p, int v, size_t n) {
  const hipError_t e = cfdp_setup_copies_wait() ? hipMemset(p, v, n) : hipMemsetAsync(p, v, n, nullptr);
  return e != hipSuccess || !cfdp_setup_copies_wait() ? e : hipStreamSynchronize(nullptr);
}
static inline hipError_t cfdp_copy_d2d_sync(void *dst, const void *src, size_t n) {
  const hipError_t e = cfdp_setup_copies_wait() ? hipMemcpy(dst, src, n, hipMemcpyDeviceToDevice)
                                                : hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, nullptr);
  return e != hipSuccess || !cfdp_setup_copies_wait() ? e : hipStreamSynchronize(nullptr);
}

#define NEED_UPLOAD(g)                                                 \
  do {                                                                 \
    if (!(g) || !(g)->uploaded) return fail("no plan uploaded");       \
    HIP_TRY(hipSetDevice((g)->device));                                \
  } while (0)

struct cfdp_gpu {
  int device = 0;
  hipStream_t s_main = nullptr, s_comm = nullptr;
  hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_pack = nullptr, ev_senddone = nullptr,
             ev_fluxdone = nullptr, ev_fork = nullptr;
  bool uploaded = false;
  int nown = 0, nall = 0, ntiles = 0, nbtiles = 0;
  // launch groups: [0] the boundary tiles (possibly none), then the interior groups of the plan (cfdp_plan::group_begin) --
  // contiguous tile ranges, each with the maxima that select its kernel form.  A launch covers one SEGMENT = one group,
  // or the boundary tiles together with the first interior group (segs_of)
  struct tile_group {
    int begin = 0, n = 0, cls = 0;  // cls: the largest capacity class among its tiles (cfdp_tile_class)
    int tp = 0, max_halo = 0, max_blob = 0, max_rows = 0;  // owned points, halo rows, blob 16-byte units, staged rows (own + halo:
                                                           // NOT tp + max_halo -- the tile with the most halo rows usually is not one with the most points)
    size_t lds_grad = 0, lds_flux = 0;                     // the packed image of the register-staged kernels
  };
  std::vector<tile_group> groups;
  int rowlist_stride = 0;      // entries per tile of d_rowlist
  cfdp_tile_desc *d_tiles = nullptr;
  uint4 *d_blob = nullptr;
  int *d_halo = nullptr, *d_sendidx = nullptr;
  int *d_rowlist = nullptr;     // fixed-stride row lists of the fused pass (gg_args::rowlist), or null
  double *d_var = nullptr, *d_grad = nullptr, *d_flux = nullptr,
         *d_sendbuf = nullptr;
  bool own_grad = true, own_sendbuf = true;
  // fused iterations (flux(i) + gradients(i+1) in one pass): grad is double-buffered; d_grad
  // always is the buffer holding the latest gradients, d_grad_alt the one the next fused pass
  // writes.  flux_pending: flux mode of an iteration whose flux has been deferred, or -1.
  double *d_grad_alt = nullptr;
  bool own_grad_alt = true;
  int fusion = 0, flux_pending = -1;
  bool beside_rccl = false;    // the tiles being launched share the device with an RCCL kernel
  // true while ev_fluxdone marks the END of everything enqueued on the main stream (set when it is
  // recorded, cleared by every launch): the comm stream of the next step can then fork off that
  // record instead of paying for another marker in the main queue (~5 us of device time each)
  bool main_marked = false;
  // one process per GPU: this rank's RCCL communicator and the communicator rank of every partner
  ncclComm_t comm = nullptr;
  int comm_nranks = 0;  // ncclCommCount of `comm`
  bool rccl_self_exchange = false;  // cfdp_gpu_rccl_allow_self_exchange: a one-rank communicator may exchange with itself (measurements)
  std::vector<int> peer;
  // xGMI write + notify exchange (cfdp_gpu_ipc_*): this rank's IPC block [header | landing arena 0
  // | landing arena 1] -- partners write their rows and their arrival counters into it -- and the
  // partners' blocks opened here.  The ghost block the kernels read is then landing arena
  // (xiter & 1), xiter = exchanges started so far.
  struct ipc_state {
    bool on = false;
    unsigned char *block = nullptr;
    // memory of the block (CFDP_IPC_MODE; ipc_mode_from_env): 0 coarse-grained (system-scope loads / fences in the
    // kernels), 1 fine-grained, 2 split: the flag words in a small fine-grained block of their own (`flags`), the
    // landing arenas coarse-grained, and an explicit cache invalidate once a tile has seen its partners' flags
    int mode = 0;
    unsigned char *flags = nullptr;
    unsigned char my_handle[64] = {0};
    // notification: per partner (done[1 + s] / need[s] / tile_mask[t], gg_push_args) or one counter for all partners
    bool per_partner = false;
    bool counters = false;  // notification by counters (gg_push_args::counters); needs the per-partner protocol
    int cfg_mode = -1, cfg_wait_inkernel = -1, cfg_notify = -1, cfg_inkernel = -1;  // cfdp_gpu_ipc_configure (-1: environment / default)
    int *d_done = nullptr, *d_need = nullptr, *d_tile_iter = nullptr;
    unsigned long long *d_tile_mask = nullptr;
    size_t land_bytes = 0;
    long xiter = 0;
    std::vector<void *> opened;                  // partner blocks (hipIpcOpenMemHandle)
    std::vector<std::vector<unsigned char>> opened_handle;
    std::vector<double *> dst[2];                // [parity][slot] where my rows land at the partner
    std::vector<int *> rflag;                    // [slot] my arrival counter in the partner's header
    double **d_dst[2] = {nullptr, nullptr};
    int **d_rflag = nullptr;
    int *d_slot_of_row = nullptr, *d_send_off = nullptr;
    int *d_tile_off = nullptr, *d_ent = nullptr, *d_ent_row = nullptr;  // send rows per boundary tile
    // ... point-major: the FIRST destination of every point of every boundary tile, {slot or -1, row}, pt_stride entries
    // per tile (pushed from the lanes' registers); tile_xoff[t] = where, in tile t's entries, the further destinations
    // of points sent to several partners start (pushed by the parallel re-read form)
    int2 *d_pt_first = nullptr;
    int *d_tile_xoff = nullptr;
    int pt_stride = 64;
    bool inkernel = false;   // the fused pass pushes and notifies by itself
    // the copy-engine "put" rung (cfdp_gpu_ipc_configure push_inkernel = 2; the reference's MPI_Put variants,
    // src/exchange_data_mpidma.c:93-127): rows are packed into the send arena by gg_pack_kernel and every partner's slice
    // leaves as ONE hipMemcpyAsync into its landing slice, the notify kernel behind the copies on the same stream
    bool put = false;
    // the latest exchange has been started but nothing on the main stream waits for its arrival yet: the
    // boundary tiles of the next pushing pass wait themselves (gg_push_args::wait_polls); anything else that
    // touches ghost rows first enqueues the wait kernel (ipc_settle)
    bool wait_pending = false;
    bool wait_inkernel = true;
    // FAULT INJECTION (tests only, CFDP_IPC_FAULT=skip_wait): the boundary tiles of a pushing pass do NOT wait for
    // the previous exchange -- they read whatever the landing arena holds.  Exists so that a test can show that the
    // scaled-field validation sees a ghost row read one exchange early, and that a comparison of final states does not
    bool fault_skip_wait = false;
    // TESTS ONLY (CFDP_IPC_JITTER_US=M): a pseudo-random idle time of up to M microseconds in front of every step, drawn on
    // the device (new delays at every hipGraph replay): ranks drift against each other step by step
    int jitter_us = 0;
    unsigned *d_rng = nullptr;
    // hipGraphs of cfdp_gpu_run_steps_ipc, one set per configuration: a captured chunk has the schedule, the flux mode, the
    // arena parity AND the current grad buffer baked into its kernels' arguments.  Runs of different schedules alternate
    // (a benchmark times with exchange / without / bulk in turn) and an odd number of passes flips the grad buffers, so
    // several sets stay cached -- a set that had to be re-captured inside a timed region would be timed with its capture
    struct graph_set {
      hipGraphExec_t graph = nullptr, graph_rem = nullptr;  // main chunk of 50 steps; what is left after whole chunks
      hipGraph_t tmpl = nullptr, tmpl_rem = nullptr;        // what they were instantiated from (cfdp_refresh_exec)
      int graph_n = 0, graph_rem_n = 0;
      int exch = -1, overlap = -1, flux = -1, mode = -1, xpar = -1, scaled = -1, closed = -1;
      bool closed_flips = false;  // a closed batch with an odd number of passes leaves the two grad buffers swapped
      const double *cur = nullptr;
      unsigned long used = 0;
    } gs[8];
    unsigned long gs_clock = 0;
    long steps_replayed = 0, steps_streamed = 0, captures_failed = 0;  // cfdp_gpu_ipc_graph_stats
    void drop_graph_sets() {
      bool any = false;
      for (auto &x : gs) any = any || x.graph || x.graph_rem;
      if (any) (void)hipDeviceSynchronize();  // a graph that is still replaying must not be destroyed under it
      for (auto &x : gs) {
        if (x.graph) (void)hipGraphExecDestroy(x.graph);
        if (x.graph_rem) (void)hipGraphExecDestroy(x.graph_rem);
        if (x.tmpl) (void)hipGraphDestroy(x.tmpl);
        if (x.tmpl_rem) (void)hipGraphDestroy(x.tmpl_rem);
        x = graph_set();
      }
    }
  } ipc;
  double *land(int parity) const {
    return reinterpret_cast<double *>(ipc.block + GG_IPC_HDR_BYTES + (size_t)parity * ipc.land_bytes);
  }
  int *ipc_hdr() const { return reinterpret_cast<int *>(ipc.flags ? ipc.flags : ipc.block); }
  long iter = 0;               // phase-1 calls so far (in-process rank groups run in lockstep)
  std::vector<int> new2old, partner, send_off, recv_off, send_idx_host;
  std::vector<cfdp_tile_desc> h_tiles;
  bool interior_reads_ghosts = false;  // some tile without send points has a ghost in its halo
  std::vector<unsigned long long> tile_recv_mask;  // [nbtiles] partner slots (bit s) whose ghost rows a boundary tile reads
  // scaled-field validation of the exchange (cfdp_gpu_scaled_check_begin / _end; gg_validate_kernel): the reference
  // flux, the device-side state block, which flux rows no kernel ever writes (points without faces)
  struct scaled_state {
    bool on = false;
    int saved_flux_lanes = 0;
    int *d_state = nullptr;
    double *d_fref = nullptr;
    double *d_var0 = nullptr;   // var as it was at _begin: _end verifies var == var0 * 2^(iterations mod 3) before it restores it
    unsigned char *d_skip = nullptr;
  } sc;
  std::vector<int> faceless;           // owned points without faces, device numbering
  bool faceless_send = false;          // some send point has no faces: its stored row travels, no tile computes one
  std::vector<double> vol;     // [nown] dual volumes, device numbering (slot 7 of each var row)
  bool streaming = false;      // per-iteration bytes exceed the Infinity Cache: non-temporal blobs/rows
  // fused passes over ALL tiles alternate the direction in which every XCD walks its run of tiles, so
  // that a pass starts on what the previous one left in the Infinity Cache (only worth it when a pass
  // streams more than the cache holds; the values do not depend on the order)
  bool alternate = false;
  unsigned fused_passes = 0;
  int grad_lanes = 4, flux_lanes = 8;
  int last_flux_mode = CFDP_FLUX_CONSISTENT;  // of the latest flux launch (fused or not)
  bool pending_exchange = false;
  bool streams_exported = false;  // handed to the caller: not destroyed with the context
  // hipGraphs of cfdp_gpu_run_iterations: [0] the main chunk (50 fused passes / 25 iterations), [1] what
  // is left of a run after whole chunks -- so that ANY iteration count is replayed, not stream-launched
  hipGraphExec_t graph = nullptr, graph_rem = nullptr;
  hipGraph_t graph_tmpl = nullptr, graph_rem_tmpl = nullptr;  // what they were instantiated from (cfdp_refresh_exec)
  int graph_iters = 0, graph_rem_iters = 0, graph_flux = -1, graph_mode = -1, graph_gl = 0, graph_fl = 0, graph_fuse = -1;
  const double *graph_cur = nullptr;  // d_grad at the last capture: the graphs' pointers are baked in
  const double *graph_cur_slot[2] = {nullptr, nullptr};  // ... per slot ([0] graph, [1] graph_rem)
  const double *graph_whole_final = nullptr;  // whole-run graph: the buffer holding its last gradients
  void drop_graphs() {  // every cached graph has the grad buffers / kernel variants of its capture baked in
    ipc.drop_graph_sets();
    if (graph) { (void)hipGraphExecDestroy(graph); graph = nullptr; }
    if (graph_rem) { (void)hipGraphExecDestroy(graph_rem); graph_rem = nullptr; }
    if (graph_tmpl) { (void)hipGraphDestroy(graph_tmpl); graph_tmpl = nullptr; }
    if (graph_rem_tmpl) { (void)hipGraphDestroy(graph_rem_tmpl); graph_rem_tmpl = nullptr; }
    graph_iters = graph_rem_iters = 0;
  }

  // d_grad: nall*21 doubles laid out [A1: nown x 6][ghost rows: nghost x 21][A2: nown x 4][B: nown x 11] (gg_kernels.h)
  gg_grad_view grad_view() const {
    gg_grad_view v = gg_grad_view::of(d_grad, nown, nall);
    if (ipc.on) v.ghost = land((int)(ipc.xiter & 1));  // the latest exchange landed here
    return v;
  }
  gg_grad_view alt_view() const { return gg_grad_view::of(d_grad_alt, nown, nall); }
  bool will_fuse() const { return fusion && flux_pending >= 0 && d_grad_alt; }
  // pinned staging image of the field transfers (file numbering <-> device numbering happens on the
  // host, in parallel); grown on demand, freed with the context
  double *h_stage = nullptr;
  size_t h_stage_len = 0;
  double *stage(size_t n) {
    if (n > h_stage_len) {
      if (h_stage) (void)hipHostFree(h_stage);
      h_stage = nullptr;
      h_stage_len = 0;
      if (hipHostMalloc((void **)&h_stage, n * sizeof(double), hipHostMallocDefault) != hipSuccess) return nullptr;
      h_stage_len = n;
    }
    return h_stage;
  }
  // device image <-> rows in FILE numbering
  void rows_to_device(const double *rows, double *img) const {
    double *a1 = img, *gh = a1 + (size_t)nown * 6, *a2 = gh + (size_t)(nall - nown) * 21, *b = a2 + (size_t)nown * 4;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nall; i++) {
      const double *r = rows + (size_t)new2old[i] * 21;
      if (i < nown) {
        double e[10];
        gg_a_encode(r, e);  // the first ten doubles as the device keeps them (gg_kernels.h)
        memcpy(a1 + (size_t)i * 6, e, 6 * sizeof(double));
        memcpy(a2 + (size_t)i * 4, e + 6, 4 * sizeof(double));
        memcpy(b + (size_t)i * 11, r + 10, 11 * sizeof(double));
      } else {
        double *d = gh + (size_t)(i - nown) * 21;
        gg_a_encode(r, d);
        memcpy(d + 10, r + 10, 11 * sizeof(double));
      }
    }
  }
  void device_to_rows(const double *img, double *rows) const {
    const double *a1 = img, *gh = a1 + (size_t)nown * 6, *a2 = gh + (size_t)(nall - nown) * 21, *b = a2 + (size_t)nown * 4;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nall; i++) {
      double *r = rows + (size_t)new2old[i] * 21;
      if (i < nown) {
        double e[10];
        memcpy(e, a1 + (size_t)i * 6, 6 * sizeof(double));
        memcpy(e + 6, a2 + (size_t)i * 4, 4 * sizeof(double));
        for (int c = 0; c < 10; c++) r[c] = gg_a_decode(e, c);
        memcpy(r + 10, b + (size_t)i * 11, 11 * sizeof(double));
      } else {
        const double *d = gh + (size_t)(i - nown) * 21;
        for (int c = 0; c < 10; c++) r[c] = gg_a_decode(d, c);
        memcpy(r + 10, d + 10, 11 * sizeof(double));
      }
    }
  }
  gg_args args() const {
    gg_args a;
    a.tiles = d_tiles; a.blob = d_blob; a.halo_idx = d_halo; a.rowlist = d_rowlist; a.rowlist_stride = rowlist_stride; a.var = d_var;
    a.grad = grad_view(); a.flux = d_flux; a.nown = nown;
    return a;
  }
};

struct tile_range {
  int begin, n, tp, max_halo, max_blob, max_rows;
  size_t lds_grad, lds_flux;
  int cls;  // the largest capacity class among its tiles (cfdp_tile_class)
  // what the fixed-capacity kernels size their row regions with: they take (points, halo rows) and add them
  int row_halo() const { return max_rows > tp ? max_rows - tp : 0; }
};

namespace cfdp_detail __attribute__((visibility("hidden"))) {  // library-internal: not part of the C ABI
// gpu_abi.hip
int ipc_settle(cfdp_gpu *g);
int scaled_tail(cfdp_gpu *g, int lag, bool scale, hipStream_t st);
int scaled_lag(const cfdp_gpu *g, int with_flux);
int flush_flux(cfdp_gpu *g, bool record = true, hipStream_t st = nullptr);
int mark_main(cfdp_gpu *g);
int fork_comm(cfdp_gpu *g);
// the launches that cover a tile selector, in order; range_of = the first of them (the one that holds the boundary tiles
// when the selector includes them)
__attribute__((visibility("hidden"))) std::vector<tile_range> segs_of(const cfdp_gpu *g, int which);
__attribute__((visibility("hidden"))) tile_range range_of(const cfdp_gpu *g, int which);
int launch_grad(cfdp_gpu *g, int which, hipStream_t st, const gg_grad_view *into = nullptr);
__attribute__((visibility("hidden"))) int launch_flux_tiles(cfdp_gpu *g, int mode, int which, hipStream_t st, const gg_push_args *wait = nullptr);
int launch_flux(cfdp_gpu *g, int mode, hipStream_t st, const gg_push_args *wait = nullptr);
int launch_fused(cfdp_gpu *g, int which, hipStream_t st, const gg_push_args *push = nullptr);
void fused_done(cfdp_gpu *g);
// gpu_exchange.hip
void ipc_release(cfdp_gpu *g);
void ipc_push_args(cfdp_gpu *g, int parity, gg_push_args *out);
long ipc_max_polls();
void drop_ipc_graphs(cfdp_gpu *g);
}  // namespace cfdp_detail
using namespace cfdp_detail;

#endif
