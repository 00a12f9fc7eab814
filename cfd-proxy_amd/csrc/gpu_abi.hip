// gpu_abi.hip -- implementation of the C ABI declared in include/cfdproxy_hip.h.
// Device memory, streams, events, hipGraph capture and peer copies; the arithmetic lives in
// gg_kernels.hip.  No CPU fallback: every entry point needs a HIP device and says so.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is resolved at run time (cfdp_rccl_load)

#include "cfdproxy_hip.h"
#include "gg_kernels.h"

namespace {

thread_local char g_err[512] = "no error";

int fail(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return 1;
}

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail("%s failed: %s [%s:%d]", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

}  // namespace

// for the other translation units of the library (plan_kernels.hip): same buffer as cfdp_gpu_last_error()
int cfdp_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return 1;
}

struct cfdp_gpu {
  int device = 0;
  hipStream_t s_main = nullptr, s_comm = nullptr;
  hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_pack = nullptr, ev_senddone = nullptr,
             ev_fluxdone = nullptr, ev_fork = nullptr;
  bool uploaded = false;
  int nown = 0, nall = 0, ntiles = 0, nbtiles = 0;
  int tp[2] = {0, 0};         // max owned points per tile: [0] boundary, [1] interior
  size_t lds_grad[2] = {0, 0}, lds_flux[2] = {0, 0};
  cfdp_tile_desc *d_tiles = nullptr;
  uint4 *d_blob = nullptr;
  int *d_halo = nullptr, *d_sendidx = nullptr;
  int *d_rowlist = nullptr;     // fixed-stride row lists of the fused pass (gg_args::rowlist), or null
  // tile-resident iterations (gg_resident_kernel): the neighbour tiles of every tile (owners of its halo rows) and
  // one block [err: 4 ints][flags: ntiles ints] that is zeroed before every launch; resident: 0 off, 1 on when the
  // partition qualifies, 2 = the staleness test (G_k stores its rows times 2^(k-1))
  int *d_nbr_off = nullptr, *d_nbr = nullptr, *d_resident_state = nullptr;
  int max_nbr = 0, resident = 0;
  long resident_runs = 0;
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
    int *d_done = nullptr, *d_need = nullptr;
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
    // the latest exchange has been started but nothing on the main stream waits for its arrival yet: the
    // boundary tiles of the next pushing pass wait themselves (gg_push_args::wait_polls); anything else that
    // touches ghost rows first enqueues the wait kernel (ipc_settle)
    bool wait_pending = false;
    bool wait_inkernel = true;
    // FAULT INJECTION (tests only, CFDP_IPC_FAULT=skip_wait): the boundary tiles of a pushing pass do NOT wait for
    // the previous exchange -- they read whatever the landing arena holds.  Exists so that a test can show that the
    // scaled-field validation sees a ghost row read one exchange early, and that a comparison of final states does not
    bool fault_skip_wait = false;
    hipGraphExec_t graph = nullptr, graph_rem = nullptr;  // main chunk; what is left after whole chunks
    int graph_n = 0, graph_rem_n = 0;
    int g_exch = -1, g_overlap = -1, g_flux = -1, g_mode = -1, g_xpar = -1;
    const double *g_cur = nullptr;
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
    unsigned char *d_skip = nullptr;
  } sc;
  std::vector<int> faceless;           // owned points without faces, device numbering
  bool faceless_send = false;          // some send point has no faces: its stored row travels, no tile computes one
  std::vector<double> vol;     // [nown] dual volumes, device numbering (slot 7 of each var row)
  int max_halo[2] = {0, 0}, max_blob[2] = {0, 0};  // per tile class: halo rows, blob 16-byte units
  int max_rows[2] = {0, 0};    // per tile class: rows a tile stages (own + halo) -- NOT tp + max_halo: the tile with
                               // the most halo rows usually is not one with the most points
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
  int graph_iters = 0, graph_rem_iters = 0, graph_flux = -1, graph_mode = -1, graph_gl = 0, graph_fl = 0, graph_fuse = -1;
  const double *graph_cur = nullptr;  // d_grad at the last capture: the graphs' pointers are baked in
  const double *graph_cur_slot[2] = {nullptr, nullptr};  // ... per slot ([0] graph, [1] graph_rem)
  const double *graph_whole_final = nullptr;  // whole-run graph: the buffer holding its last gradients
  void drop_graphs() {
    if (graph) { (void)hipGraphExecDestroy(graph); graph = nullptr; }
    if (graph_rem) { (void)hipGraphExecDestroy(graph_rem); graph_rem = nullptr; }
    graph_iters = graph_rem_iters = 0;
  }

  // d_grad: nall*21 doubles laid out [A: nown x 10][ghost rows: nghost x 21][B: nown x 11]
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
    double *a = img, *gh = a + (size_t)nown * 10, *b = gh + (size_t)(nall - nown) * 21;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nall; i++) {
      const double *r = rows + (size_t)new2old[i] * 21;
      if (i < nown) {
        memcpy(a + (size_t)i * 10, r, 10 * sizeof(double));
        memcpy(b + (size_t)i * 11, r + 10, 11 * sizeof(double));
      } else {
        memcpy(gh + (size_t)(i - nown) * 21, r, 21 * sizeof(double));
      }
    }
  }
  void device_to_rows(const double *img, double *rows) const {
    const double *a = img, *gh = a + (size_t)nown * 10, *b = gh + (size_t)(nall - nown) * 21;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nall; i++) {
      double *r = rows + (size_t)new2old[i] * 21;
      if (i < nown) {
        memcpy(r, a + (size_t)i * 10, 10 * sizeof(double));
        memcpy(r + 10, b + (size_t)i * 11, 11 * sizeof(double));
      } else {
        memcpy(r, gh + (size_t)(i - nown) * 21, 21 * sizeof(double));
      }
    }
  }
  gg_args args() const {
    gg_args a;
    a.tiles = d_tiles; a.blob = d_blob; a.halo_idx = d_halo; a.rowlist = d_rowlist; a.var = d_var;
    a.grad = grad_view(); a.flux = d_flux; a.nown = nown;
    return a;
  }
};

namespace {
void ipc_release(cfdp_gpu *g);
}

extern "C" {

int cfdp_gpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *cfdp_gpu_last_error(void) { return g_err; }

int cfdp_gpu_create(int device, cfdp_gpu **out) {
  *out = nullptr;
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  if (n <= 0) return fail("no HIP device: the CFD-Proxy hot path has no CPU fallback");
  if (device < 0 || device >= n) return fail("device %d out of range [0,%d)", device, n);
  HIP_TRY(hipSetDevice(device));
  cfdp_gpu *g = new cfdp_gpu();
  g->device = device;
  if (const char *e = getenv("CFDP_DEBUG_ABLATE")) gg_debug_flags = atoi(e);
  if (const char *e = getenv("CFDP_RESIDENT")) {
    g->resident = atoi(e);
    if (g->resident < 0 || g->resident > 2) {
      delete g;
      return fail("CFDP_RESIDENT=%s: 0 (off), 1 (on where the partition qualifies) or 2 (staleness test)", e);
    }
  }
  if (const char *e = getenv("CFDP_FUSED_SPLIT")) gg_fused_split = atoi(e);
  HIP_TRY(hipStreamCreateWithFlags(&g->s_main, hipStreamNonBlocking));
  {
    // the comm stream carries the latency chain of an iteration (boundary tiles -> pack/push ->
    // exchange) while thousands of interior workgroups queue on the main stream: its kernels must
    // get the slots that free up first
    int lo = 0, hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    const char *pe = getenv("CFDP_COMM_PRIORITY");
    const int prio = pe ? atoi(pe) : hi;  // numerically lowest = highest priority
    HIP_TRY(hipStreamCreateWithPriority(&g->s_comm, hipStreamNonBlocking, prio));
  }
  HIP_TRY(hipEventCreate(&g->ev_a));
  HIP_TRY(hipEventCreate(&g->ev_b));
  HIP_TRY(hipEventCreateWithFlags(&g->ev_pack, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&g->ev_senddone, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&g->ev_fluxdone, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&g->ev_fork, hipEventDisableTiming));
  *out = g;
  return 0;
}

static void free_device(cfdp_gpu *g) {
  g->drop_graphs();
  (void)hipFree(g->d_tiles); (void)hipFree(g->d_blob); (void)hipFree(g->d_halo); (void)hipFree(g->d_sendidx); (void)hipFree(g->d_rowlist);
  (void)hipFree(g->d_var); (void)hipFree(g->d_flux);
  (void)hipFree(g->d_nbr_off); (void)hipFree(g->d_nbr); (void)hipFree(g->d_resident_state);
  g->d_nbr_off = g->d_nbr = g->d_resident_state = nullptr;
  (void)hipFree(g->sc.d_state); (void)hipFree(g->sc.d_fref); (void)hipFree(g->sc.d_skip);
  g->sc = cfdp_gpu::scaled_state();
  if (g->own_grad) (void)hipFree(g->d_grad);
  if (g->own_grad_alt) (void)hipFree(g->d_grad_alt);
  if (g->own_sendbuf) (void)hipFree(g->d_sendbuf);
  g->d_grad_alt = nullptr; g->own_grad_alt = true; g->flux_pending = -1; g->iter = 0;
  g->d_tiles = nullptr; g->d_blob = nullptr; g->d_halo = g->d_sendidx = nullptr; g->d_rowlist = nullptr;
  g->d_var = g->d_grad = g->d_flux = g->d_sendbuf = nullptr;
  g->own_grad = g->own_sendbuf = true;
  g->uploaded = false;
}

void cfdp_gpu_destroy(cfdp_gpu *g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  (void)hipDeviceSynchronize();
  (void)cfdp_gpu_rccl_finalize(g);
  ipc_release(g);
  free_device(g);
  if (g->h_stage) (void)hipHostFree(g->h_stage);
  if (!g->streams_exported) {  // exported streams may still be referenced by the caller's runtime
    if (g->s_main) (void)hipStreamDestroy(g->s_main);
    if (g->s_comm) (void)hipStreamDestroy(g->s_comm);
  }
  for (hipEvent_t e : {g->ev_a, g->ev_b, g->ev_pack, g->ev_senddone, g->ev_fluxdone, g->ev_fork})
    if (e) (void)hipEventDestroy(e);
  delete g;
}

int cfdp_gpu_upload_plan(cfdp_gpu *g, const cfdp_plan *p) {
  if (!g || !p) return fail("null argument");
  HIP_TRY(hipSetDevice(g->device));
  if (g->uploaded) free_device(g);
  g->nown = p->nown; g->nall = p->nall; g->ntiles = p->ntiles; g->nbtiles = p->nbtiles;
  g->tp[0] = g->tp[1] = 0;
  g->max_halo[0] = g->max_halo[1] = 0;
  g->max_blob[0] = g->max_blob[1] = 0;
  g->max_rows[0] = g->max_rows[1] = 0;
  for (int t = 0; t < p->ntiles; t++) {
    int c = t < p->nbtiles ? 0 : 1;
    if (p->tiles[t].npts + p->tiles[t].nhalo > g->max_rows[c]) g->max_rows[c] = p->tiles[t].npts + p->tiles[t].nhalo;
    if (p->tiles[t].npts > g->tp[c]) g->tp[c] = p->tiles[t].npts;
    if (p->tiles[t].nhalo > g->max_halo[c]) g->max_halo[c] = p->tiles[t].nhalo;
    if (p->tiles[t].blob_qw > g->max_blob[c]) g->max_blob[c] = p->tiles[t].blob_qw;
  }
  for (int c = 0; c < 2; c++) {
    g->lds_grad[c] = (size_t)p->lds_grad_cls[c];
    g->lds_flux[c] = (size_t)p->lds_flux_cls[c];
  }
  if (p->lds_grad > 160 * 1024 || p->lds_flux > 160 * 1024)
    return fail("tile needs %ld / %ld bytes of LDS (> 160 KiB): use a smaller tile_points",
                p->lds_grad, p->lds_flux);
  g->new2old.assign(p->new2old, p->new2old + p->nall);
  g->h_tiles.assign(p->tiles, p->tiles + p->ntiles);
  g->interior_reads_ghosts = false;
  for (int t = p->nbtiles; t < p->ntiles && !g->interior_reads_ghosts; t++)
    for (int h = 0; h < p->tiles[t].nhalo; h++)
      if (p->halo_idx[p->tiles[t].halo_off + h] >= p->nown) { g->interior_reads_ghosts = true; break; }
  g->partner.assign(p->partner, p->partner + p->npartners);
  g->send_off.assign(1, 0);
  g->recv_off.assign(1, 0);
  if (p->npartners) {
    g->send_off.assign(p->send_off, p->send_off + p->npartners + 1);
    g->recv_off.assign(p->recv_off, p->recv_off + p->npartners + 1);
  }
  g->tile_recv_mask.assign((size_t)p->nbtiles, 0ull);
  if (p->npartners && p->npartners <= 64)
    for (int t = 0; t < p->nbtiles; t++)
      for (int h = 0; h < p->tiles[t].nhalo; h++) {
        const int gi = p->halo_idx[p->tiles[t].halo_off + h] - p->nown;  // ghost rows are grouped by partner, message order
        if (gi < 0) continue;
        int s = 0;
        while (s + 1 < p->npartners && gi >= p->recv_off[s + 1]) s++;
        if (gi < p->recv_off[p->npartners]) g->tile_recv_mask[(size_t)t] |= 1ull << s;
        else g->tile_recv_mask[(size_t)t] = ~0ull;  // a ghost nobody sends: never matches a send mask
      }
  const size_t nsend = (size_t)g->send_off.back();
  HIP_TRY(hipMalloc(&g->d_tiles, sizeof(cfdp_tile_desc) * (size_t)p->ntiles));
  HIP_TRY(hipMalloc(&g->d_blob, (size_t)p->blob_bytes + 16));
  HIP_TRY(hipMalloc(&g->d_halo, sizeof(int) * (size_t)(p->nhalo_total + 1)));
  HIP_TRY(hipMalloc(&g->d_var, sizeof(double) * 8 * (size_t)p->nall));
  HIP_TRY(hipMalloc(&g->d_grad, sizeof(double) * 21 * (size_t)p->nall));
  HIP_TRY(hipMalloc(&g->d_flux, sizeof(double) * 3 * (size_t)p->nown));
  HIP_TRY(hipMalloc(&g->d_sendidx, sizeof(int) * (nsend + 1)));
  HIP_TRY(hipMalloc(&g->d_sendbuf, sizeof(double) * 21 * (nsend + 1)));
  g->own_grad = g->own_sendbuf = true;
  HIP_TRY(hipMemcpy(g->d_tiles, p->tiles, sizeof(cfdp_tile_desc) * (size_t)p->ntiles, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(g->d_blob, p->blob, (size_t)p->blob_bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(g->d_halo, 0, sizeof(int) * (size_t)(p->nhalo_total + 1)));
  if (p->nhalo_total)
    HIP_TRY(hipMemcpy(g->d_halo, p->halo_idx, sizeof(int) * (size_t)p->nhalo_total, hipMemcpyHostToDevice));
  {  // row lists at a fixed stride, when every tile fits one (the 256-thread fused pass: <= 204 rows)
    bool fits = p->ntiles > 0 && p->tile_points <= 64;
    for (int t = 0; t < p->ntiles && fits; t++) fits = p->tiles[t].npts + p->tiles[t].nhalo <= 204 && p->tiles[t].npts > 0;
    if (const char *e = getenv("CFDP_ROWLIST")) fits = fits && atoi(e) != 0;
    if (fits) {
      std::vector<int> rl((size_t)p->ntiles * GG_ROW_STRIDE);
      for (int t = 0; t < p->ntiles; t++) {
        const cfdp_tile_desc &td = p->tiles[t];
        int *r = rl.data() + (size_t)t * GG_ROW_STRIDE;
        const int n = td.npts + td.nhalo;
        for (int i = 0; i < td.npts; i++) r[i] = td.pstart + i;
        for (int i = 0; i < td.nhalo; i++) r[td.npts + i] = p->halo_idx[td.halo_off + i];
        for (int i = n; i < GG_ROW_STRIDE; i++) r[i] = r[n - 1];
      }
      HIP_TRY(hipMalloc(&g->d_rowlist, rl.size() * sizeof(int)));
      HIP_TRY(hipMemcpy(g->d_rowlist, rl.data(), rl.size() * sizeof(int), hipMemcpyHostToDevice));
    }
  }
  {  // neighbour tiles = the owners of a tile's owned halo rows (symmetric: a cut face is stored with both tiles)
    std::vector<int> tile_of((size_t)p->nown, -1), off((size_t)p->ntiles + 1, 0), nbr;
    for (int t = 0; t < p->ntiles; t++)
      for (int i = 0; i < p->tiles[t].npts; i++) tile_of[(size_t)p->tiles[t].pstart + i] = t;
    std::vector<int> seen((size_t)p->ntiles, -1);
    g->max_nbr = 0;
    for (int t = 0; t < p->ntiles; t++) {
      for (int h = 0; h < p->tiles[t].nhalo; h++) {
        const int row = p->halo_idx[p->tiles[t].halo_off + h];
        if (row >= p->nown) continue;
        const int u = tile_of[row];
        if (u >= 0 && u != t && seen[u] != t) { seen[u] = t; nbr.push_back(u); }
      }
      off[(size_t)t + 1] = (int)nbr.size();
      if (off[(size_t)t + 1] - off[t] > g->max_nbr) g->max_nbr = off[(size_t)t + 1] - off[t];
    }
    HIP_TRY(hipMalloc(&g->d_nbr_off, off.size() * sizeof(int)));
    HIP_TRY(hipMalloc(&g->d_nbr, (nbr.size() + 1) * sizeof(int)));
    HIP_TRY(hipMalloc(&g->d_resident_state, (((size_t)p->ntiles + 4 + 3) & ~(size_t)3) * sizeof(int)));
    HIP_TRY(hipMemcpy(g->d_nbr_off, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice));
    if (!nbr.empty()) HIP_TRY(hipMemcpy(g->d_nbr, nbr.data(), nbr.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  g->vol.assign(p->vol, p->vol + p->nown);
  {  // blob + var rows + grad rows streamed per iteration vs the 256 MiB Infinity Cache
    const double per_iter = (double)p->blob_bytes + (double)p->nall * (64.0 + 168.0);
    g->streaming = per_iter > 192.0 * 1024 * 1024;
    if (const char *e = getenv("CFDP_STREAMING")) g->streaming = atoi(e) != 0;
    g->alternate = g->streaming;
    if (const char *e = getenv("CFDP_ALTERNATE")) g->alternate = atoi(e) != 0;
  }
  g->send_idx_host.assign(p->send_idx, p->send_idx + nsend);
  g->faceless_send = false;
  for (size_t j = 0; j < nsend && p->degree; j++)
    if (p->degree[p->send_idx[j]] == 0) g->faceless_send = true;
  g->faceless.clear();
  for (int i = 0; i < p->nown && p->degree; i++)
    if (p->degree[i] == 0) g->faceless.push_back(i);
  if (nsend)
    HIP_TRY(hipMemcpy(g->d_sendidx, p->send_idx, sizeof(int) * nsend, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(g->d_var, 0, sizeof(double) * 8 * (size_t)p->nall));
  HIP_TRY(hipMemset(g->d_grad, 0, sizeof(double) * 21 * (size_t)p->nall));
  HIP_TRY(hipMemset(g->d_flux, 0, sizeof(double) * 3 * (size_t)p->nown));
  g->uploaded = true;
  if (g->fusion) return cfdp_gpu_set_fusion(g, 1);
  return 0;
}

#define NEED_UPLOAD(g)                                                 \
  do {                                                                 \
    if (!(g) || !(g)->uploaded) return fail("no plan uploaded");       \
    HIP_TRY(hipSetDevice((g)->device));                                \
  } while (0)

int cfdp_gpu_bind_grad(cfdp_gpu *g, void *dev_grad) {
  NEED_UPLOAD(g);
  if (!dev_grad) return fail("null device pointer");
  if ((uintptr_t)dev_grad & 15) return fail("grad buffer must be 16-byte aligned");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(dev_grad, g->d_grad, sizeof(double) * 21 * (size_t)g->nall, hipMemcpyDeviceToDevice));
  if (g->own_grad) (void)hipFree(g->d_grad);
  g->d_grad = static_cast<double *>(dev_grad);
  g->own_grad = false;
  g->drop_graphs();
  return 0;
}

int cfdp_gpu_bind_sendbuf(cfdp_gpu *g, void *dev_sendbuf) {
  NEED_UPLOAD(g);
  if (!dev_sendbuf) return fail("null device pointer");
  HIP_TRY(hipDeviceSynchronize());
  if (g->own_sendbuf) (void)hipFree(g->d_sendbuf);
  g->d_sendbuf = static_cast<double *>(dev_sendbuf);
  g->own_sendbuf = false;
  return 0;
}

static int launch_flux(cfdp_gpu *g, int mode, hipStream_t st);
static long ipc_max_polls_c();

// xGMI write + notify: make the main stream wait (bounded, on the device) for the latest exchange's arrival,
// if nothing has done so yet -- before anything but a pushing fused pass touches ghost rows
static int ipc_settle(cfdp_gpu *g) {
  if (!g->ipc.on || !g->ipc.wait_pending) return 0;
  g->ipc.wait_pending = false;
  g->main_marked = false;
  HIP_TRY(gg_launch_wait(g->ipc_hdr(), (int)g->partner.size(), ipc_max_polls_c(), g->s_main));
  return 0;
}

// scaled-field validation (cfdp_gpu_scaled_check_begin): at the end of a step, on the main stream once it has joined
// the comm stream -- every reader of var of this iteration is ahead of it, the next gradient launch behind it.
// lag: see gg_validate_kernel.  scale = false: compare only (the deferred flux of a run's last iteration).
static int scaled_tail(cfdp_gpu *g, int lag, bool scale, hipStream_t st) {
  if (!g->sc.on) return 0;
  g->main_marked = false;
  HIP_TRY(gg_launch_validate(g->d_var, g->nall, g->d_flux, g->sc.d_fref, g->sc.d_skip, g->nown, lag, scale, g->sc.d_state, st));
  return 0;
}
// the lag of the flux a step leaves in d_flux when it closes with (with_flux, fused deferral or not)
static int scaled_lag(const cfdp_gpu *g, int with_flux) { return !with_flux ? -1 : (g->fusion && g->d_grad_alt ? 1 : 0); }

// run the deferred flux of the last fused-mode iteration, if any
static int flush_flux(cfdp_gpu *g, bool record = true, hipStream_t st = nullptr) {
  if (ipc_settle(g)) return 1;
  if (g->flux_pending < 0) return 0;
  const int mode = g->flux_pending;
  g->flux_pending = -1;
  if (launch_flux(g, mode, st ? st : g->s_main)) return 1;
  if (scaled_tail(g, 1, false, st ? st : g->s_main)) return 1;
  if (record) {
    HIP_TRY(hipEventRecord(g->ev_fluxdone, st ? st : g->s_main));
    g->main_marked = !st || st == g->s_main;
  }
  return 0;
}

int cfdp_gpu_set_fusion(cfdp_gpu *g, int on) {
  if (!g) return fail("null context");
  if (!g->uploaded) { g->fusion = on != 0; return 0; }  // takes effect at upload
  HIP_TRY(hipSetDevice(g->device));
  if (!on) {
    if (flush_flux(g)) return 1;
    g->fusion = 0;
    return 0;
  }
  if (!g->d_grad_alt) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMalloc(&g->d_grad_alt, sizeof(double) * 21 * (size_t)g->nall));
    HIP_TRY(hipMemcpy(g->d_grad_alt, g->d_grad, sizeof(double) * 21 * (size_t)g->nall, hipMemcpyDeviceToDevice));
    g->own_grad_alt = true;
  }
  g->fusion = 1;
  g->drop_graphs();
  return 0;
}

int cfdp_gpu_bind_grad_alt(cfdp_gpu *g, void *dev_grad) {
  NEED_UPLOAD(g);
  if (!dev_grad) return fail("null device pointer");
  if ((uintptr_t)dev_grad & 15) return fail("grad buffer must be 16-byte aligned");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(dev_grad, g->d_grad_alt ? g->d_grad_alt : g->d_grad, sizeof(double) * 21 * (size_t)g->nall,
                    hipMemcpyDeviceToDevice));
  if (g->d_grad_alt && g->own_grad_alt) (void)hipFree(g->d_grad_alt);
  g->d_grad_alt = static_cast<double *>(dev_grad);
  g->own_grad_alt = false;
  g->drop_graphs();
  return 0;
}

int cfdp_gpu_set_var(cfdp_gpu *g, const double *var) {
  NEED_UPLOAD(g);
  const size_t len = (size_t)g->nall * 8;
  double *tmp = g->stage(len);
  if (!tmp) return fail("no pinned host memory for the staging image");
  const int nall = g->nall, nown = g->nown;
  const int *new2old = g->new2old.data();
  const double *vol = g->vol.data();
#pragma omp parallel for schedule(static)
  for (int i = 0; i < nall; i++) {
    memcpy(tmp + (size_t)i * 8, var + (size_t)new2old[i] * 7, 7 * sizeof(double));
    tmp[(size_t)i * 8 + 7] = i < nown ? vol[i] : 0.0;  // pvolume rides in the row's pad
  }
  HIP_TRY(hipDeviceSynchronize());  // the context's streams are non-blocking: nothing may still read var
  HIP_TRY(hipMemcpy(g->d_var, tmp, len * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

int cfdp_gpu_set_grad(cfdp_gpu *g, const double *grad) {
  NEED_UPLOAD(g);
  if (flush_flux(g)) return 1;
  const size_t len = (size_t)g->nall * 21;
  double *tmp = g->stage(len);
  if (!tmp) return fail("no pinned host memory for the staging image");
  g->rows_to_device(grad, tmp);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(g->d_grad, tmp, len * sizeof(double), hipMemcpyHostToDevice));
  if (g->d_grad_alt)  // rows no kernel writes (ghosts without an exchange, faceless points) read the same from either buffer
    HIP_TRY(hipMemcpy(g->d_grad_alt, tmp, len * sizeof(double), hipMemcpyHostToDevice));
  if (g->ipc.on && g->nall > g->nown)
    for (int par = 0; par < 2; par++)
      HIP_TRY(hipMemcpy(g->land(par), tmp + (size_t)g->nown * 10,
                        sizeof(double) * 21 * (size_t)(g->nall - g->nown), hipMemcpyHostToDevice));
  return 0;
}

int cfdp_gpu_set_flux(cfdp_gpu *g, const double *flux) {
  NEED_UPLOAD(g);
  if (flush_flux(g)) return 1;
  HIP_TRY(hipDeviceSynchronize());
  const size_t len = (size_t)g->nown * 3;
  double *tmp = g->stage(len);
  if (!tmp) return fail("no pinned host memory for the staging image");
  for (int i = 0; i < g->nown; i++)
    memcpy(tmp + (size_t)i * 3, flux + (size_t)g->new2old[i] * 3, 3 * sizeof(double));
  HIP_TRY(hipMemcpy(g->d_flux, tmp, len * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

int cfdp_gpu_get_grad(cfdp_gpu *g, double *grad) {
  NEED_UPLOAD(g);
  if (flush_flux(g)) return 1;
  HIP_TRY(hipDeviceSynchronize());
  const size_t len = (size_t)g->nall * 21;
  double *tmp = g->stage(len);
  if (!tmp) return fail("no pinned host memory for the staging image");
  HIP_TRY(hipMemcpy(tmp, g->d_grad, len * sizeof(double), hipMemcpyDeviceToHost));
  if (g->ipc.on && g->nall > g->nown)  // the ghost rows live in the landing arena of the latest exchange
    HIP_TRY(hipMemcpy(tmp + (size_t)g->nown * 10, g->grad_view().ghost,
                      sizeof(double) * 21 * (size_t)(g->nall - g->nown), hipMemcpyDeviceToHost));
  g->device_to_rows(tmp, grad);
  return 0;
}

int cfdp_gpu_get_flux(cfdp_gpu *g, double *flux) {
  NEED_UPLOAD(g);
  if (flush_flux(g)) return 1;
  HIP_TRY(hipDeviceSynchronize());
  const size_t len = (size_t)g->nown * 3;
  double *tmp = g->stage(len);
  if (!tmp) return fail("no pinned host memory for the staging image");
  HIP_TRY(hipMemcpy(tmp, g->d_flux, len * sizeof(double), hipMemcpyDeviceToHost));
  for (int i = 0; i < g->nown; i++) /* ghost rows of psd_flux are left untouched */
    memcpy(flux + (size_t)g->new2old[i] * 3, tmp + (size_t)i * 3, 3 * sizeof(double));
  return 0;
}

int cfdp_gpu_set_variant(cfdp_gpu *g, int grad_lanes, int flux_lanes) {
  if (!g) return fail("null context");
  auto ok = [](int l) { return l == 1 || l == 2 || l == 4 || l == 8; };
  if (grad_lanes == 0) grad_lanes = 4;
  if (flux_lanes == 0) flux_lanes = 8;
  if (!ok(grad_lanes) || !ok(flux_lanes)) return fail("lanes per point must be 1, 2, 4 or 8");
  g->grad_lanes = grad_lanes;
  g->flux_lanes = flux_lanes;
  return 0;
}

// A launch covers one tile range.  The two tile classes (boundary tiles are half-size sheets)
// only get launches of their own when the schedule needs the boundary tiles early; otherwise
// ALL tiles go in ONE launch sized for the larger class -- a separate launch for the few hundred
// boundary tiles of a rank costs ~15 us of mostly idle device per iteration (measured: 58 vs 43 us
// for rank 0 of the 2- and 8-rank decompositions).
// "the iteration is complete": record ev_fluxdone at the end of the main stream
static int mark_main(cfdp_gpu *g) {
  HIP_TRY(hipEventRecord(g->ev_fluxdone, g->s_main));
  g->main_marked = true;
  return 0;
}
// the comm stream starts after everything enqueued on the main stream so far
static int fork_comm(cfdp_gpu *g) {
  if (g->main_marked) {
    HIP_TRY(hipStreamWaitEvent(g->s_comm, g->ev_fluxdone, 0));
  } else {
    HIP_TRY(hipEventRecord(g->ev_fork, g->s_main));
    HIP_TRY(hipStreamWaitEvent(g->s_comm, g->ev_fork, 0));
  }
  return 0;
}

struct tile_range {
  int begin, n, tp, max_halo, max_blob, max_rows;
  size_t lds_grad, lds_flux;
  // what the fixed-capacity kernels size their row regions with: they take (points, halo rows) and add them
  int row_halo() const { return max_rows > tp ? max_rows - tp : 0; }
};
static tile_range range_of(const cfdp_gpu *g, int which) {
  auto cls = [&](int c) {
    return tile_range{c ? g->nbtiles : 0, c ? g->ntiles - g->nbtiles : g->nbtiles, g->tp[c], g->max_halo[c],
                      g->max_blob[c], g->max_rows[c], g->lds_grad[c], g->lds_flux[c]};
  };
  if (which == CFDP_TILES_BOUNDARY) return cls(0);
  if (which == CFDP_TILES_INTERIOR) return cls(1);
  const tile_range b = cls(0), i = cls(1);
  if (b.n == 0) return i;
  if (i.n == 0) return b;
  return tile_range{0, g->ntiles, b.tp > i.tp ? b.tp : i.tp, b.max_halo > i.max_halo ? b.max_halo : i.max_halo,
                    b.max_blob > i.max_blob ? b.max_blob : i.max_blob, b.max_rows > i.max_rows ? b.max_rows : i.max_rows,
                    b.lds_grad > i.lds_grad ? b.lds_grad : i.lds_grad, b.lds_flux > i.lds_flux ? b.lds_flux : i.lds_flux};
}

static int launch_grad(cfdp_gpu *g, int which, hipStream_t st, const gg_grad_view *into = nullptr) {
  g->main_marked = false;
  gg_args a = g->args();
  if (into) a.grad = *into;
  const tile_range r = range_of(g, which);
  HIP_TRY(gg_launch_gradient(a, g->grad_lanes, r.begin, r.n, r.tp, r.lds_grad, r.row_halo(), r.max_blob, g->streaming, st));
  return 0;
}

static int launch_flux_tiles(cfdp_gpu *g, int mode, int which, hipStream_t st) {
  g->main_marked = false;
  g->last_flux_mode = mode;
  const gg_args a = g->args();
  const tile_range r = range_of(g, which);
  HIP_TRY(gg_launch_flux(a, g->flux_lanes, mode == CFDP_FLUX_REFERENCE, r.begin, r.n, r.tp, r.lds_flux, r.row_halo(),
                         r.max_blob, g->streaming, st));
  return 0;
}

static int launch_flux(cfdp_gpu *g, int mode, hipStream_t st) { return launch_flux_tiles(g, mode, CFDP_TILES_ALL, st); }

// the deferred flux (from d_grad) + the next gradients (into d_grad_alt) over the selected tiles
// in one pass; falls back to the two separate kernels when no fused capacity fits the tiles.
// The caller swaps the buffers (fused_done) once every tile range of the iteration is enqueued.
// push != nullptr: the tiles push and notify themselves; returns 2 (nothing launched) if no fused
// kernel fits these tiles, so that the caller can take the separate-kernel path
static int launch_fused(cfdp_gpu *g, int which, hipStream_t st, const gg_push_args *push = nullptr) {
  g->main_marked = false;
  const gg_args a = g->args();
  const gg_grad_view gnew = g->alt_view();
  const int mode = g->flux_pending;
  g->last_flux_mode = mode;
  const tile_range r = range_of(g, which);
  if (push && !gg_fused_fits(r.tp, r.row_halo(), r.max_blob)) return 2;
  const bool reverse = g->alternate && which == CFDP_TILES_ALL && !push && (g->fused_passes++ & 1u);
  const hipError_t e = gg_launch_fused(a, gnew, mode == CFDP_FLUX_REFERENCE, r.begin, r.n, r.tp, r.row_halo(), r.max_blob,
                                       g->streaming, !g->beside_rccl, st, push, reverse);
  if (e == hipErrorNotSupported && push) return 2;
  if (e == hipErrorNotSupported) {
    if (launch_flux_tiles(g, mode, which, st)) return 1;
    if (launch_grad(g, which, st, &gnew)) return 1;
  } else {
    HIP_TRY(e);
  }
  return 0;
}

static void fused_done(cfdp_gpu *g) {
  std::swap(g->d_grad, g->d_grad_alt);
  std::swap(g->own_grad, g->own_grad_alt);
  g->flux_pending = -1;
}

int cfdp_gpu_gradients(cfdp_gpu *g, int which_tiles, void *stream) {
  NEED_UPLOAD(g);
  if (which_tiles < 0 || which_tiles > 2) return fail("bad tile selector %d", which_tiles);
  if (flush_flux(g)) return 1;
  return launch_grad(g, which_tiles, stream ? (hipStream_t)stream : g->s_main);
}

int cfdp_gpu_flux(cfdp_gpu *g, int mode, void *stream) {
  NEED_UPLOAD(g);
  if (mode != CFDP_FLUX_CONSISTENT && mode != CFDP_FLUX_REFERENCE) return fail("bad flux mode %d", mode);
  if (flush_flux(g)) return 1;
  return launch_flux(g, mode, stream ? (hipStream_t)stream : g->s_main);
}

int cfdp_gpu_pack(cfdp_gpu *g, void *stream) {
  NEED_UPLOAD(g);
  HIP_TRY(gg_launch_pack(g->d_sendidx, g->send_off.back(), g->grad_view(), g->d_sendbuf,
                         stream ? (hipStream_t)stream : g->s_main));
  return 0;
}

int cfdp_gpu_unpack(cfdp_gpu *g, const void *dev_recvbuf, void *stream) {
  NEED_UPLOAD(g);
  HIP_TRY(gg_launch_unpack(static_cast<const double *>(dev_recvbuf), g->recv_off.back(), g->grad_view(),
                           stream ? (hipStream_t)stream : g->s_main));
  return 0;
}

int cfdp_gpu_sync(cfdp_gpu *g) {
  if (!g) return fail("null context");
  HIP_TRY(hipSetDevice(g->device));
  if (g->uploaded && flush_flux(g)) return 1;
  HIP_TRY(hipDeviceSynchronize());
  return 0;
}

void *cfdp_gpu_stream(cfdp_gpu *g, int which) {
  g->streams_exported = true;  // a foreign runtime may keep references (events, allocator bookkeeping)
  return which ? g->s_comm : g->s_main;
}
int cfdp_gpu_npartners(const cfdp_gpu *g) { return (int)g->partner.size(); }
int cfdp_gpu_partner_rank(const cfdp_gpu *g, int s) {
  return (s >= 0 && s < (int)g->partner.size()) ? g->partner[s] : -1;
}
void *cfdp_gpu_send_ptr(cfdp_gpu *g, int s, size_t *bytes) {
  if (s < 0 || s >= (int)g->partner.size()) return nullptr;
  if (bytes) *bytes = (size_t)(g->send_off[s + 1] - g->send_off[s]) * 21 * sizeof(double);
  return g->d_sendbuf + (size_t)g->send_off[s] * 21;
}
void *cfdp_gpu_recv_ptr(cfdp_gpu *g, int s, size_t *bytes) {
  if (s < 0 || s >= (int)g->partner.size()) return nullptr;
  if (bytes) *bytes = (size_t)(g->recv_off[s + 1] - g->recv_off[s]) * 21 * sizeof(double);
  return g->grad_view().ghost + (size_t)g->recv_off[s] * 21;  // whole rows, message order
}
void *cfdp_gpu_grad_ptr(cfdp_gpu *g) { return g->d_grad; }
void *cfdp_gpu_var_ptr(cfdp_gpu *g) { return g->d_var; }

int cfdp_gpu_counts(const cfdp_gpu *g, int *nown, int *nall, int *nsend, int *nrecv) {
  if (!g || !g->uploaded) return fail("no plan uploaded");
  if (nown) *nown = g->nown;
  if (nall) *nall = g->nall;
  if (nsend) *nsend = g->send_off.back();
  if (nrecv) *nrecv = g->recv_off.back();
  return 0;
}

// --------------------------------------------------- one rank per process: step brackets
// The caller owns the transport (e.g. RCCL send/recv enqueued on this context's comm stream
// between the two calls); these two calls enqueue everything else of one iteration, so a
// host pays two ABI calls + one communication call per step.
//   pre : comm forks off main (ev_fork: after the previous iteration); [boundary tiles -> pack] on comm,
//         interior tiles on main, concurrently (bulk: all tiles -> pack on main, comm waits ev_pack)
//   post: main waits for everything enqueued on comm so far; flux; ev_fluxdone
int cfdp_gpu_step_pre(cfdp_gpu *g, int with_exchange, int overlap) {
  NEED_UPLOAD(g);
  const bool comm = with_exchange && !g->partner.empty();
  g->pending_exchange = comm;
  g->iter++;
  // fused mode: this iteration's gradients ride with the previous iteration's deferred flux and
  // go to the other grad buffer, which then becomes the current one (its ghost block is where
  // the exchange between pre() and post() delivers)
  const bool fused = g->will_fuse();
  if (!fused && flush_flux(g)) return 1;
  auto grad_tiles = [&](int which, hipStream_t st) { return fused ? launch_fused(g, which, st) : launch_grad(g, which, st); };
  if (!comm) {
    if (grad_tiles(CFDP_TILES_ALL, g->s_main)) return 1;
    if (fused) fused_done(g);
    return 0;
  }
  const gg_grad_view src = fused ? g->alt_view() : g->grad_view();  // where this iteration's gradients go
  // the comm stream starts after everything enqueued on the main stream so far (the previous
  // iteration, or whatever else the caller launched there)
  if (fork_comm(g)) return 1;
  if (overlap) {
    // boundary tiles + pack (+ the caller's exchange) on the comm stream, interior tiles on the main
    // stream AT THE SAME TIME: the few hundred boundary tiles alone would leave most of the device
    // idle for the ~7 us a tile takes
    g->beside_rccl = g->comm != nullptr;
    int rc = grad_tiles(CFDP_TILES_BOUNDARY, g->s_comm);
    if (!rc && gg_launch_pack(g->d_sendidx, g->send_off.back(), src, g->d_sendbuf, g->s_comm) != hipSuccess)
      rc = fail("pack launch failed");
    if (!rc) rc = grad_tiles(CFDP_TILES_INTERIOR, g->s_main);
    g->beside_rccl = false;
    if (rc) return 1;
  } else {
    if (grad_tiles(CFDP_TILES_ALL, g->s_main)) return 1;
    HIP_TRY(gg_launch_pack(g->d_sendidx, g->send_off.back(), src, g->d_sendbuf, g->s_main));
    HIP_TRY(hipEventRecord(g->ev_pack, g->s_main));
    HIP_TRY(hipStreamWaitEvent(g->s_comm, g->ev_pack, 0));
  }
  if (fused) fused_done(g);
  return 0;
}

int cfdp_gpu_step_post(cfdp_gpu *g, int with_flux, int flux_mode) {
  NEED_UPLOAD(g);
  if (g->pending_exchange) {
    HIP_TRY(hipEventRecord(g->ev_senddone, g->s_comm));
    HIP_TRY(hipStreamWaitEvent(g->s_main, g->ev_senddone, 0));
  }
  g->pending_exchange = false;
  if (with_flux) {
    if (flux_mode != CFDP_FLUX_CONSISTENT && flux_mode != CFDP_FLUX_REFERENCE) return fail("bad flux mode %d", flux_mode);
    if (g->fusion && g->d_grad_alt) g->flux_pending = flux_mode;  // rides with the next gradients (or the next sync)
    else if (launch_flux(g, flux_mode, g->s_main)) return 1;
  }
  if (scaled_tail(g, scaled_lag(g, with_flux), true, g->s_main)) return 1;
  return mark_main(g);
}

// ----------------------------------------------------------------- in-process rank group
// Phase 1 of rank a's iteration: gradients (+ pack + peer copies into the partners' ghost rows).  Two parts, so that G
// host threads can drive G devices: _launch enqueues rank a's own kernels (and swaps its grad buffers in fused mode),
// _send the copies into the partners -- once EVERY rank of the group has done its _launch (the caller's barrier), so
// that "the partner's ghost block of this iteration" is simply its current one.  cfdp_gpu_rank_gradients does both for a
// single caller that walks the ranks one after the other (a partner may then not have swapped yet: handled in place).
static int rank_launch(cfdp_gpu **ranks, int G, int a, int with_exchange, int overlap) {
  if (!ranks || G < 1 || a < 0 || a >= G) return fail("bad rank group");
  cfdp_gpu *ga = ranks[a];
  NEED_UPLOAD(ga);
  const bool comm = with_exchange && !ga->partner.empty();
  ga->pending_exchange = comm;
  ga->iter++;
  const bool fused = ga->will_fuse();  // see cfdp_gpu_step_pre
  if (!fused && flush_flux(ga)) return 1;
  auto grad_tiles = [&](int which, hipStream_t st) { return fused ? launch_fused(ga, which, st) : launch_grad(ga, which, st); };
  if (!comm) {
    if (grad_tiles(CFDP_TILES_ALL, ga->s_main)) return 1;
    if (fused) fused_done(ga);
    return 0;
  }
  const gg_grad_view src = fused ? ga->alt_view() : ga->grad_view();
  // the comm stream forks off the main stream here (after this rank's previous iteration); the send
  // arena is free by then too (last iteration's copies are earlier on the comm stream)
  if (fork_comm(ga)) return 1;
  if (overlap) {  // boundary tiles + pack + copies on the comm stream, interior tiles beside them (cfdp_gpu_step_pre)
    if (grad_tiles(CFDP_TILES_BOUNDARY, ga->s_comm)) return 1;
    HIP_TRY(gg_launch_pack(ga->d_sendidx, ga->send_off.back(), src, ga->d_sendbuf, ga->s_comm));
    if (grad_tiles(CFDP_TILES_INTERIOR, ga->s_main)) return 1;
  } else {
    if (grad_tiles(CFDP_TILES_ALL, ga->s_main)) return 1;
    HIP_TRY(hipStreamWaitEvent(ga->s_main, ga->ev_senddone, 0));  // arena free (copies of the last iteration)
    HIP_TRY(gg_launch_pack(ga->d_sendidx, ga->send_off.back(), src, ga->d_sendbuf, ga->s_main));
    HIP_TRY(hipEventRecord(ga->ev_pack, ga->s_main));
    HIP_TRY(hipStreamWaitEvent(ga->s_comm, ga->ev_pack, 0));
  }
  if (fused) fused_done(ga);
  return 0;
}

static int rank_send(cfdp_gpu **ranks, int G, int a) {
  if (!ranks || G < 1 || a < 0 || a >= G) return fail("bad rank group");
  cfdp_gpu *ga = ranks[a];
  NEED_UPLOAD(ga);
  if (!ga->pending_exchange) return 0;
  for (size_t s = 0; s < ga->partner.size(); s++) {
    const int b = ga->partner[s];
    if (b < 0 || b >= G) return fail("partner rank %d outside the in-process group", b);
    cfdp_gpu *gb = ranks[b];
    int slot = -1;
    for (size_t i = 0; i < gb->partner.size(); i++)
      if (gb->partner[i] == a) slot = (int)i;
    if (slot < 0) return fail("rank %d sends to %d which does not list it as partner", a, b);
    size_t sbytes = 0, rbytes = 0;
    void *from = cfdp_gpu_send_ptr(ga, (int)s, &sbytes);
    void *dst = cfdp_gpu_recv_ptr(gb, slot, &rbytes);
    // b's ghost block of THIS iteration: b has either done its launch part already (iter equal: its buffers are
    // swapped) or -- single caller walking the ranks in order -- will fuse, and swap, when it gets there
    if (gb->iter != ga->iter && gb->will_fuse())
      dst = gb->alt_view().ghost + (size_t)gb->recv_off[slot] * 21;
    if (sbytes != rbytes) return fail("halo size mismatch %d->%d: %zu vs %zu bytes", a, b, sbytes, rbytes);
    if (!sbytes) continue;
    // b's ghost rows may still be read by b's previous flux (write-after-read)
    HIP_TRY(hipStreamWaitEvent(ga->s_comm, gb->ev_fluxdone, 0));
    HIP_TRY(hipMemcpyPeerAsync(dst, gb->device, from, ga->device, sbytes, ga->s_comm));
  }
  HIP_TRY(hipEventRecord(ga->ev_senddone, ga->s_comm));
  return 0;
}

int cfdp_gpu_rank_gradients_launch(cfdp_gpu **ranks, int G, int a, int with_exchange, int overlap) {
  return rank_launch(ranks, G, a, with_exchange, overlap);
}
int cfdp_gpu_rank_gradients_send(cfdp_gpu **ranks, int G, int a) { return rank_send(ranks, G, a); }

int cfdp_gpu_rank_gradients(cfdp_gpu **ranks, int G, int a, int with_exchange, int overlap) {
  if (rank_launch(ranks, G, a, with_exchange, overlap)) return 1;
  return rank_send(ranks, G, a);
}

// direct loads / stores and copies between the devices of a group: hipDeviceEnablePeerAccess for every pair of ranks
// that live on different devices (already enabled is fine).  *npairs (optional) = pairs enabled or found enabled;
// a pair the runtime refuses is not an error (copies then travel through the host): the count says so.
int cfdp_gpu_enable_peer_access(cfdp_gpu **ranks, int G, int *npairs) {
  if (!ranks || G < 1) return fail("bad rank group");
  int n = 0;
  for (int a = 0; a < G; a++)
    for (int b = 0; b < G; b++) {
      if (!ranks[a] || !ranks[b] || ranks[a]->device == ranks[b]->device) continue;
      bool seen = false;  // one call per ordered device pair
      for (int c = 0; c < a && !seen; c++)
        for (int d = 0; d < G && !seen; d++)
          seen = ranks[c] && ranks[d] && ranks[c]->device == ranks[a]->device && ranks[d]->device == ranks[b]->device;
      if (seen) continue;
      int can = 0;
      HIP_TRY(hipDeviceCanAccessPeer(&can, ranks[a]->device, ranks[b]->device));
      if (!can) continue;
      HIP_TRY(hipSetDevice(ranks[a]->device));
      const hipError_t e = hipDeviceEnablePeerAccess(ranks[b]->device, 0);
      if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) n++;
      (void)hipGetLastError();
    }
  if (npairs) *npairs = n;
  return 0;
}

// Phase 2 of rank b's iteration: wait for the partners' copies, then the pseudo-flux loop.
int cfdp_gpu_rank_flux(cfdp_gpu **ranks, int G, int b, int with_flux, int flux_mode) {
  if (!ranks || G < 1 || b < 0 || b >= G) return fail("bad rank group");
  cfdp_gpu *gb = ranks[b];
  NEED_UPLOAD(gb);
  if (gb->pending_exchange) {
    HIP_TRY(hipStreamWaitEvent(gb->s_main, gb->ev_senddone, 0));  // this rank's own comm stream (boundary tiles, copies)
    for (int a : gb->partner) HIP_TRY(hipStreamWaitEvent(gb->s_main, ranks[a]->ev_senddone, 0));
  }
  gb->pending_exchange = false;
  if (with_flux) {
    if (flux_mode != CFDP_FLUX_CONSISTENT && flux_mode != CFDP_FLUX_REFERENCE) return fail("bad flux mode %d", flux_mode);
    if (gb->fusion && gb->d_grad_alt) gb->flux_pending = flux_mode;
    else if (launch_flux(gb, flux_mode, gb->s_main)) return 1;
  }
  if (scaled_tail(gb, scaled_lag(gb, with_flux), true, gb->s_main)) return 1;
  return mark_main(gb);
}

int cfdp_gpu_iteration_group(cfdp_gpu **ranks, int G, int with_exchange, int overlap, int with_flux,
                             int flux_mode) {
  for (int a = 0; a < G; a++)
    if (cfdp_gpu_rank_gradients(ranks, G, a, with_exchange, overlap)) return 1;
  for (int b = 0; b < G; b++)
    if (cfdp_gpu_rank_flux(ranks, G, b, with_flux, flux_mode)) return 1;
  return 0;
}

int cfdp_gpu_sync_group(cfdp_gpu **ranks, int G) {
  for (int a = 0; a < G; a++)
    if (cfdp_gpu_sync(ranks[a])) return 1;
  return 0;
}

// ----------------------------------------------------- scaled-field validation of the exchange
// See gg_validate_kernel.  begin: the flux the context holds NOW (from an iteration whose exchange the caller knows to be
// complete: device syncs and a barrier between the ranks, then one step without exchange) becomes the reference; from
// then on every step entry point (cfdp_gpu_step_post, _step_ipc*, _run_steps_*, _rank_flux; the drop-in layer's
// compute_psd_flux) ends with the validation kernel: compare the flux the step produced with reference * 2^e, then
// var *= 2, 2, 1/4, ...  Every step must exchange and compute the flux while the mode is on.  end: the deferred flux of
// the last iteration is compared too, var is restored exactly, the evidence is returned.
static void drop_ipc_graphs(cfdp_gpu *g) {
  auto &I = g->ipc;
  if (I.graph) { (void)hipGraphExecDestroy(I.graph); I.graph = nullptr; }
  if (I.graph_rem) { (void)hipGraphExecDestroy(I.graph_rem); I.graph_rem = nullptr; }
  I.graph_n = I.graph_rem_n = 0;
}

int cfdp_gpu_scaled_check_begin(cfdp_gpu *g) {
  NEED_UPLOAD(g);
  if (g->sc.on) return fail("the scaled-field validation is already on");
  if (flush_flux(g)) return 1;
  HIP_TRY(hipDeviceSynchronize());
  const size_t nf = (size_t)g->nown * 3;
  if (!g->sc.d_state) HIP_TRY(hipMalloc(&g->sc.d_state, GG_V_WORDS * sizeof(int)));
  if (!g->sc.d_fref) HIP_TRY(hipMalloc(&g->sc.d_fref, (nf + 1) * sizeof(double)));
  if (!g->sc.d_skip && !g->faceless.empty()) {
    std::vector<unsigned char> skip((size_t)g->nown, 0);
    for (int i : g->faceless) skip[(size_t)i] = 1;
    HIP_TRY(hipMalloc(&g->sc.d_skip, skip.size()));
    HIP_TRY(hipMemcpy(g->sc.d_skip, skip.data(), skip.size(), hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMemset(g->sc.d_state, 0, GG_V_WORDS * sizeof(int)));
  // bit for bit: the reference must come from the kernel form the steps will use.  The flux phase of the fused pass
  // sums a point's faces on 4 lanes, the separate flux kernel by default on 8 (another association): with fused
  // iterations on, the separate kernel -- reference now, last iteration's deferred flux later -- runs on 4 as well
  g->sc.saved_flux_lanes = g->flux_lanes;
  if (g->fusion && g->d_grad_alt) g->flux_lanes = 4;
  if (launch_flux(g, g->last_flux_mode, g->s_main)) return 1;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(g->sc.d_fref, g->d_flux, nf * sizeof(double), hipMemcpyDeviceToDevice));
  g->sc.on = true;
  g->drop_graphs();  // graphs captured without the validation kernel
  drop_ipc_graphs(g);
  return 0;
}

int cfdp_gpu_scaled_check_end(cfdp_gpu *g, cfdp_scaled_check *out) {
  NEED_UPLOAD(g);
  if (!g->sc.on) return fail("the scaled-field validation is not on");
  const int rc = flush_flux(g);  // compares the last iteration's flux as well
  g->sc.on = false;
  g->flux_lanes = g->sc.saved_flux_lanes;
  g->drop_graphs();
  drop_ipc_graphs(g);
  if (rc) return 1;
  HIP_TRY(hipDeviceSynchronize());
  int st[GG_V_WORDS];
  HIP_TRY(hipMemcpy(st, g->sc.d_state, sizeof st, hipMemcpyDeviceToHost));
  const int m = st[GG_V_ITER] % 3;  // var holds var0 * 2^m
  if (m) {
    HIP_TRY(gg_launch_scale_var(g->d_var, g->nall, m == 1 ? 0.5 : 0.25, g->s_main));
    HIP_TRY(hipDeviceSynchronize());
  }
  if (out) {
    unsigned long long bad = 0;
    memcpy(&bad, &st[GG_V_BAD], sizeof bad);
    out->iterations = st[GG_V_ITER];
    out->flux_checks = st[GG_V_CHECKS];
    out->mismatches = bad > 0x7FFFFFFFull ? 0x7FFFFFFF : (int)bad;
    out->first_iteration = st[GG_V_CLAIM] ? st[GG_V_FIRST_ITER] : 0;
    out->first_point = st[GG_V_CLAIM] ? g->new2old[(size_t)(st[GG_V_FIRST_IDX] / 3)] : -1;
    out->first_component = st[GG_V_CLAIM] ? st[GG_V_FIRST_IDX] % 3 : -1;
    out->seen = out->expected = 0.0;
    if (st[GG_V_CLAIM]) {
      memcpy(&out->seen, &st[GG_V_SEEN], sizeof(double));
      memcpy(&out->expected, &st[GG_V_EXPECT], sizeof(double));
    }
  }
  return 0;
}

// --------------------------------------------------------------------------- measurement
int cfdp_gpu_time_kernels(cfdp_gpu *g, int iters, int flux_mode, float *ms_grad, float *ms_flux) {
  NEED_UPLOAD(g);
  if (iters < 1) return fail("iters must be >= 1");
  if (flush_flux(g)) return 1;
  hipStream_t st = g->s_main;
  for (int w = 0; w < 2; w++) {
    if (launch_grad(g, CFDP_TILES_ALL, st)) return 1;
    if (launch_flux(g, flux_mode, st)) return 1;
  }
  HIP_TRY(hipEventRecord(g->ev_a, st));
  for (int i = 0; i < iters; i++)
    if (launch_grad(g, CFDP_TILES_ALL, st)) return 1;
  HIP_TRY(hipEventRecord(g->ev_b, st));
  HIP_TRY(hipEventSynchronize(g->ev_b));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, g->ev_a, g->ev_b));
  if (ms_grad) *ms_grad = ms / (float)iters;
  HIP_TRY(hipEventRecord(g->ev_a, st));
  for (int i = 0; i < iters; i++)
    if (launch_flux(g, flux_mode, st)) return 1;
  HIP_TRY(hipEventRecord(g->ev_b, st));
  HIP_TRY(hipEventSynchronize(g->ev_b));
  HIP_TRY(hipEventElapsedTime(&ms, g->ev_a, g->ev_b));
  if (ms_flux) *ms_flux = ms / (float)iters;
  return 0;
}

// n iterations of one partition on stream `st`, no exchange.  Fused mode: gradients(1), then n-1
// passes of flux(i) + gradients(i+1), then flux(n) -- the same values as n x (gradients, flux),
// with the tile blobs streamed once per iteration.  An odd n leaves the two grad buffers where
// they were.
static int enqueue_iterations(cfdp_gpu *g, int n, int with_flux, int flux_mode, hipStream_t st) {
  const bool fuse = g->fusion && g->d_grad_alt && with_flux;
  for (int i = 0; i < n; i++) {
    if (fuse && g->flux_pending >= 0) {
      if (launch_fused(g, CFDP_TILES_ALL, st)) return 1;
      fused_done(g);
    } else if (launch_grad(g, CFDP_TILES_ALL, st)) {
      return 1;
    }
    if (fuse) g->flux_pending = flux_mode;
    else if (with_flux && launch_flux(g, flux_mode, st)) return 1;
  }
  return flush_flux(g, false, st);
}

int cfdp_gpu_time_fused(cfdp_gpu *g, int iters, int flux_mode, float *ms_fused) {
  NEED_UPLOAD(g);
  if (iters < 1) return fail("iters must be >= 1");
  if (!g->fusion || !g->d_grad_alt) return fail("fusion is off");
  if (flux_mode != CFDP_FLUX_CONSISTENT && flux_mode != CFDP_FLUX_REFERENCE) return fail("bad flux mode %d", flux_mode);
  if (flush_flux(g)) return 1;
  hipStream_t st = g->s_main;
  if (launch_grad(g, CFDP_TILES_ALL, st)) return 1;
  auto pass = [&]() -> int {
    g->flux_pending = flux_mode;
    if (launch_fused(g, CFDP_TILES_ALL, st)) return 1;
    fused_done(g);
    return 0;
  };
  for (int w = 0; w < 2; w++) if (pass()) return 1;
  // timed as the iteration loop runs it: an even number of passes replayed from one hipGraph
  // (stream launches add ~4 us of dependent-launch gap to every kernel)
  iters += iters & 1;
  hipGraph_t gr = nullptr;
  hipGraphExec_t ge = nullptr;
  HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  int rc = 0;
  for (int i = 0; i < iters && !rc; i++) rc = pass();
  hipError_t ec = hipStreamEndCapture(st, &gr);
  if (rc) { if (gr) (void)hipGraphDestroy(gr); return 1; }
  HIP_TRY(ec);
  HIP_TRY(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
  HIP_TRY(hipGraphDestroy(gr));
  g->main_marked = false;
  HIP_TRY(hipGraphLaunch(ge, st));
  HIP_TRY(hipEventRecord(g->ev_a, st));
  HIP_TRY(hipGraphLaunch(ge, st));
  HIP_TRY(hipEventRecord(g->ev_b, st));
  HIP_TRY(hipEventSynchronize(g->ev_b));
  HIP_TRY(hipGraphExecDestroy(ge));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, g->ev_a, g->ev_b));
  if (ms_fused) *ms_fused = ms / (float)iters;
  g->flux_pending = flux_mode;  // the last pass's gradients still owe their flux
  return flush_flux(g);
}

// ---- tile-resident iterations: K iterations of one partition in ONE launch (gg_resident_kernel)
// Qualifies: every tile co-resident on the device, tile sizes within the phase-split pass's capacity, fixed-stride
// row lists, at most 64 neighbour tiles per tile, both grad buffers bound.  cfdp_gpu_run_iterations never exchanges
// (a partition with partners keeps its ghost rows as they are, like compute_gradients_gg_comm_free); the in-kernel
// exchange of cfdp_gpu_run_steps_ipc rides in the one-launch-per-pass form only.
static bool resident_qualifies(const cfdp_gpu *g, const char **why) {
  const char *w = nullptr;
  const int tp = g->tp[0] > g->tp[1] ? g->tp[0] : g->tp[1];
  const int mr = g->max_rows[0] > g->max_rows[1] ? g->max_rows[0] : g->max_rows[1];
  const int mb = g->max_blob[0] > g->max_blob[1] ? g->max_blob[0] : g->max_blob[1];
  if (!g->d_grad_alt) w = "fused iterations are off (no second grad buffer)";
  else if (g->ipc.on) w = "xGMI landing arenas in use (the resident kernel reads the grad buffers' own ghost blocks)";
  else if (!g->d_rowlist) w = "no fixed-stride row lists (tiles of more than 64 points or 204 rows)";
  else if (tp > 64 || !gg_resident_fits(64, mr - 64 > 0 ? mr - 64 : 0, mb)) w = "a tile exceeds the 36-KiB LDS image";
  else if (g->max_nbr > 64) w = "a tile has more than 64 neighbour tiles";
  else if (g->ntiles > gg_resident_capacity()) w = "more tiles than the device holds workgroups at once";
  if (why) *why = w;
  return w == nullptr;
}

static long resident_max_polls() {
  double sec = 2.0;
  if (const char *e = getenv("CFDP_RESIDENT_WAIT_SECONDS")) sec = atof(e);
  return (long)(sec * 1e6 / 0.4) + 1;  // a poll + s_sleep(8) round is >= 0.4 us
}

static int run_resident(cfdp_gpu *g, int iters, int with_flux, int flux_mode, float *ms_total) {
  hipStream_t st = g->s_main;
  if (flush_flux(g)) return 1;
  gg_resident_args ra;
  ra.nbr_off = g->d_nbr_off; ra.nbr = g->d_nbr;
  ra.err = g->d_resident_state; ra.flags = g->d_resident_state + 4;
  ra.max_polls = resident_max_polls();
  ra.iters = iters; ra.with_flux = with_flux;
  const size_t state_bytes = (((size_t)g->ntiles + 4 + 3) & ~(size_t)3) * sizeof(int);
  HIP_TRY(hipEventRecord(g->ev_a, st));
  HIP_TRY(hipMemsetAsync(g->d_resident_state, 0, state_bytes, st));
  g->main_marked = false;
  const gg_args a = g->args();
  HIP_TRY(gg_launch_resident(a, g->grad_view(), g->alt_view(), flux_mode == CFDP_FLUX_REFERENCE, g->ntiles, ra,
                             g->resident == 2, st));
  HIP_TRY(hipEventRecord(g->ev_b, st));
  HIP_TRY(hipEventSynchronize(g->ev_b));
  if ((iters - 1) & 1) {  // G_k wrote buffer (k - 1) & 1: the last gradients are in the other buffer
    std::swap(g->d_grad, g->d_grad_alt);
    std::swap(g->own_grad, g->own_grad_alt);
  }
  g->flux_pending = -1;
  g->resident_runs++;
  int err[4] = {0, 0, 0, 0};
  HIP_TRY(hipMemcpy(err, g->d_resident_state, sizeof err, hipMemcpyDeviceToHost));
  if (err[0])
    return fail("tile-resident iterations: tile %d gave up waiting for its neighbours' iteration %d (saw %d) -- the grid "
                "was not fully resident; results are invalid", err[1], err[2], err[3]);
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, g->ev_a, g->ev_b));
  if (ms_total) *ms_total = ms;
  return 0;
}

int cfdp_gpu_set_resident(cfdp_gpu *g, int mode) {
  if (!g) return fail("null context");
  if (mode < 0 || mode > 2) return fail("resident mode must be 0 (off), 1 (on where the partition qualifies) or 2 (staleness test)");
  g->resident = mode;
  return 0;
}

// 1: run_iterations would run the tile-resident kernel; 0: it would not, *why says why (static text)
int cfdp_gpu_resident_qualifies(cfdp_gpu *g, const char **why) {
  if (!g || !g->uploaded) { if (why) *why = "no plan uploaded"; return 0; }
  if (hipSetDevice(g->device) != hipSuccess) { if (why) *why = "device"; return 0; }
  return resident_qualifies(g, why) ? 1 : 0;
}

// The data-movement floor of the fused pass: the same kernel with neither face loop (a diagnostic instantiation:
// every load, every store, zeros as results), timed exactly as cfdp_gpu_time_fused times the real pass.  What the
// real pass takes beyond it is arithmetic and latency its resident tiles do not hide.  Leaves grad / flux holding one
// correct iteration again.
int cfdp_gpu_time_fused_movement(cfdp_gpu *g, int iters, float *ms_pass) {
  NEED_UPLOAD(g);
  if (iters < 1) return fail("iters must be >= 1");
  if (!g->fusion || !g->d_grad_alt) return fail("fusion is off");
  if (!g->d_rowlist) return fail("no fixed-stride row lists: the movement-only instantiation needs them");
  {  // the movement-only kernel is an instantiation of the phase-split form: refuse where that form would not run (the
     // launch would otherwise fall back to the two REAL kernels and their time be reported as the floor)
    const tile_range r = range_of(g, CFDP_TILES_ALL);
    const int block = ((r.tp * 4 + 63) / 64) * 64;
    const int cb = (r.max_blob + block - 1) / block, kv = ((r.tp + r.row_halo()) * 4 + block - 1) / block,
              kg = ((r.tp + r.row_halo()) * 5 + block - 1) / block;
    if (!gg_fused_split || g->beside_rccl || block > 1024 || cb > 5 || kv > 4 || kg > 4)
      return fail("the phase-split fused pass would not run on this partition (CFDP_FUSED_SPLIT=0, or tiles beyond its capacity): "
                  "no movement-only form to time");
  }
  if (flush_flux(g)) return 1;
  hipStream_t st = g->s_main;
  const int dbg0 = gg_debug_flags;
  gg_debug_flags |= 0x40000;  // GG_DBG_MOVE
  auto pass = [&]() -> int {
    g->flux_pending = CFDP_FLUX_CONSISTENT;
    if (launch_fused(g, CFDP_TILES_ALL, st)) return 1;
    fused_done(g);
    return 0;
  };
  int rc = 0;
  for (int w = 0; w < 2 && !rc; w++) rc = pass();
  iters += iters & 1;
  hipGraph_t gr = nullptr;
  hipGraphExec_t ge = nullptr;
  hipError_t ec = hipSuccess;
  if (!rc) {
    ec = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    if (ec == hipSuccess) {
      for (int i = 0; i < iters && !rc; i++) rc = pass();
      ec = hipStreamEndCapture(st, &gr);
    }
  }
  gg_debug_flags = dbg0;
  g->flux_pending = -1;
  if (rc) { if (gr) (void)hipGraphDestroy(gr); return 1; }
  HIP_TRY(ec);
  HIP_TRY(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
  HIP_TRY(hipGraphDestroy(gr));
  g->main_marked = false;
  HIP_TRY(hipGraphLaunch(ge, st));
  HIP_TRY(hipEventRecord(g->ev_a, st));
  HIP_TRY(hipGraphLaunch(ge, st));
  HIP_TRY(hipEventRecord(g->ev_b, st));
  HIP_TRY(hipEventSynchronize(g->ev_b));
  HIP_TRY(hipGraphExecDestroy(ge));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, g->ev_a, g->ev_b));
  if (ms_pass) *ms_pass = ms / (float)iters;
  g->drop_graphs();
  if (launch_grad(g, CFDP_TILES_ALL, st)) return 1;  // real values again
  g->flux_pending = CFDP_FLUX_CONSISTENT;
  return flush_flux(g);
}

// K iterations of one partition, replayed from hipGraphs whatever K is.
// Fused mode: gradients(1), K-1 fused passes (flux(i) + gradients(i+1)), flux(K).  The passes are cut
// into whole chunks of 50 (one graph, replayed), a remainder graph with the even part of what is left
// (an even count leaves the two grad buffers in place -- the graph's pointers are baked in) and at most
// one stream-launched pass.  Separate kernels: chunks of 25 iterations = NITER of the reference harness
// (src/hybrid.f6.c:72), a remainder graph for the rest.  run = false: capture and instantiate only
// (cfdp_gpu_prepare_iterations) -- nothing executes, so a timed region need not contain a capture.
static int run_or_prepare_iterations(cfdp_gpu *g, int iters, int with_flux, int flux_mode, int use_graph, bool run,
                                     float *ms_total) {
  if (iters < 1) return fail("iters must be >= 1");
  if (with_flux && flux_mode != CFDP_FLUX_CONSISTENT && flux_mode != CFDP_FLUX_REFERENCE)
    return fail("bad flux mode %d", flux_mode);
  if (g->resident && g->fusion && resident_qualifies(g, nullptr)) {
    if (!run) return 0;  // nothing to capture: the run is one launch
    if (run_resident(g, iters, with_flux, flux_mode, ms_total) == 0) return 0;
    // a neighbour wait gave up: the grid was not co-resident after all (other work on the device: ranks sharing it,
    // another stream, a profiler).  Say so, switch tile residency off for this context and run the K iterations the
    // usual way -- every gradient is recomputed from var, so nothing of the failed launch survives
    fprintf(stderr, "[cfdp] %s -- tile-resident iterations are off for this context from here on; re-running from hipGraphs\n", g_err);
    g->resident = 0;
    HIP_TRY(hipDeviceSynchronize());
  }
  if (flush_flux(g)) return 1;
  hipStream_t st = g->s_main;
  const bool fuse = g->fusion && g->d_grad_alt && with_flux;
  // units: fused passes (after the leading gradient launch) or whole iterations
  const int units = fuse ? iters - 1 : iters;
  const int full = fuse ? 50 : 25;
  // a SHORT fused run is one graph from its first gradient launch to its last flux launch: the two
  // un-fused launches at its ends and a graph's start-up latency (~20 us before its first kernel runs)
  // are a tenth of a 20-iteration run when they sit between stream launches and replays
  const bool whole = use_graph && fuse && iters >= 2 && iters <= 64;
  int chunk = 0, nchunk = 0, rem = 0;
  if (use_graph && !whole) {
    chunk = units >= full ? full : 0;
    nchunk = chunk ? units / chunk : 0;
    rem = units - nchunk * chunk;
    if (fuse) rem &= ~1;
    if (rem < 2) rem = 0;
  }
  const int eager = whole ? 0 : units - nchunk * chunk - rem;
  auto one_pass = [&]() -> int {  // fused mode, a flux pending
    if (launch_fused(g, CFDP_TILES_ALL, st)) return 1;
    fused_done(g);
    g->flux_pending = flux_mode;
    return 0;
  };
  auto body = [&](int n) -> int {
    if (!fuse) return enqueue_iterations(g, n, with_flux, flux_mode, st);
    for (int i = 0; i < n; i++)
      if (one_pass()) return 1;
    return 0;
  };
  auto whole_run = [&]() -> int {  // gradients(1), iters-1 fused passes, flux(iters)
    if (launch_grad(g, CFDP_TILES_ALL, st)) return 1;
    g->flux_pending = flux_mode;
    if (body(iters - 1)) return 1;
    return flush_flux(g, false, st);
  };
  if (g->graph_flux != with_flux || g->graph_mode != flux_mode || g->graph_gl != g->grad_lanes ||
      g->graph_fl != g->flux_lanes || g->graph_fuse != (int)fuse ||
      (g->graph_cur != g->d_grad && g->graph_cur != g->d_grad_alt))
    g->drop_graphs();  // every cached graph was captured for another configuration
  // slot_n > 0: n units that leave the grad buffers in place; slot_n < 0: a whole run of -n iterations, which
  // starts by recomputing every gradient and may end in either buffer (graph_whole_final says which)
  auto capture = [&](hipGraphExec_t &slot, int &slot_n, int n) -> int {  // 0 ok, 1 error
    const double *&slot_cur = g->graph_cur_slot[&slot == &g->graph ? 0 : 1];
    if (slot && slot_n == n && (n < 0 || slot_cur == g->d_grad)) return 0;
    if (slot) { (void)hipGraphExecDestroy(slot); slot = nullptr; }
    slot_n = 0;
    hipGraph_t gr = nullptr;
    const double *cur0 = g->d_grad;
    const int pend0 = g->flux_pending;
    const unsigned passes0 = g->fused_passes;
    if (fuse && n > 0) g->flux_pending = flux_mode;  // the state every pass of the run starts in
    HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = n < 0 ? whole_run() : body(n);
    hipError_t ec = hipStreamEndCapture(st, &gr);
    g->flux_pending = pend0;  // nothing of the capture has run
    g->fused_passes = passes0;
    const bool swapped = g->d_grad != cur0;
    if (swapped) { std::swap(g->d_grad, g->d_grad_alt); std::swap(g->own_grad, g->own_grad_alt); }
    if (rc) { if (gr) (void)hipGraphDestroy(gr); return 1; }
    HIP_TRY(ec);
    if (swapped && n > 0) { (void)hipGraphDestroy(gr); return fail("graph chunk must leave the grad buffers in place"); }
    HIP_TRY(hipGraphInstantiate(&slot, gr, nullptr, nullptr, 0));
    HIP_TRY(hipGraphDestroy(gr));
    slot_n = n;
    if (n < 0) g->graph_whole_final = swapped ? g->d_grad_alt : g->d_grad;
    g->graph_flux = with_flux; g->graph_mode = flux_mode;
    g->graph_gl = g->grad_lanes; g->graph_fl = g->flux_lanes;
    g->graph_fuse = (int)fuse; g->graph_cur = g->d_grad;
    slot_cur = g->d_grad;
    return 0;
  };
  // (an even number of alternating-direction passes per graph: a replay continues the alternation)
  if (whole && capture(g->graph_rem, g->graph_rem_iters, -iters)) return 1;
  if (chunk && capture(g->graph, g->graph_iters, chunk)) return 1;
  if (rem && capture(g->graph_rem, g->graph_rem_iters, rem)) return 1;
  if (!run) return 0;
  HIP_TRY(hipEventRecord(g->ev_a, st));
  if (whole) {
    g->main_marked = false;
    HIP_TRY(hipGraphLaunch(g->graph_rem, st));
    if (g->d_grad != g->graph_whole_final) {  // the run's last gradients are where the graph put them
      std::swap(g->d_grad, g->d_grad_alt);
      std::swap(g->own_grad, g->own_grad_alt);
    }
    g->graph_cur = g->d_grad;
  } else {
    if (fuse) {
      if (launch_grad(g, CFDP_TILES_ALL, st)) return 1;  // iteration 1's gradients; its flux rides with iteration 2
      g->flux_pending = flux_mode;
    }
    for (int c = 0; c < nchunk; c++) {
      g->main_marked = false;
      HIP_TRY(hipGraphLaunch(g->graph, st));
    }
    if (rem) {
      g->main_marked = false;
      HIP_TRY(hipGraphLaunch(g->graph_rem, st));
    }
    if (eager && body(eager)) return 1;
    if (fuse && flush_flux(g, false, st)) return 1;  // flux of the last iteration
  }
  HIP_TRY(hipEventRecord(g->ev_b, st));
  HIP_TRY(hipEventSynchronize(g->ev_b));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, g->ev_a, g->ev_b));
  if (ms_total) *ms_total = ms;
  return 0;
}

int cfdp_gpu_run_iterations(cfdp_gpu *g, int iters, int with_flux, int flux_mode, int use_graph,
                            float *ms_total) {
  NEED_UPLOAD(g);
  return run_or_prepare_iterations(g, iters, with_flux, flux_mode, use_graph, true, ms_total);
}

int cfdp_gpu_prepare_iterations(cfdp_gpu *g, int iters, int with_flux, int flux_mode) {
  NEED_UPLOAD(g);
  return run_or_prepare_iterations(g, iters, with_flux, flux_mode, 1, false, nullptr);
}

// diagnostics (tools/phase_stamps.py): run `passes` fused passes with phase stamping on and return, per tile, 8
// shader-clock stamps of the LAST pass (start, indices here, loads landed, flux done, var rows in place, gradient
// arithmetic done + stores issued, stores acknowledged, unused) and, behind them, 4 per wave of the tile (own pieces
// landed, through the flux phase, through the gradient phase, unused); stamps[ntiles*24]
int cfdp_gpu_debug_phase_stamps(cfdp_gpu *g, int passes, unsigned long long *stamps) {
  NEED_UPLOAD(g);
  if (!g->fusion || !g->d_grad_alt || passes < 1 || !stamps) return fail("fusion must be on");
  if (flush_flux(g)) return 1;
  unsigned long long *d = nullptr;
  const size_t n = (size_t)g->ntiles * 24;  // 8 per tile, then 4 x 4 per wave
  HIP_TRY(hipMalloc(&d, n * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(d, 0, n * sizeof(unsigned long long)));
  HIP_TRY(gg_set_stamp_buffer(d));
  if (launch_grad(g, CFDP_TILES_ALL, g->s_main)) return 1;
  const int saved = gg_debug_flags;
  gg_debug_flags |= 0x20000;
  int rc = 0;
  for (int i = 0; i < passes && !rc; i++) {
    g->flux_pending = CFDP_FLUX_CONSISTENT;
    rc = launch_fused(g, CFDP_TILES_ALL, g->s_main);
    if (!rc) fused_done(g);
  }
  gg_debug_flags = saved;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(gg_set_stamp_buffer(nullptr));
  if (!rc) HIP_TRY(hipMemcpy(stamps, d, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  (void)hipFree(d);
  if (rc) return 1;
  g->flux_pending = CFDP_FLUX_CONSISTENT;
  return flush_flux(g);
}

// ------------------------------------------------------- one process per GPU: RCCL from C
// The halo exchange of a step issued straight from this library: one ncclGroup of
// ncclSend/ncclRecv per iteration on the context's comm stream, between the two step brackets
// -- the analogue of exchange_dbl_mpi_send / _post_recv (src/exchange_data_mpi.c:96-166) with
// the receive side being the ghost block itself.  A host pays ONE call per iteration (or one
// per hipGraph replay of several).  RCCL is resolved at run time from the library the process
// already uses (PyTorch ships its own librccl.so; a C host names the system one), so this
// library has no link-time dependency on it.
namespace {
struct rccl_api {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
} rccl;

#define RCCL_TRY(expr)                                                                              \
  do {                                                                                              \
    ncclResult_t r_ = (expr);                                                                       \
    if (r_ != ncclSuccess)                                                                          \
      return fail("%s failed: %s [%s:%d]", #expr, rccl.GetErrorString ? rccl.GetErrorString(r_) : "?", \
                  __FILE__, __LINE__);                                                              \
  } while (0)

// this iteration's messages: sends from the packed arena, receives into the current ghost block
int enqueue_exchange(cfdp_gpu *g) {
  RCCL_TRY(rccl.GroupStart());
  for (size_t s = 0; s < g->partner.size(); s++) {
    size_t sb = 0, rb = 0;
    void *sp = cfdp_gpu_send_ptr(g, (int)s, &sb), *rp = cfdp_gpu_recv_ptr(g, (int)s, &rb);
    if (sb) RCCL_TRY(rccl.Send(sp, sb / sizeof(double), ncclDouble, g->peer[s], g->comm, g->s_comm));
    if (rb) RCCL_TRY(rccl.Recv(rp, rb / sizeof(double), ncclDouble, g->peer[s], g->comm, g->s_comm));
  }
  RCCL_TRY(rccl.GroupEnd());
  return 0;
}

int one_step(cfdp_gpu *g, int with_exchange, int overlap, int with_flux, int flux_mode) {
  if (cfdp_gpu_step_pre(g, with_exchange, overlap)) return 1;
  if (g->pending_exchange && enqueue_exchange(g)) return 1;
  return cfdp_gpu_step_post(g, with_flux, flux_mode);
}
}  // namespace

int cfdp_rccl_load(const char *libpath) {
  if (rccl.lib) return 0;
  const char *names[] = {libpath, "librccl.so.1", "librccl.so"};
  for (const char *n : names) {
    if (!n || !*n) continue;
    rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (rccl.lib) break;
  }
  if (!rccl.lib) return fail("cannot load RCCL (%s): %s", libpath ? libpath : "librccl.so.1", dlerror());
#define RCCL_SYM(field, name)                                                      \
  do {                                                                             \
    *(void **)(&rccl.field) = dlsym(rccl.lib, name);                               \
    if (!rccl.field) { rccl.lib = nullptr; return fail("RCCL symbol %s not found", name); } \
  } while (0)
  RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
  RCCL_SYM(CommInitRank, "ncclCommInitRank");
  RCCL_SYM(CommDestroy, "ncclCommDestroy");
  RCCL_SYM(GroupStart, "ncclGroupStart");
  RCCL_SYM(GroupEnd, "ncclGroupEnd");
  RCCL_SYM(Send, "ncclSend");
  RCCL_SYM(Recv, "ncclRecv");
  RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RCCL_SYM
  return 0;
}

int cfdp_rccl_unique_id(void *id128) {
  if (!rccl.lib) return fail("cfdp_rccl_load() has not been called");
  if (!id128) return fail("null argument");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  RCCL_TRY(rccl.GetUniqueId(static_cast<ncclUniqueId *>(id128)));
  return 0;
}

int cfdp_gpu_rccl_init(cfdp_gpu *g, const void *id128, int nranks, int rank, const int *rank_of_partner) {
  NEED_UPLOAD(g);
  if (!rccl.lib) return fail("cfdp_rccl_load() has not been called");
  if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail("bad communicator arguments");
  if (g->comm) return fail("this context already has a communicator");
  g->peer.resize(g->partner.size());
  for (size_t s = 0; s < g->partner.size(); s++) {
    g->peer[s] = rank_of_partner ? rank_of_partner[s] : g->partner[s];
    if (g->peer[s] < 0 || g->peer[s] >= nranks)
      return fail("partner %d maps to communicator rank %d outside [0,%d)", g->partner[s], g->peer[s], nranks);
  }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  RCCL_TRY(rccl.CommInitRank(&g->comm, nranks, id, rank));
  return 0;
}

int cfdp_gpu_rccl_finalize(cfdp_gpu *g) {
  if (!g) return fail("null context");
  if (g->comm) {
    HIP_TRY(hipSetDevice(g->device));
    HIP_TRY(hipDeviceSynchronize());
    RCCL_TRY(rccl.CommDestroy(g->comm));
    g->comm = nullptr;
  }
  return 0;
}

// the RCCL group of the iteration opened by cfdp_gpu_step_pre (nothing if that step has no exchange)
int cfdp_gpu_exchange_rccl(cfdp_gpu *g) {
  NEED_UPLOAD(g);
  if (!g->pending_exchange) return 0;
  if (!g->comm) return fail("no communicator: call cfdp_gpu_rccl_init()");
  return enqueue_exchange(g);
}

int cfdp_gpu_step_rccl(cfdp_gpu *g, int with_exchange, int overlap, int with_flux, int flux_mode) {
  NEED_UPLOAD(g);
  if (with_exchange && !g->partner.empty() && !g->comm) return fail("no communicator: call cfdp_gpu_rccl_init()");
  return one_step(g, with_exchange, overlap, with_flux, flux_mode);
}

// `steps` iterations enqueued by one call.  (Capturing the step -- RCCL group included -- in a
// hipGraph was tried: with RCCL 2.26.6 / ROCm 7.0 hipStreamEndCapture crashes once ncclSend/
// ncclRecv have been captured, so the steps are stream launches: ~54 us of host time each.)
int cfdp_gpu_run_steps_rccl(cfdp_gpu *g, int steps, int with_exchange, int overlap, int with_flux,
                            int flux_mode) {
  NEED_UPLOAD(g);
  if (steps < 1) return fail("steps must be >= 1");
  if (with_exchange && !g->partner.empty() && !g->comm) return fail("no communicator: call cfdp_gpu_rccl_init()");
  for (int i = 0; i < steps; i++)
    if (one_step(g, with_exchange, overlap, with_flux, flux_mode)) return 1;
  return 0;
}

// --------------------------------------- one process per GPU: xGMI write + notify (HIP IPC)
// See gg_push_kernel.  Setup: every rank exports its block (cfdp_gpu_ipc_export), the host
// exchanges the 64-byte handles and tells each rank, per partner slot, where in the partner's
// block its rows land for either parity and where its arrival counter is (cfdp_gpu_ipc_connect);
// cfdp_gpu_ipc_ready uploads the tables and switches the context's ghost block to the landing
// arenas.  A step needs no communication library and no host involvement beyond kernel launches,
// so a run of steps is replayed from one hipGraph.
namespace {
// polls (~1 us each) before a device-side wait for a partner gives up: about 10 s by default
double g_ipc_wait_seconds = 0.0;
long ipc_max_polls() {
  if (g_ipc_wait_seconds <= 0.0) {
    const char *e = getenv("CFDP_IPC_WAIT_SECONDS");
    g_ipc_wait_seconds = e && atof(e) > 0 ? atof(e) : 10.0;
  }
  return (long)(g_ipc_wait_seconds * 1e6);
}

}  // namespace
static long ipc_max_polls_c() { return ipc_max_polls(); }
namespace {

void ipc_release(cfdp_gpu *g) {
  auto &I = g->ipc;
  I.wait_pending = false;
  if (I.graph) { (void)hipGraphExecDestroy(I.graph); I.graph = nullptr; }
  if (I.graph_rem) { (void)hipGraphExecDestroy(I.graph_rem); I.graph_rem = nullptr; }
  I.graph_n = I.graph_rem_n = 0;
  for (void *p : I.opened) (void)hipIpcCloseMemHandle(p);
  I.opened.clear(); I.opened_handle.clear();
  for (int par = 0; par < 2; par++) { (void)hipFree(I.d_dst[par]); I.d_dst[par] = nullptr; I.dst[par].clear(); }
  (void)hipFree(I.d_rflag); I.d_rflag = nullptr; I.rflag.clear();
  (void)hipFree(I.d_slot_of_row); (void)hipFree(I.d_send_off);
  (void)hipFree(I.d_tile_off); (void)hipFree(I.d_ent); (void)hipFree(I.d_ent_row);
  (void)hipFree(I.d_pt_first); (void)hipFree(I.d_tile_xoff);
  I.d_pt_first = nullptr; I.d_tile_xoff = nullptr;
  I.d_slot_of_row = I.d_send_off = I.d_tile_off = I.d_ent = I.d_ent_row = nullptr;
  I.inkernel = false;
  (void)hipFree(I.d_done); (void)hipFree(I.d_need); (void)hipFree(I.d_tile_mask);
  I.d_done = I.d_need = nullptr; I.d_tile_mask = nullptr; I.per_partner = false;
  (void)hipFree(I.flags); I.flags = nullptr;
  (void)hipFree(I.block); I.block = nullptr;
  I.on = false; I.xiter = 0;
}

int ipc_pre(cfdp_gpu *g, int with_exchange, int overlap) {
  const bool comm = with_exchange && !g->partner.empty();
  g->pending_exchange = false;
  g->iter++;
  const bool fused = g->will_fuse();
  if (!fused && flush_flux(g)) return 1;
  auto grad_tiles = [&](int which, hipStream_t st) { return fused ? launch_fused(g, which, st) : launch_grad(g, which, st); };
  // a pending wait for the previous exchange is absorbed by the boundary tiles of a pushing fused pass;
  // every other schedule reads ghost rows without that check and needs the wait kernel first
  if (!(comm && fused && g->ipc.inkernel && g->ipc.wait_inkernel) && ipc_settle(g)) return 1;
  if (!comm) {
    if (grad_tiles(CFDP_TILES_ALL, g->s_main)) return 1;
    if (fused) fused_done(g);
  } else {
    auto &I = g->ipc;
    const int nslots = (int)g->partner.size(), par = (int)((I.xiter + 1) & 1);
    const gg_grad_view src = fused ? g->alt_view() : g->grad_view();  // the buffer this iteration's gradients go to
    int pushed = 0;
    if (fused && I.inkernel) {
      // ONE launch for all tiles: the boundary tiles (first in the grid) push their send rows to the
      // partners straight from their registers, the last of them raises the flags; the partners'
      // rows arrive while the interior tiles run.  (Both exchange schedules map to this one: a
      // fork/join between two streams costs 8-18 us per iteration inside a hipGraph.)
      gg_push_args pa;
      pa.tile_off = I.d_tile_off; pa.ent = I.d_ent; pa.ent_row = I.d_ent_row; pa.dst = I.d_dst[par];
      pa.hdr = g->ipc_hdr(); pa.rflag = I.d_rflag; pa.done = I.d_done;
      pa.need = I.per_partner ? I.d_need : nullptr; pa.tile_mask = I.per_partner ? I.d_tile_mask : nullptr;
      pa.pt_first = I.d_pt_first; pa.pt_stride = I.pt_stride; pa.tile_xoff = I.d_tile_xoff;
      pa.nbtiles = g->nbtiles; pa.nslots = nslots;
      pa.inv_after_flag = I.mode == 2 ? 1 : 0;
      pa.wait_polls = I.wait_pending && I.wait_inkernel && !I.fault_skip_wait ? (long)ipc_max_polls() : 0;
      const int rc = launch_fused(g, CFDP_TILES_ALL, g->s_main, &pa);
      if (rc == 1) return 1;
      pushed = rc == 0;
      if (pushed) I.wait_pending = false;  // absorbed (or there was none)
      else if (ipc_settle(g)) return 1;    // no fused kernel fits these tiles: the separate kernels below
    }
    if (pushed) {
    } else if (overlap) {
      // boundary tiles -> push -> notify on the comm stream, the interior tiles on the main stream at
      // the same time (see cfdp_gpu_step_pre); the wait joins them
      if (fork_comm(g)) return 1;  // the comm stream forks off the main stream here
      if (grad_tiles(CFDP_TILES_BOUNDARY, g->s_comm)) return 1;
      HIP_TRY(gg_launch_push(g->d_sendidx, g->send_off.back(), I.d_slot_of_row, I.d_send_off, src, I.d_dst[par], g->s_comm));
      HIP_TRY(gg_launch_notify(g->ipc_hdr(), I.d_rflag, nslots, g->s_comm));
      HIP_TRY(hipEventRecord(g->ev_senddone, g->s_comm));
      if (grad_tiles(CFDP_TILES_INTERIOR, g->s_main)) return 1;
      HIP_TRY(hipStreamWaitEvent(g->s_main, g->ev_senddone, 0));
    } else {
      if (grad_tiles(CFDP_TILES_ALL, g->s_main)) return 1;
      HIP_TRY(gg_launch_push(g->d_sendidx, g->send_off.back(), I.d_slot_of_row, I.d_send_off, src, I.d_dst[par], g->s_main));
      HIP_TRY(gg_launch_notify(g->ipc_hdr(), I.d_rflag, nslots, g->s_main));
    }
    if (fused) fused_done(g);
    I.xiter++;  // from here on the ghost block is the arena this exchange lands in
    g->main_marked = false;
    // the wait for this exchange: left to the boundary tiles of the next pushing pass (one launch per
    // iteration), or -- ipc_settle -- to a wait kernel in front of whatever else reads the ghost rows first
    I.wait_pending = true;
    if (!(pushed && I.wait_inkernel) && ipc_settle(g)) return 1;
  }
  return 0;
}

int ipc_post(cfdp_gpu *g, int with_flux, int flux_mode);

int one_step_ipc(cfdp_gpu *g, int with_exchange, int overlap, int with_flux, int flux_mode) {
  if (ipc_pre(g, with_exchange, overlap)) return 1;
  return ipc_post(g, with_flux, flux_mode);
}

int ipc_post(cfdp_gpu *g, int with_flux, int flux_mode) {
  if (with_flux) {
    if (flux_mode != CFDP_FLUX_CONSISTENT && flux_mode != CFDP_FLUX_REFERENCE) return fail("bad flux mode %d", flux_mode);
    if (g->fusion && g->d_grad_alt) g->flux_pending = flux_mode;
    else if (launch_flux(g, flux_mode, g->s_main)) return 1;
  }
  if (scaled_tail(g, scaled_lag(g, with_flux), true, g->s_main)) return 1;
  // no end-of-iteration marker here (each costs ~5 us on the device): a later step that needs to
  // fork its comm stream records one itself (fork_comm), and the single-stream schedules need none
  g->main_marked = false;
  return 0;
}
}  // namespace

// CFDP_IPC_MODE = coarse | split | fine (CFDP_IPC_FINEGRAINED=1 is the older spelling of fine)
static int ipc_mode_from_env() {
  if (const char *m = getenv("CFDP_IPC_MODE")) {
    if (!strcmp(m, "fine")) return 1;
    if (!strcmp(m, "split")) return 2;
    if (!strcmp(m, "coarse")) return 0;
  }
  const char *fg = getenv("CFDP_IPC_FINEGRAINED");
  return fg && atoi(fg) != 0 ? 1 : 0;
}

// a partner's block (or flag block), mapped once per handle
static int ipc_open(cfdp_gpu *g, const void *handle64, unsigned char **base_out) {
  auto &I = g->ipc;
  void *base = nullptr;
  for (size_t i = 0; i < I.opened.size(); i++)
    if (!memcmp(I.opened_handle[i].data(), handle64, 64)) base = I.opened[i];
  if (!base) {
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof h);
    HIP_TRY(hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess));
    I.opened.push_back(base);
    I.opened_handle.emplace_back((const unsigned char *)handle64, (const unsigned char *)handle64 + 64);
  }
  *base_out = static_cast<unsigned char *>(base);
  return 0;
}

int cfdp_gpu_ipc_export(cfdp_gpu *g, void *handle64, size_t *land_bytes) {
  NEED_UPLOAD(g);
  if (!handle64) return fail("null argument");
  if ((int)g->partner.size() > GG_IPC_MAXSLOTS) return fail("more than %d partners", GG_IPC_MAXSLOTS);
  for (size_t s = 0; s < g->partner.size(); s++)  // the double-buffered arenas rely on traffic in both directions
    if (g->send_off[s + 1] == g->send_off[s] || g->recv_off[s + 1] == g->recv_off[s])
      return fail("partner %d is not a two-way partner", g->partner[s]);
  // the two landing arenas are safe without credit messages because only tiles that hold send
  // points read ghost rows, and those tiles are done before this rank's next push is announced
  if (g->interior_reads_ghosts)
    return fail("a tile without send points reads ghost rows (one-way halo): not supported by this exchange");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "HIP IPC handles are 64 bytes");
  ipc_release(g);
  auto &I = g->ipc;
  I.land_bytes = (((size_t)(g->nall - g->nown) * 21 * sizeof(double)) + 255) & ~(size_t)255;
  const size_t bytes = GG_IPC_HDR_BYTES + 2 * I.land_bytes;
  I.mode = ipc_mode_from_env();
  if (I.mode == 1) HIP_TRY(hipExtMallocWithFlags((void **)&I.block, bytes, hipDeviceMallocFinegrained));
  else HIP_TRY(hipMalloc(&I.block, bytes));
  HIP_TRY(hipMemset(I.block, 0, bytes));
  if (I.mode == 2) {  // the flag words alone in fine-grained memory (the header of `block` stays unused)
    HIP_TRY(hipExtMallocWithFlags((void **)&I.flags, 64 * 1024, hipDeviceMallocFinegrained));
    HIP_TRY(hipMemset(I.flags, 0, 64 * 1024));
  }
  hipIpcMemHandle_t h;
  HIP_TRY(hipIpcGetMemHandle(&h, I.block));
  memcpy(handle64, &h, sizeof h);
  memcpy(I.my_handle, &h, sizeof h);
  if (land_bytes) *land_bytes = I.land_bytes;
  const int nslots = (int)g->partner.size();
  I.dst[0].assign(nslots, nullptr); I.dst[1].assign(nslots, nullptr); I.rflag.assign(nslots, nullptr);
  return 0;
}

int cfdp_gpu_ipc_connect(cfdp_gpu *g, int slot, const void *partner_handle64, size_t land_off0,
                         size_t land_off1, size_t flag_off) {
  NEED_UPLOAD(g);
  auto &I = g->ipc;
  if (!I.block) return fail("cfdp_gpu_ipc_export() first");
  if (slot < 0 || slot >= (int)g->partner.size() || !partner_handle64) return fail("bad partner slot");
  unsigned char *b = nullptr;
  if (ipc_open(g, partner_handle64, &b)) return 1;
  I.dst[0][slot] = reinterpret_cast<double *>(b + land_off0);
  I.dst[1][slot] = reinterpret_cast<double *>(b + land_off1);
  I.rflag[slot] = reinterpret_cast<int *>(b + flag_off);
  return 0;
}

// the handle of the block that holds this rank's flag words: a block of its own in split mode, else the main block
int cfdp_gpu_ipc_export_flags(cfdp_gpu *g, void *handle64) {
  NEED_UPLOAD(g);
  auto &I = g->ipc;
  if (!I.block || !handle64) return fail("cfdp_gpu_ipc_export() first");
  if (I.flags) {
    hipIpcMemHandle_t h;
    HIP_TRY(hipIpcGetMemHandle(&h, I.flags));
    memcpy(handle64, &h, sizeof h);
  } else {
    memcpy(handle64, I.my_handle, 64);
  }
  return 0;
}

// my arrival counter at partner `slot` lives at flag_off in the block of THAT handle (after cfdp_gpu_ipc_connect)
int cfdp_gpu_ipc_connect_flags(cfdp_gpu *g, int slot, const void *partner_flags_handle64, size_t flag_off) {
  NEED_UPLOAD(g);
  auto &I = g->ipc;
  if (!I.block) return fail("cfdp_gpu_ipc_export() first");
  if (slot < 0 || slot >= (int)g->partner.size() || !partner_flags_handle64) return fail("bad partner slot");
  unsigned char *b = nullptr;
  if (ipc_open(g, partner_flags_handle64, &b)) return 1;
  I.rflag[slot] = reinterpret_cast<int *>(b + flag_off);
  return 0;
}

// MEASUREMENT ONLY (tools/loopback_probe.py): partner slot `slot` is this rank ITSELF -- its rows land in its own arenas
// at the slot's receive offset, its flag is its own flag word.  The ghost rows then hold this rank's own send rows (wrong
// values, right traffic): what one iteration of the write + notify protocol costs when the partner is never late.
int cfdp_gpu_ipc_connect_loopback(cfdp_gpu *g, int slot) {
  NEED_UPLOAD(g);
  auto &I = g->ipc;
  if (!I.block) return fail("cfdp_gpu_ipc_export() first");
  if (slot < 0 || slot >= (int)g->partner.size()) return fail("bad partner slot");
  if (g->send_off[slot + 1] - g->send_off[slot] > g->recv_off[slot + 1] - g->recv_off[slot])
    return fail("loopback needs a receive slice at least as long as the send slice (slot %d)", slot);
  const size_t base = GG_IPC_HDR_BYTES + (size_t)g->recv_off[slot] * 21 * sizeof(double);
  I.dst[0][slot] = reinterpret_cast<double *>(I.block + base);
  I.dst[1][slot] = reinterpret_cast<double *>(I.block + base + I.land_bytes);
  I.rflag[slot] = g->ipc_hdr() + slot;
  return 0;
}

// what the exchange set up by cfdp_gpu_ipc_ready does: bit 0 the fused pass pushes and notifies itself, bit 1 its
// boundary tiles wait themselves, bit 2 per-partner notification and wait masks, bits 4-5 the memory mode
// (0 coarse, 1 fine, 2 split)
int cfdp_gpu_ipc_mode(const cfdp_gpu *g) {
  if (!g || !g->ipc.block) return -1;
  const auto &I = g->ipc;
  return (I.inkernel ? 1 : 0) | (I.inkernel && I.wait_inkernel ? 2 : 0) | (I.per_partner ? 4 : 0) | (I.mode << 4);
}

int cfdp_gpu_ipc_ready(cfdp_gpu *g) {
  NEED_UPLOAD(g);
  auto &I = g->ipc;
  if (!I.block) return fail("cfdp_gpu_ipc_export() first");
  const int nslots = (int)g->partner.size();
  for (int s = 0; s < nslots; s++)
    if (!I.dst[0][s] || !I.dst[1][s] || !I.rflag[s]) return fail("partner slot %d is not connected", s);
  const size_t nsend = (size_t)g->send_off.back();
  std::vector<int> slot_of_row(nsend ? nsend : 1, 0);
  for (int s = 0; s < nslots; s++)
    for (int j = g->send_off[s]; j < g->send_off[s + 1]; j++) slot_of_row[j] = s;
  for (int par = 0; par < 2; par++) {
    HIP_TRY(hipMalloc(&I.d_dst[par], sizeof(double *) * (size_t)(nslots ? nslots : 1)));
    if (nslots) HIP_TRY(hipMemcpy(I.d_dst[par], I.dst[par].data(), sizeof(double *) * (size_t)nslots, hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMalloc(&I.d_rflag, sizeof(int *) * (size_t)(nslots ? nslots : 1)));
  if (nslots) HIP_TRY(hipMemcpy(I.d_rflag, I.rflag.data(), sizeof(int *) * (size_t)nslots, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&I.d_slot_of_row, sizeof(int) * slot_of_row.size()));
  HIP_TRY(hipMemcpy(I.d_slot_of_row, slot_of_row.data(), sizeof(int) * slot_of_row.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&I.d_send_off, sizeof(int) * g->send_off.size()));
  HIP_TRY(hipMemcpy(I.d_send_off, g->send_off.data(), sizeof(int) * g->send_off.size(), hipMemcpyHostToDevice));
  {  // the send rows of every boundary tile: (tile-local point | slot << 16, row in the partner's slice)
    std::vector<int> tile_off((size_t)g->ntiles + 1, 0), ent(nsend ? nsend : 1, 0), ent_row(nsend ? nsend : 1, 0), tile_of(nsend ? nsend : 1, 0);
    bool ok = nslots <= 0x7FFF;
    for (size_t j = 0; j < nsend && ok; j++) {
      const int p = g->send_idx_host[j];
      int lo = 0, hi = g->nbtiles - 1, t = -1;  // boundary tiles hold the send points, sorted by pstart
      while (lo <= hi) {
        const int mid = (lo + hi) / 2;
        if (p < g->h_tiles[mid].pstart) hi = mid - 1;
        else if (p >= g->h_tiles[mid].pstart + g->h_tiles[mid].npts) lo = mid + 1;
        else { t = mid; break; }
      }
      if (t < 0 || p - g->h_tiles[t].pstart > 0xFFFF) { ok = false; break; }
      tile_of[j] = t;
      tile_off[t + 1]++;
    }
    I.inkernel = false;
    if (ok && nsend) {
      for (int t = 0; t < g->ntiles; t++) tile_off[t + 1] += tile_off[t];
      // a point's FIRST destination (message order) goes into the point-major table and to the front of its tile's
      // entries; further destinations (points on an edge or corner between partners) behind them, from tile_xoff on
      // (stride = the lanes-per-point groups of the LARGEST workgroup a launch over all tiles can have: every thread of
      // a boundary tile's workgroup reads its group's entry, also the groups beyond the tile's points)
      int tpmax = 1;
      for (int t = 0; t < g->ntiles; t++) tpmax = g->h_tiles[t].npts > tpmax ? g->h_tiles[t].npts : tpmax;
      I.pt_stride = (tpmax + 63) & ~63;
      std::vector<int2> pt_first((size_t)(g->nbtiles ? g->nbtiles : 1) * I.pt_stride, make_int2(-1, 0));
      std::vector<int> tile_xoff((size_t)g->ntiles + 1, 0), nfirst((size_t)g->ntiles, 0);
      std::vector<char> is_first(nsend, 0);
      for (size_t j = 0; j < nsend; j++) {
        const int t = tile_of[j], li = g->send_idx_host[j] - g->h_tiles[t].pstart;
        int2 &f = pt_first[(size_t)t * I.pt_stride + li];
        if (f.x < 0) { f = make_int2(slot_of_row[j], (int)j - g->send_off[slot_of_row[j]]); is_first[j] = 1; nfirst[t]++; }
      }
      std::vector<int> fill(tile_off.begin(), tile_off.end() - 1), fillx((size_t)g->ntiles, 0);
      for (int t = 0; t < g->ntiles; t++) { tile_xoff[t] = tile_off[t] + nfirst[t]; fillx[t] = tile_xoff[t]; }
      tile_xoff[g->ntiles] = tile_off[g->ntiles];
      for (size_t j = 0; j < nsend; j++) {
        const int t = tile_of[j], s = slot_of_row[j], at = is_first[j] ? fill[t]++ : fillx[t]++;
        ent[at] = (g->send_idx_host[j] - g->h_tiles[t].pstart) | (s << 16);
        ent_row[at] = (int)j - g->send_off[s];
      }
      HIP_TRY(hipMalloc(&I.d_pt_first, sizeof(int2) * pt_first.size()));
      HIP_TRY(hipMemcpy(I.d_pt_first, pt_first.data(), sizeof(int2) * pt_first.size(), hipMemcpyHostToDevice));
      HIP_TRY(hipMalloc(&I.d_tile_xoff, sizeof(int) * tile_xoff.size()));
      HIP_TRY(hipMemcpy(I.d_tile_xoff, tile_xoff.data(), sizeof(int) * tile_xoff.size(), hipMemcpyHostToDevice));
      HIP_TRY(hipMalloc(&I.d_tile_off, sizeof(int) * tile_off.size()));
      HIP_TRY(hipMemcpy(I.d_tile_off, tile_off.data(), sizeof(int) * tile_off.size(), hipMemcpyHostToDevice));
      HIP_TRY(hipMalloc(&I.d_ent, sizeof(int) * ent.size()));
      HIP_TRY(hipMemcpy(I.d_ent, ent.data(), sizeof(int) * ent.size(), hipMemcpyHostToDevice));
      HIP_TRY(hipMalloc(&I.d_ent_row, sizeof(int) * ent_row.size()));
      HIP_TRY(hipMemcpy(I.d_ent_row, ent_row.data(), sizeof(int) * ent_row.size(), hipMemcpyHostToDevice));
      const char *e = getenv("CFDP_IPC_INKERNEL");
      // the tiles push what they have just computed; a send point WITHOUT faces is computed by nobody
      // and its stored row must travel (as pack / the push kernel send it, and the reference's
      // exchange_dbl_copy_in, src/threads.c:791-813): such partitions keep the separate push kernel
      I.inkernel = g->nbtiles > 0 && !g->faceless_send && !(e && atoi(e) == 0);
      const char *w = getenv("CFDP_IPC_WAIT_INKERNEL");  // 0: always a separate wait kernel (A/B timing)
      I.wait_inkernel = !(w && atoi(w) == 0);
      // per-partner notification needs: every boundary tile reads ghost rows only of partners it sends to (then the
      // flags a tile waits for also cover the rows it is about to overwrite at those partners, see gg_kernels.hip)
      std::vector<unsigned long long> smask((size_t)g->nbtiles, 0ull);
      std::vector<int> need((size_t)(nslots ? nslots : 1), 0);
      for (int t = 0; t < g->nbtiles; t++)
        for (int e2 = tile_off[t]; e2 < tile_off[t + 1]; e2++) smask[(size_t)t] |= 1ull << (ent[e2] >> 16);
      bool pp = nslots <= GG_IPC_MAXSLOTS && (int)g->tile_recv_mask.size() == g->nbtiles;
      for (int t = 0; t < g->nbtiles && pp; t++) pp = (g->tile_recv_mask[(size_t)t] & ~smask[(size_t)t]) == 0;
      for (int t = 0; t < g->nbtiles; t++)
        for (int s2 = 0; s2 < nslots; s2++)
          if ((smask[(size_t)t] >> s2) & 1ull) need[(size_t)s2]++;
      for (int s2 = 0; s2 < nslots; s2++) pp = pp && need[(size_t)s2] > 0;
      const char *ppe = getenv("CFDP_IPC_PER_PARTNER");  // 0: one counter, all flags raised by the last boundary tile (A/B)
      I.per_partner = pp && !(ppe && atoi(ppe) == 0);
      HIP_TRY(hipMalloc(&I.d_need, sizeof(int) * need.size()));
      HIP_TRY(hipMemcpy(I.d_need, need.data(), sizeof(int) * need.size(), hipMemcpyHostToDevice));
      HIP_TRY(hipMalloc(&I.d_tile_mask, sizeof(unsigned long long) * (smask.size() + 1)));
      if (!smask.empty())
        HIP_TRY(hipMemcpy(I.d_tile_mask, smask.data(), sizeof(unsigned long long) * smask.size(), hipMemcpyHostToDevice));
      const char *f = getenv("CFDP_IPC_FAULT");
      I.fault_skip_wait = f && !strcmp(f, "skip_wait");
      if (I.fault_skip_wait) fprintf(stderr, "[cfdp] FAULT INJECTION: boundary tiles do not wait for the previous exchange (CFDP_IPC_FAULT)\n");
    }
  }
  HIP_TRY(hipMalloc(&I.d_done, sizeof(int) * (GG_IPC_MAXSLOTS + 1) * GG_DONE_STRIDE));
  HIP_TRY(hipMemset(I.d_done, 0, sizeof(int) * (GG_IPC_MAXSLOTS + 1) * GG_DONE_STRIDE));
  if (flush_flux(g)) return 1;
  HIP_TRY(hipDeviceSynchronize());
  // the ghost rows move into the landing arenas
  if (g->nall > g->nown)
    for (int par = 0; par < 2; par++)
      HIP_TRY(hipMemcpy(g->land(par), g->d_grad + (size_t)g->nown * 10, sizeof(double) * 21 * (size_t)(g->nall - g->nown),
                        hipMemcpyDeviceToDevice));
  I.xiter = 0;
  I.on = true;
  g->drop_graphs();
  return 0;
}

// switch between the landing arenas and the ghost block of grad (e.g. to time another transport
// on the same context); the mappings stay
int cfdp_gpu_ipc_enable(cfdp_gpu *g, int on) {
  NEED_UPLOAD(g);
  auto &I = g->ipc;
  if (on && (!I.block || !I.d_rflag)) return fail("cfdp_gpu_ipc_ready() has not been called");
  if (flush_flux(g)) return 1;
  HIP_TRY(hipDeviceSynchronize());
  I.on = on != 0;
  g->drop_graphs();
  return 0;
}

// how long a device-side wait polls before it gives up (process-wide; also CFDP_IPC_WAIT_SECONDS).
// Graphs captured earlier keep the bound they were captured with.
int cfdp_ipc_set_wait_seconds(double seconds) {
  if (!(seconds > 0.0)) return fail("the wait bound must be positive");
  g_ipc_wait_seconds = seconds;
  return 0;
}

int cfdp_gpu_ipc_disconnect(cfdp_gpu *g) {
  if (!g) return fail("null context");
  HIP_TRY(hipSetDevice(g->device));
  HIP_TRY(hipDeviceSynchronize());
  ipc_release(g);
  return 0;
}

// 1 if a wait for a partner's rows gave up (the partner is gone or far behind), else 0; -1 on error
int cfdp_gpu_ipc_error(cfdp_gpu *g) {
  if (!g || !g->ipc.block) return 0;
  if (hipSetDevice(g->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -1;
  int h[64];
  if (hipMemcpy(h, g->ipc_hdr(), sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  const int *e = h + GG_IPC_ERR;
  if (getenv("CFDP_DEBUG_TRACE"))
    fprintf(stderr, "[cfdp] ipc state: host xiter %ld, device iteration counter %d, arrival counter of slot 0: %d; "
                    "%d waits gave up (last: slot %d, waiting for %d, saw %d)\n",
            g->ipc.xiter, h[GG_IPC_ITER], h[0], e[4], e[1], e[2], e[3]);
  return e[0] != 0;
}

// for hosts that call the two face loops separately: the part of an iteration before the flux
// (gradients, push, notify, wait) and the flux part
int cfdp_gpu_step_ipc_post(cfdp_gpu *g, int with_flux, int flux_mode) {
  NEED_UPLOAD(g);
  if (!g->ipc.on) return fail("cfdp_gpu_ipc_ready() has not been called");
  return ipc_post(g, with_flux, flux_mode);
}

int cfdp_gpu_step_ipc_pre(cfdp_gpu *g, int with_exchange, int overlap) {
  NEED_UPLOAD(g);
  if (!g->ipc.on) return fail("cfdp_gpu_ipc_ready() has not been called");
  return ipc_pre(g, with_exchange, overlap);
}

int cfdp_gpu_step_ipc(cfdp_gpu *g, int with_exchange, int overlap, int with_flux, int flux_mode) {
  NEED_UPLOAD(g);
  if (!g->ipc.on) return fail("cfdp_gpu_ipc_ready() has not been called");
  return one_step_ipc(g, with_exchange, overlap, with_flux, flux_mode);
}

// `steps` iterations.  use_graph: after two lead-in steps (the first iteration of a run has no flux to
// fuse with), the steps are replayed from hipGraphs -- both streams, the push / notify / wait kernels
// included: whole chunks of 50 from one graph, the even part of the remainder from a second one (an
// even count restores the parity of the landing arenas and of the two grad buffers, which the kernels'
// arguments bake in), at most one step launched from the streams -- so short runs replay as well.
int cfdp_gpu_run_steps_ipc(cfdp_gpu *g, int steps, int with_exchange, int overlap, int with_flux,
                           int flux_mode, int use_graph) {
  NEED_UPLOAD(g);
  if (!g->ipc.on) return fail("cfdp_gpu_ipc_ready() has not been called");
  if (steps < 1) return fail("steps must be >= 1");
  auto &I = g->ipc;
  const int full = 50;  // steps per replay of the main graph (2 kernels each with the in-kernel push)
  int done = 0;
  if (use_graph && steps >= 4) {
    for (; done < 2; done++)
      if (one_step_ipc(g, with_exchange, overlap, with_flux, flux_mode)) return 1;
    if (I.xiter & 1) {  // a graph is tied to the arena parity it was captured at: even
      if (one_step_ipc(g, with_exchange, overlap, with_flux, flux_mode)) return 1;
      done++;
    }
    // (the arena the ghost rows are read from is baked into the kernels' arguments too)
    if (I.g_exch != with_exchange || I.g_overlap != overlap || I.g_flux != with_flux || I.g_mode != flux_mode ||
        I.g_cur != g->d_grad || I.g_xpar != (int)(I.xiter & 1)) {
      if (I.graph) { (void)hipGraphExecDestroy(I.graph); I.graph = nullptr; }
      if (I.graph_rem) { (void)hipGraphExecDestroy(I.graph_rem); I.graph_rem = nullptr; }
      I.graph_n = I.graph_rem_n = 0;
    }
    auto capture = [&](hipGraphExec_t &slot, int &slot_n, int n) -> bool {  // false: run from the streams instead
      if (slot && slot_n == n) return true;
      if (slot) { (void)hipGraphExecDestroy(slot); slot = nullptr; }
      slot_n = 0;
      const double *cur0 = g->d_grad;
      const int pend0 = g->flux_pending;
      const long iter0 = g->iter, x0 = I.xiter;
      const unsigned passes0 = g->fused_passes;
      const bool wait0 = I.wait_pending;  // a chunk starts and ends with the wait of its last exchange pending
      hipGraph_t gr = nullptr;
      if (hipStreamBeginCapture(g->s_main, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return false; }
      g->main_marked = false;  // the first captured step must fork off a record made INSIDE the capture
      int rc = 0;
      for (int i = 0; i < n && !rc; i++) rc = one_step_ipc(g, with_exchange, overlap, with_flux, flux_mode);
      hipError_t ec = hipStreamEndCapture(g->s_main, &gr);
      (void)hipEventRecord(g->ev_fork, g->s_main);      // events last recorded inside a capture may not
      (void)hipEventRecord(g->ev_senddone, g->s_comm);  // be waited for outside it: re-arm them
      (void)mark_main(g);
      const bool ok = !rc && ec == hipSuccess && gr && g->d_grad == cur0 && g->flux_pending == pend0 && I.wait_pending == wait0;
      if (ok && hipGraphInstantiate(&slot, gr, nullptr, nullptr, 0) != hipSuccess) slot = nullptr;
      if (gr) (void)hipGraphDestroy(gr);
      g->iter = iter0;
      I.xiter = x0;  // nothing of the capture has run
      g->fused_passes = passes0;
      I.wait_pending = wait0;
      if (!ok || !slot) {
        if (g->d_grad != cur0) { std::swap(g->d_grad, g->d_grad_alt); std::swap(g->own_grad, g->own_grad_alt); }
        g->flux_pending = pend0;
        (void)hipGetLastError();
        return false;
      }
      slot_n = n;
      I.g_exch = with_exchange; I.g_overlap = overlap; I.g_flux = with_flux; I.g_mode = flux_mode; I.g_cur = g->d_grad;
      I.g_xpar = (int)(I.xiter & 1);
      return true;
    };
    auto replay = [&](hipGraphExec_t ge, int n) -> int {
      HIP_TRY(hipGraphLaunch(ge, g->s_main));
      // (a replay does not touch the event OBJECTS recorded inside the capture: whatever is ordered
      // after "the previous iteration" later needs a fresh record -- fork_comm makes one)
      g->main_marked = false;
      g->iter += n;
      if (with_exchange && !g->partner.empty()) I.xiter += n;
      done += n;
      return 0;
    };
    if (steps - done >= full && capture(I.graph, I.graph_n, full))
      while (steps - done >= full)
        if (replay(I.graph, full)) return 1;
    const int rem = (steps - done) & ~1;
    if (rem >= 2 && rem < full && capture(I.graph_rem, I.graph_rem_n, rem) && replay(I.graph_rem, rem)) return 1;
  }
  for (; done < steps; done++)
    if (one_step_ipc(g, with_exchange, overlap, with_flux, flux_mode)) return 1;
  return 0;
}

// measurement: the schedule of an exchange step WITHOUT the exchange (brackets only: boundary tiles
// + pack beside / before the interior tiles, fork and join), `steps` of them replayed from one
// hipGraph or launched from the streams; average milliseconds per step.  What does the two-stream
// schedule itself cost?
int cfdp_gpu_time_schedule(cfdp_gpu *g, int steps, int with_exchange, int overlap, int use_graph, float *ms_step) {
  NEED_UPLOAD(g);
  if (steps < 2) return fail("steps must be >= 2");
  steps += steps & 1;
  if (flush_flux(g)) return 1;
  auto one = [&]() -> int {
    if (cfdp_gpu_step_pre(g, with_exchange, overlap)) return 1;
    return cfdp_gpu_step_post(g, 1, CFDP_FLUX_CONSISTENT);
  };
  for (int i = 0; i < 4; i++)
    if (one()) return 1;
  hipGraphExec_t ge = nullptr;
  if (use_graph) {
    hipGraph_t gr = nullptr;
    HIP_TRY(hipStreamBeginCapture(g->s_main, hipStreamCaptureModeThreadLocal));
    g->main_marked = false;
    int rc = 0;
    for (int i = 0; i < steps && !rc; i++) rc = one();
    hipError_t ec = hipStreamEndCapture(g->s_main, &gr);
    (void)hipEventRecord(g->ev_fork, g->s_main);
    (void)hipEventRecord(g->ev_pack, g->s_main);
    (void)hipEventRecord(g->ev_senddone, g->s_comm);
    (void)mark_main(g);
    if (rc) { if (gr) (void)hipGraphDestroy(gr); return 1; }
    HIP_TRY(ec);
    HIP_TRY(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
    HIP_TRY(hipGraphDestroy(gr));
    g->main_marked = false;
    HIP_TRY(hipGraphLaunch(ge, g->s_main));
  }
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipEventRecord(g->ev_a, g->s_main));
  if (ge) {
    g->main_marked = false;
    HIP_TRY(hipGraphLaunch(ge, g->s_main));
  } else {
    for (int i = 0; i < steps; i++)
      if (one()) return 1;
  }
  HIP_TRY(hipEventRecord(g->ev_b, g->s_main));
  HIP_TRY(hipEventSynchronize(g->ev_b));
  HIP_TRY(hipDeviceSynchronize());
  if (ge) HIP_TRY(hipGraphExecDestroy(ge));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, g->ev_a, g->ev_b));
  if (ms_step) *ms_step = ms / (float)steps;
  return flush_flux(g);
}

// ------------------------------------------------------------------------ multigrid V cycle
// The published experiment is a "3V multigrid cycle" (documentation/CFD-Proxy.pdf p.3): `sweeps`
// iterations on every level going down, and again going up; the reference has no transfer
// operators -- a level is just another set of dualgrid files (src/hybrid.f6.c:38-47, -lvl).  The
// coarse levels are a few tiles each and pure launch latency, so one whole cycle over all levels
// (one partition per level, all on one device) is captured in ONE hipGraph and replayed.
int cfdp_gpu_vcycle(cfdp_gpu **levels, int nlevels, int sweeps, int cycles, int flux_mode,
                    int use_graph, float *ms_per_cycle) {
  if (!levels || nlevels < 1 || sweeps < 1 || cycles < 1) return fail("bad V-cycle arguments");
  if (flux_mode != CFDP_FLUX_CONSISTENT && flux_mode != CFDP_FLUX_REFERENCE) return fail("bad flux mode %d", flux_mode);
  for (int l = 0; l < nlevels; l++) {
    NEED_UPLOAD(levels[l]);
    if (levels[l]->device != levels[0]->device) return fail("all levels of a V cycle must live on one device");
    if (flush_flux(levels[l])) return 1;
  }
  HIP_TRY(hipDeviceSynchronize());
  cfdp_gpu *g0 = levels[0];
  hipStream_t st = g0->s_main;
  auto one_cycle = [&]() -> int {
    for (int l = 0; l < nlevels; l++)
      if (enqueue_iterations(levels[l], sweeps, 1, flux_mode, st)) return 1;
    for (int l = nlevels - 2; l >= 0; l--)
      if (enqueue_iterations(levels[l], sweeps, 1, flux_mode, st)) return 1;
    return 0;
  };
  hipGraphExec_t ge = nullptr;
  if (use_graph) {
    std::vector<const double *> cur0;
    for (int l = 0; l < nlevels; l++) cur0.push_back(levels[l]->d_grad);
    hipGraph_t gr = nullptr;
    HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = one_cycle();
    hipError_t ec = hipStreamEndCapture(st, &gr);
    if (rc) { if (gr) (void)hipGraphDestroy(gr); return 1; }
    HIP_TRY(ec);
    bool in_place = true;
    for (int l = 0; l < nlevels; l++) in_place = in_place && levels[l]->d_grad == cur0[l];
    if (in_place) HIP_TRY(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
    // else (even `sweeps` with fusion: a cycle swaps the coarsest level's buffers): the capture
    // only served as cycle number one's dry run; launch from the stream instead
    HIP_TRY(hipGraphDestroy(gr));
    if (!in_place)
      for (int l = 0; l < nlevels; l++)
        if (levels[l]->d_grad != cur0[l]) {
          std::swap(levels[l]->d_grad, levels[l]->d_grad_alt);
          std::swap(levels[l]->own_grad, levels[l]->own_grad_alt);
        }
  }
  auto run = [&]() -> int {
    if (ge) { for (int l = 0; l < nlevels; l++) levels[l]->main_marked = false; HIP_TRY(hipGraphLaunch(ge, st)); return 0; }
    return one_cycle();
  };
  if (run()) return 1;  // warm
  HIP_TRY(hipEventRecord(g0->ev_a, st));
  for (int c = 0; c < cycles; c++)
    if (run()) return 1;
  HIP_TRY(hipEventRecord(g0->ev_b, st));
  HIP_TRY(hipEventSynchronize(g0->ev_b));
  if (ge) HIP_TRY(hipGraphExecDestroy(ge));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, g0->ev_a, g0->ev_b));
  if (ms_per_cycle) *ms_per_cycle = ms / (float)cycles;
  return 0;
}

}  // extern "C"
