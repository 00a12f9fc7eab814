// gpu_abi.hip -- implementation of the C ABI declared in include/cfdproxy_hip.h: the context, device memory, field
// transfers, kernel launches, iteration brackets, in-process rank groups, hipGraph-replayed iterations, measurement.
// The exchanges between ranks (RCCL, xGMI write + notify) and their validation live in gpu_exchange.hip; the arithmetic
// in gg_kernels.hip.  No CPU fallback: every entry point needs a HIP device and says so.
#include "gpu_ctx.h"

#include <dlfcn.h>

namespace {
thread_local char g_err[512] = "no error";
}

namespace cfdp_detail {
int fail(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return 1;
}
}  // namespace cfdp_detail

// for the other translation units of the library (plan_kernels.hip): same buffer as cfdp_gpu_last_error()
int cfdp_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return 1;
}

extern "C" {


int cfdp_gpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// the PCI bus id of a device ("0000:c1:00.0"): the same string in every process that sees the same physical device,
// whatever HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES made of its ordinal -- what ranks compare to find out whether they
// share a device
int cfdp_gpu_device_bus_id(int device, char *buf, int len) {
  if (!buf || len < 16) return fail("bus id buffer too small");
  HIP_TRY(hipDeviceGetPCIBusId(buf, len, device));
  return 0;
}

int cfdp_gpu_device(const cfdp_gpu *g) { return g ? g->device : -1; }

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
  if (const char *e = cfdp_experiment_getenv("CFDP_DEBUG_ABLATE")) gg_debug_flags = atoi(e);  // honoured only with CFDP_EXPERIMENTS=1
  if (const char *e = cfdp_experiment_getenv("CFDP_EXP_SKIP_PRE"))  // timing experiment, values WRONG (EXPERIMENTS.md D.2)
    gg_debug_flags = (gg_debug_flags & ~0x80000) | (atoi(e) ? 0x80000 : 0);
  if (const char *e = getenv("CFDP_FUSED_SPLIT")) gg_fused_split = atoi(e);
  if (const char *e = getenv("CFDP_GRAD_ALIAS")) gg_grad_alias = atoi(e);
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
  (void)hipFree(g->sc.d_state); (void)hipFree(g->sc.d_fref); (void)hipFree(g->sc.d_skip); (void)hipFree(g->sc.d_var0);
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
  {  // launch groups: the boundary tiles, then the interior groups of the plan, each with the maxima that pick its kernels
    g->groups.clear();
    const int ng = p->ngroups >= 1 && p->ngroups <= 4 && p->group_begin[0] == p->nbtiles && p->group_begin[p->ngroups] == p->ntiles
                       ? p->ngroups : 0;  // (a plan from a builder that does not know groups: one interior group)
    auto add = [&](int begin, int end) {
      cfdp_gpu::tile_group G;
      G.begin = begin; G.n = end - begin;
      for (int t = begin; t < end; t++) {
        const cfdp_tile_desc &td = p->tiles[t];
        const int rows = td.npts + td.nhalo;
        const int c = cfdp_tile_class(p->tile_points, rows, (long)td.blob_qw * 16);
        if (c > G.cls) G.cls = c;
        if (rows > G.max_rows) G.max_rows = rows;
        // lane groups the tile needs: its points and the helper groups of its long lists (cfdproxy_host.h: the helper table
        // sits behind the offsets of a blob that has one)
        int groups = td.npts;
        {
          const long plane = cfdp_blob_plane_bytes(td.nfaces), base = 3 * plane + cfdp_blob_inc_bytes(td.ninc) + cfdp_blob_off_bytes(td.npts);
          if ((long)td.blob_qw * 16 > base) {
            unsigned nh = 0;
            memcpy(&nh, p->blob + (size_t)td.blob_off * 16 + base, sizeof nh);
            groups += (int)nh;
          }
        }
        if (groups > G.tp) G.tp = groups;
        if (td.nhalo > G.max_halo) G.max_halo = td.nhalo;
        if (td.blob_qw > G.max_blob) G.max_blob = td.blob_qw;
        const size_t lg = (size_t)td.blob_qw * 16 + (size_t)rows * 64, lf = (size_t)td.blob_qw * 16 + (size_t)rows * 80;
        if (lg > G.lds_grad) G.lds_grad = lg;
        if (lf > G.lds_flux) G.lds_flux = lf;
      }
      g->groups.push_back(G);
    };
    add(0, p->nbtiles);
    if (ng) for (int k = 0; k < ng; k++) add(p->group_begin[k], p->group_begin[k + 1]);
    else add(p->nbtiles, p->ntiles);
    for (const auto &G : g->groups)
      if (G.lds_grad > 160 * 1024 || G.lds_flux > 160 * 1024)
        return fail("tile needs %zu / %zu bytes of LDS (> 160 KiB): use a smaller tile_points", G.lds_grad, G.lds_flux);
    // the kernels run tile_points x lanes threads per tile (4 / 8 lanes per point by default): refuse here, with a message,
    // what the first launch would refuse as "invalid configuration argument"
    int tpmax = 0;
    for (const auto &G : g->groups) tpmax = G.tp > tpmax ? G.tp : tpmax;
    const int lanes = g->grad_lanes > g->flux_lanes ? g->grad_lanes : g->flux_lanes;
    if ((long)tpmax * lanes > 1024)
      return fail("tiles of up to %d points x %d lanes per point = %ld threads per workgroup (> 1024): build the plan with tile_points <= %d, "
                  "or lower the lanes per point (cfdp_gpu_set_variant) before the upload", tpmax, lanes, (long)tpmax * lanes, 1024 / lanes);
  }
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
  HIP_TRY(cfdp_memset_sync(g->d_halo, 0, sizeof(int) * (size_t)(p->nhalo_total + 1)));
  if (p->nhalo_total)
    HIP_TRY(hipMemcpy(g->d_halo, p->halo_idx, sizeof(int) * (size_t)p->nhalo_total, hipMemcpyHostToDevice));
  {  // row lists at a fixed stride for the tiles of every group a fixed-capacity kernel can run (GG_ROW_STRIDE at <= 204
     // staged rows -- the small image --, GG_ROW_STRIDE_LARGE up to 256); the tiles of a generic group never read theirs
    int maxrows = 0;
    bool any = false;
    for (const auto &G : g->groups)
      if (G.n > 0 && G.cls != CFDP_TILE_GENERIC) { any = true; maxrows = G.max_rows > maxrows ? G.max_rows : maxrows; }
    bool fits = any && p->tile_points <= 64 && maxrows <= 256;
    if (const char *e = getenv("CFDP_ROWLIST")) fits = fits && atoi(e) != 0;
    g->rowlist_stride = 0;
    if (fits) {
      const int stride = maxrows <= 204 ? GG_ROW_STRIDE : GG_ROW_STRIDE_LARGE;
      std::vector<int> rl((size_t)p->ntiles * stride, 0);
      for (const auto &G : g->groups) {
        if (G.cls == CFDP_TILE_GENERIC) continue;
        for (int t = G.begin; t < G.begin + G.n; t++) {
          const cfdp_tile_desc &td = p->tiles[t];
          int *r = rl.data() + (size_t)t * stride;
          const int n = td.npts + td.nhalo;
          if (n <= 0) continue;
          for (int i = 0; i < td.npts; i++) r[i] = td.pstart + i;
          for (int i = 0; i < td.nhalo; i++) r[td.npts + i] = p->halo_idx[td.halo_off + i];
          for (int i = n; i < stride; i++) r[i] = r[n - 1];
        }
      }
      HIP_TRY(hipMalloc(&g->d_rowlist, rl.size() * sizeof(int)));
      HIP_TRY(hipMemcpy(g->d_rowlist, rl.data(), rl.size() * sizeof(int), hipMemcpyHostToDevice));
      g->rowlist_stride = stride;
    }
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
  HIP_TRY(cfdp_memset_sync(g->d_var, 0, sizeof(double) * 8 * (size_t)p->nall));
  HIP_TRY(cfdp_memset_sync(g->d_grad, 0, sizeof(double) * 21 * (size_t)p->nall));
  HIP_TRY(cfdp_memset_sync(g->d_flux, 0, sizeof(double) * 3 * (size_t)p->nown));
  g->uploaded = true;
  if (g->fusion) return cfdp_gpu_set_fusion(g, 1);
  return 0;
}


int cfdp_gpu_bind_grad(cfdp_gpu *g, void *dev_grad) {
  NEED_UPLOAD(g);
  if (!dev_grad) return fail("null device pointer");
  if ((uintptr_t)dev_grad & 15) return fail("grad buffer must be 16-byte aligned");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(cfdp_copy_d2d_sync(dev_grad, g->d_grad, sizeof(double) * 21 * (size_t)g->nall));
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


// xGMI write + notify: make the main stream wait (bounded, on the device) for the latest exchange's arrival,
// if nothing has done so yet -- before anything but a pushing fused pass touches ghost rows
extern "C++" int cfdp_detail::ipc_settle(cfdp_gpu *g) {
  if (!g->ipc.on || !g->ipc.wait_pending) return 0;
  g->ipc.wait_pending = false;
  g->main_marked = false;
  HIP_TRY(gg_launch_wait(g->ipc_hdr(), (int)g->partner.size(), ipc_max_polls(), g->ipc.counters ? g->ipc.d_tile_iter : nullptr, g->s_main));
  return 0;
}

// scaled-field validation (cfdp_gpu_scaled_check_begin): at the end of a step, on the main stream once it has joined
// the comm stream -- every reader of var of this iteration is ahead of it, the next gradient launch behind it.
// lag: see gg_validate_kernel.  scale = false: compare only (the deferred flux of a run's last iteration).
extern "C++" int cfdp_detail::scaled_tail(cfdp_gpu *g, int lag, bool scale, hipStream_t st) {
  if (!g->sc.on) return 0;
  g->main_marked = false;
  HIP_TRY(gg_launch_validate(g->d_var, g->nall, g->d_flux, g->sc.d_fref, g->sc.d_skip, g->nown, lag, scale, g->sc.d_state, st));
  return 0;
}
// the lag of the flux a step leaves in d_flux when it closes with (with_flux, fused deferral or not)
extern "C++" int cfdp_detail::scaled_lag(const cfdp_gpu *g, int with_flux) { return !with_flux ? -1 : (g->fusion && g->d_grad_alt ? 1 : 0); }

// run the deferred flux of the last fused-mode iteration, if any
extern "C++" int cfdp_detail::flush_flux(cfdp_gpu *g, bool record, hipStream_t st) {
  if (g->flux_pending < 0) return ipc_settle(g);
  // the flux that closes a run of exchanging iterations reads the ghost rows of the LAST exchange: where the passes'
  // boundary tiles wait themselves and the flux kernel of these tiles can do the same, ITS boundary tiles wait -- no wait
  // kernel (4-5 us and a kernel boundary) between the last pass and the flux; everywhere else the wait kernel first
  gg_push_args wa;
  const gg_push_args *wait = nullptr;
  {
    auto &I = g->ipc;
    const tile_range r = range_of(g, CFDP_TILES_ALL);
    if (I.on && I.wait_pending && I.inkernel && I.wait_inkernel && !I.fault_skip_wait && (!st || st == g->s_main) &&
        gg_flux_can_wait(g->flux_lanes, r.tp, r.row_halo(), r.max_blob)) {
      ipc_push_args(g, (int)(I.xiter & 1), &wa);
      wait = &wa;
      I.wait_pending = false;  // absorbed by the flux kernel's boundary tiles
    } else if (ipc_settle(g)) {
      return 1;
    }
  }
  const int mode = g->flux_pending;
  g->flux_pending = -1;
  if (launch_flux(g, mode, st ? st : g->s_main, wait)) return 1;
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
    HIP_TRY(cfdp_copy_d2d_sync(g->d_grad_alt, g->d_grad, sizeof(double) * 21 * (size_t)g->nall));
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
  HIP_TRY(cfdp_copy_d2d_sync(dev_grad, g->d_grad_alt ? g->d_grad_alt : g->d_grad, sizeof(double) * 21 * (size_t)g->nall));
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
      HIP_TRY(hipMemcpy(g->land(par), tmp + (size_t)g->nown * 6,
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
    HIP_TRY(hipMemcpy(tmp + (size_t)g->nown * 6, g->grad_view().ghost,
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

int cfdp_gpu_kernel_forms(char *buf, size_t len) { return gg_forms_take(buf, len); }

int cfdp_gpu_set_variant(cfdp_gpu *g, int grad_lanes, int flux_lanes) {
  if (!g) return fail("null context");
  auto ok = [](int l) { return l == 1 || l == 2 || l == 4 || l == 8; };
  if (grad_lanes == 0) grad_lanes = 4;
  if (flux_lanes == 0) flux_lanes = 8;
  if (!ok(grad_lanes) || !ok(flux_lanes)) return fail("lanes per point must be 1, 2, 4 or 8");
  if (g->uploaded) {  // (the same refusal as at upload: a workgroup has tile points x lanes threads)
    int tpmax = 0;
    for (const auto &G : g->groups) tpmax = G.tp > tpmax ? G.tp : tpmax;
    const int lanes = grad_lanes > flux_lanes ? grad_lanes : flux_lanes;
    if ((long)tpmax * lanes > 1024)
      return fail("tiles of up to %d points x %d lanes per point = %ld threads per workgroup (> 1024)", tpmax, lanes, (long)tpmax * lanes);
  }
  if (g->grad_lanes != grad_lanes || g->flux_lanes != flux_lanes) g->ipc.drop_graph_sets();  // the kernel forms are baked in
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
extern "C++" int cfdp_detail::mark_main(cfdp_gpu *g) {
  HIP_TRY(hipEventRecord(g->ev_fluxdone, g->s_main));
  g->main_marked = true;
  return 0;
}
// the comm stream starts after everything enqueued on the main stream so far
extern "C++" int cfdp_detail::fork_comm(cfdp_gpu *g) {
  if (g->main_marked) {
    HIP_TRY(hipStreamWaitEvent(g->s_comm, g->ev_fluxdone, 0));
  } else {
    HIP_TRY(hipEventRecord(g->ev_fork, g->s_main));
    HIP_TRY(hipStreamWaitEvent(g->s_comm, g->ev_fork, 0));
  }
  return 0;
}

// The launches that cover a tile selector.  Boundary tiles (half-size sheets) only get launches of their own when the
// schedule needs them early; in a launch over ALL tiles they ride with the first interior group, sized for the larger of
// the two (a separate launch for the few hundred boundary tiles of a rank costs ~15 us of mostly idle device per
// iteration).  The other groups of the plan -- tiles of another capacity class, kept apart because one launch for all
// would put every tile into the slower kernel form -- follow as launches of their own.
extern "C++" std::vector<tile_range> cfdp_detail::segs_of(const cfdp_gpu *g, int which) {
  auto of = [](const cfdp_gpu::tile_group &G) {
    return tile_range{G.begin, G.n, G.tp, G.max_halo, G.max_blob, G.max_rows, G.lds_grad, G.lds_flux, G.cls};
  };
  auto join = [](const tile_range &a, const tile_range &b) {
    return tile_range{a.begin, a.n + b.n, a.tp > b.tp ? a.tp : b.tp, a.max_halo > b.max_halo ? a.max_halo : b.max_halo,
                      a.max_blob > b.max_blob ? a.max_blob : b.max_blob, a.max_rows > b.max_rows ? a.max_rows : b.max_rows,
                      a.lds_grad > b.lds_grad ? a.lds_grad : b.lds_grad, a.lds_flux > b.lds_flux ? a.lds_flux : b.lds_flux,
                      a.cls > b.cls ? a.cls : b.cls};
  };
  std::vector<tile_range> out;
  if (g->groups.empty()) return out;
  const tile_range b = of(g->groups[0]);
  if (which == CFDP_TILES_BOUNDARY) { out.push_back(b); return out; }
  size_t k = 1;
  if (which == CFDP_TILES_ALL && b.n > 0) {
    // (a generic group on either side stays apart: the point of the groups is that the others keep their kernels)
    if (g->groups.size() > 1 && g->groups[1].n > 0 && b.cls != CFDP_TILE_GENERIC && g->groups[1].cls != CFDP_TILE_GENERIC) {
      out.push_back(join(b, of(g->groups[1])));
      k = 2;
    } else {
      out.push_back(b);
    }
  }
  for (; k < g->groups.size(); k++)
    if (g->groups[k].n > 0) out.push_back(of(g->groups[k]));
  if (out.empty()) out.push_back(which == CFDP_TILES_ALL ? b : of(g->groups.back()));  // (no tiles at all: an empty range)
  return out;
}
extern "C++" tile_range cfdp_detail::range_of(const cfdp_gpu *g, int which) {
  const std::vector<tile_range> s = segs_of(g, which);
  return s.empty() ? tile_range{0, 0, 0, 0, 0, 0, 0, 0, 0} : s[0];
}

extern "C++" int cfdp_detail::launch_grad(cfdp_gpu *g, int which, hipStream_t st, const gg_grad_view *into) {
  g->main_marked = false;
  gg_args a = g->args();
  if (into) a.grad = *into;
  for (const tile_range &r : segs_of(g, which))
    HIP_TRY(gg_launch_gradient(a, g->grad_lanes, r.begin, r.n, r.tp, r.lds_grad, r.row_halo(), r.max_blob, g->streaming, st));
  return 0;
}

// wait: the boundary tiles wait for the latest exchange themselves -- in the launch that holds them, the first
extern "C++" int cfdp_detail::launch_flux_tiles(cfdp_gpu *g, int mode, int which, hipStream_t st, const gg_push_args *wait) {
  g->main_marked = false;
  g->last_flux_mode = mode;
  const gg_args a = g->args();
  bool first = true;
  for (const tile_range &r : segs_of(g, which)) {
    HIP_TRY(gg_launch_flux(a, g->flux_lanes, mode == CFDP_FLUX_REFERENCE, r.begin, r.n, r.tp, r.lds_flux, r.row_halo(),
                           r.max_blob, g->streaming, st, first ? wait : nullptr));
    first = false;
  }
  return 0;
}

extern "C++" int cfdp_detail::launch_flux(cfdp_gpu *g, int mode, hipStream_t st, const gg_push_args *wait) {
  return launch_flux_tiles(g, mode, CFDP_TILES_ALL, st, wait);
}

// the deferred flux (from d_grad) + the next gradients (into d_grad_alt) over the selected tiles
// in one pass per launch group; a group no fused capacity fits runs the two separate kernels.
// The caller swaps the buffers (fused_done) once every tile range of the iteration is enqueued.
// push != nullptr: the boundary tiles push and notify themselves -- in the first launch, which holds them; returns 2
// (nothing launched) if no fused kernel fits THAT launch, so that the caller can take the separate-kernel path
extern "C++" int cfdp_detail::launch_fused(cfdp_gpu *g, int which, hipStream_t st, const gg_push_args *push) {
  g->main_marked = false;
  const gg_args a = g->args();
  const gg_grad_view gnew = g->alt_view();
  const int mode = g->flux_pending;
  g->last_flux_mode = mode;
  const std::vector<tile_range> segs = segs_of(g, which);
  if (push && (segs.empty() || segs[0].begin != 0 || !gg_fused_fits(segs[0].tp, segs[0].row_halo(), segs[0].max_blob))) return 2;
  const bool reverse = g->alternate && which == CFDP_TILES_ALL && !push && (g->fused_passes++ & 1u);
  bool first = true;
  for (const tile_range &r : segs) {
    const gg_push_args *pu = first ? push : nullptr;
    first = false;
    const hipError_t e = gg_launch_fused(a, gnew, mode == CFDP_FLUX_REFERENCE, r.begin, r.n, r.tp, r.row_halo(), r.max_blob,
                                         g->streaming, !g->beside_rccl, st, pu, reverse);
    if (e == hipErrorNotSupported && pu) return 2;
    if (e == hipErrorNotSupported) {  // this group's tiles fit no fused capacity: flux, then gradients, of these tiles only
      HIP_TRY(gg_launch_flux(a, g->flux_lanes, mode == CFDP_FLUX_REFERENCE, r.begin, r.n, r.tp, r.lds_flux, r.row_halo(),
                             r.max_blob, g->streaming, st, nullptr));
      gg_args an = a;
      an.grad = gnew;
      HIP_TRY(gg_launch_gradient(an, g->grad_lanes, r.begin, r.n, r.tp, r.lds_grad, r.row_halo(), r.max_blob, g->streaming, st));
    } else {
      HIP_TRY(e);
    }
  }
  return 0;
}

extern "C++" void cfdp_detail::fused_done(cfdp_gpu *g) {
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

// The data-movement floor of the fused pass: the same kernel with neither face loop (a diagnostic instantiation:
// every load, every store, zeros as results), timed exactly as cfdp_gpu_time_fused times the real pass.  What the
// real pass takes beyond it is arithmetic and latency its resident tiles do not hide.  Leaves grad / flux holding one
// correct iteration again.
int cfdp_gpu_time_fused_movement(cfdp_gpu *g, int iters, float *ms_pass) {
  NEED_UPLOAD(g);
  if (iters < 1) return fail("iters must be >= 1");
  if (!g->fusion || !g->d_grad_alt) return fail("fusion is off");
  if (!g->d_rowlist) return fail("no fixed-stride row lists: the movement-only instantiation needs them");
  if (const char *why = gg_diag_available()) return fail("%s", why);  // (a diagnostic instantiation: lib/libcfdproxy_diag.so)
  {  // the movement-only kernel is an instantiation of the phase-split form: refuse where that form would not run (the
     // launch would otherwise fall back to the two REAL kernels and their time be reported as the floor)
    const tile_range r = range_of(g, CFDP_TILES_ALL);
    const int block = ((r.tp * 4 + 63) / 64) * 64;
    const int cb = (r.max_blob + block - 1) / block, kv = ((r.tp + r.row_halo()) * 4 + block - 1) / block,
              kg = ((r.tp + r.row_halo()) * 3 + block - 1) / block;
    if (!gg_fused_split || g->beside_rccl || block > 1024 || cb > 6 || kv > 4 || kg > 3)
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
    hipGraph_t &tmpl = &slot == &g->graph ? g->graph_tmpl : g->graph_rem_tmpl;
    if (slot && slot_n == n && (n < 0 || slot_cur == g->d_grad)) return 0;
    if (slot) { (void)hipGraphExecDestroy(slot); slot = nullptr; }
    if (tmpl) { (void)hipGraphDestroy(tmpl); tmpl = nullptr; }
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
    if (hipGraphInstantiate(&slot, gr, nullptr, nullptr, 0) != hipSuccess) {
      slot = nullptr;
      (void)hipGraphDestroy(gr);
      return fail("hipGraphInstantiate failed: %s", hipGetErrorString(hipGetLastError()));
    }
    tmpl = gr;  // kept: cfdp_gpu_refresh_graphs instantiates from it again
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
  // (ms_total == NULL: the caller times the run itself -- no event pair around it, the call returns when the stream is through)
  if (ms_total) HIP_TRY(hipEventRecord(g->ev_a, st));
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
  if (!ms_total) {
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
  }
  HIP_TRY(hipEventRecord(g->ev_b, st));
  HIP_TRY(hipEventSynchronize(g->ev_b));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, g->ev_a, g->ev_b));
  *ms_total = ms;
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

int cfdp_gpu_refresh_graphs(cfdp_gpu *g) {
  NEED_UPLOAD(g);
  HIP_TRY(hipDeviceSynchronize());  // none of them may be replaying
  int lost = 0;
  auto again = [&](hipGraphExec_t &exec, hipGraph_t tmpl, int &n) {
    if (!exec || !tmpl) return;
    (void)hipGraphExecDestroy(exec);
    exec = nullptr;
    if (hipGraphInstantiate(&exec, tmpl, nullptr, nullptr, 0) != hipSuccess) {
      (void)hipGetLastError();
      exec = nullptr;
      n = 0;  // the next run captures it anew
      lost++;
    }
  };
  again(g->graph, g->graph_tmpl, g->graph_iters);
  again(g->graph_rem, g->graph_rem_tmpl, g->graph_rem_iters);
  for (auto &x : g->ipc.gs) {
    again(x.graph, x.tmpl, x.graph_n);
    again(x.graph_rem, x.tmpl_rem, x.graph_rem_n);
  }
  return lost ? fail("%d graph(s) could not be instantiated again (they will be captured anew)", lost) : 0;
}

// diagnostics (tools/phase_stamps.py): run `passes` fused passes with phase stamping on and return, per tile, 8
// shader-clock stamps of the LAST pass (start, indices here, loads landed, flux done, var rows in place, gradient
// arithmetic done + stores issued, stores acknowledged, unused) and, behind them, 4 per wave of the tile (own pieces
// landed, through the flux phase, through the gradient phase, unused); stamps[ntiles*24]
int cfdp_gpu_debug_phase_stamps(cfdp_gpu *g, int passes, unsigned long long *stamps) {
  NEED_UPLOAD(g);
  if (!g->fusion || !g->d_grad_alt || passes < 1 || !stamps) return fail("fusion must be on");
  if (const char *why = gg_diag_available()) return fail("%s", why);  // (a diagnostic instantiation: lib/libcfdproxy_diag.so)
  if (flush_flux(g)) return 1;
  unsigned long long *d = nullptr;
  const size_t n = (size_t)g->ntiles * 24;  // 8 per tile, then 4 x 4 per wave
  HIP_TRY(hipMalloc(&d, n * sizeof(unsigned long long)));
  HIP_TRY(cfdp_memset_sync(d, 0, n * sizeof(unsigned long long)));
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
