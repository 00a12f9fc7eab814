/*
 * dropin_mpi.c -- the MPI side of the drop-in boundary (lib/libcfdproxy_mpi.so, built when an MPI is
 * installed; libcfdproxy_hip.so itself has no MPI dependency).
 *
 * The reference is one MPI rank per process (src/comm_data.c:257-307); its main() (src/hybrid.f6.c:54-91)
 * and its harness (src/solver.c:35-314) call init_communication / compute_communication_tables /
 * init_threads / compute_gradients_gg_<variant> / compute_psd_flux / free_communication_ressources and,
 * in the harness, MPI_Barrier.  Linking this library next to libcfdproxy_hip.so gives those entry points
 * their multi-process meaning -- a constructor registers the hooks below, no symbol is overridden:
 *
 *   init_communication            MPI_Init_thread, nProc / iProc            (src/comm_data.c:257-307)
 *   compute_communication_tables  + the index exchange that builds sendindex: every rank sends each
 *                                 partner the owner-local ids of the ghosts it needs from it
 *                                                                           (src/comm_data.c:203-249)
 *   init_threads                  + the data path between the ranks' GPUs: xGMI write + notify through
 *                                 HIP IPC when all ranks share a node (VALIDATED before use: the flux of
 *                                 64 iterations in a field scaled per iteration, so that a ghost row read
 *                                 one exchange early cannot hide; ghost rows against owner rows; short
 *                                 wait bound; retried with a fine-grained block; then RCCL), else RCCL
 *                                 -- in place of src/exchange_data_{mpi,mpidma,gaspi}.c
 *   test_solver                   MPI_Barrier around every sample, rank 0 prints  (src/solver.c:42-58)
 *   free_communication_ressources unmap / tear down collectively, MPI_Finalize (when this library did
 *                                 the MPI_Init)                             (src/comm_data.c:504-521)
 *
 * MPI is the control plane only: no ghost row ever travels through it.
 */
#define CFDP_WITH_MPI 1
#include "cfdproxy_hip.h"
#include "host_util.h"

#include <math.h>
#include <string.h>

static int g_own_init = 0;   /* MPI_Init_thread was called here: finalize here too */
static int g_use_ipc = 0;    /* data path chosen by attach(): 1 = xGMI write + notify, 0 = RCCL (or none) */
static char g_path[256] = "none (one rank)";

const char *cfdp_mpi_exchange_path(void) { return g_path; }

static void hook_init(int *argc, char ***argv, comm_data *cd) {
  int inited = 0, provided = 0;
  MPI_Initialized(&inited);
  if (!inited) {
    /* every thread of the caller's omp region may call the entry points (src/solver.c:45-55); the
     * library elects one per call, so SERIALIZED is all it needs -- ask for MULTIPLE like the reference
     * (src/comm_data.c:264-272) and accept less */
    MPI_Init_thread(argc, argv, MPI_THREAD_MULTIPLE, &provided);
    CFDP_ASSERT(provided >= MPI_THREAD_SERIALIZED);
    g_own_init = 1;
  }
  MPI_Comm_size(MPI_COMM_WORLD, &cd->nProc);
  MPI_Comm_rank(MPI_COMM_WORLD, &cd->iProc);
}

/* the index exchange of src/comm_data.c:203-249: sendindex[k][j] = MY local id of the point that is
 * partner k's j-th ghost owned by me (k sends the owner-local ids of its ghosts, in its file order) */
static void hook_tables(comm_data *cd) {
  if (cd->ndomains == 1 || cd->nProc == 1) return;
  CFDP_ASSERT(cd->ndomains == cd->nProc); /* one domain file per rank, as in the reference (:94) */
  const int nc = cd->ncommdomains;
  MPI_Request *req = cfdp_malloc((size_t)(2 * nc + 1) * sizeof(MPI_Request));
  int **out = cfdp_calloc((size_t)nc + 1, sizeof(int *));
  int nreq = 0;
  for (int i = 0; i < nc; i++) {
    const int k = cd->commpartner[i];
    if (cd->sendcount[k] > 0) {
      free(cd->sendindex[k]);
      cd->sendindex[k] = cfdp_malloc((size_t)cd->sendcount[k] * sizeof(int));
      MPI_Irecv(cd->sendindex[k], cd->sendcount[k], MPI_INT, k, 4711, MPI_COMM_WORLD, &req[nreq++]);
    }
  }
  for (int i = 0; i < nc; i++) {
    const int k = cd->commpartner[i], n = cd->recvcount[k];
    if (n <= 0) continue;
    out[i] = cfdp_malloc((size_t)n * sizeof(int));
    for (int j = 0; j < n; j++) out[i][j] = cd->addpoint_id[cd->recvindex[k][j] - cd->nownpoints];
    MPI_Isend(out[i], n, MPI_INT, k, 4711, MPI_COMM_WORLD, &req[nreq++]);
  }
  MPI_Waitall(nreq, req, MPI_STATUSES_IGNORE);
  for (int i = 0; i < nc; i++) {
    const int k = cd->commpartner[i];
    for (int j = 0; j < cd->sendcount[k]; j++) CFDP_ASSERT(cd->sendindex[k][j] >= 0 && cd->sendindex[k][j] < cd->nownpoints);
    free(out[i]);
  }
  free(out);
  free(req);
}

/* ---- the xGMI write + notify path: map the partners' landing arenas (handles travel in one allgather) */
static int ipc_setup(cfdp_gpu *gpu, int r, int G) {
  enum { MAXP = 48 };
  typedef struct { unsigned char handle[64], fhandle[64]; long land; int np, partner[MAXP], recv_off[MAXP + 1]; } ipc_info;
  ipc_info mine, *all_info = cfdp_malloc((size_t)G * sizeof(ipc_info));
  memset(&mine, 0, sizeof mine);
  size_t land = 0;
  int ok = cfdp_gpu_ipc_export(gpu, mine.handle, &land) == 0 && cfdp_gpu_ipc_export_flags(gpu, mine.fhandle) == 0 &&
           cfdp_gpu_npartners(gpu) <= MAXP;
  mine.land = (long)land;
  mine.np = ok ? cfdp_gpu_npartners(gpu) : -1;
  for (int s = 0; s < mine.np; s++) {
    size_t bytes = 0;
    mine.partner[s] = cfdp_gpu_partner_rank(gpu, s);
    (void)cfdp_gpu_recv_ptr(gpu, s, &bytes);
    mine.recv_off[s + 1] = mine.recv_off[s] + (int)(bytes / (NGRAD * 3 * sizeof(double)));
  }
  MPI_Allgather(&mine, (int)sizeof mine, MPI_BYTE, all_info, (int)sizeof mine, MPI_BYTE, MPI_COMM_WORLD);
  for (int p = 0; p < G; p++) ok = ok && all_info[p].np >= 0;
  /* rank r's rows for its partner p land in p's block at header + parity*arena + recv_off_p[slot of r] rows */
  for (int s = 0; ok && s < mine.np; s++) {
    const ipc_info *pi = &all_info[mine.partner[s]];
    int t = -1;
    for (int i = 0; i < pi->np; i++)
      if (pi->partner[i] == r) t = i;
    const size_t base = CFDP_IPC_HEADER_BYTES + (size_t)pi->recv_off[t < 0 ? 0 : t] * NGRAD * 3 * sizeof(double);
    const size_t foff = cfdp_gpu_ipc_flag_offset(t < 0 ? 0 : t); /* the word of my slot there: a cache line of its own */
    ok = t >= 0 && cfdp_gpu_ipc_connect(gpu, s, pi->handle, base, base + (size_t)pi->land, foff) == 0 &&
         cfdp_gpu_ipc_connect_flags(gpu, s, pi->fhandle, foff) == 0; /* (a block of its own in "split" mode) */
  }
  ok = ok && cfdp_gpu_ipc_ready(gpu) == 0;
  int all_ok = 0;
  MPI_Allreduce(&ok, &all_ok, 1, MPI_INT, MPI_MIN, MPI_COMM_WORLD); /* also: nobody pushes before everybody is ready */
  free(all_info);
  return all_ok;
}

static void ipc_teardown(cfdp_gpu *gpu) {
  (void)cfdp_gpu_sync(gpu);
  MPI_Barrier(MPI_COMM_WORLD); /* nobody unmaps a block a partner may still write to */
  (void)cfdp_gpu_ipc_disconnect(gpu);
  MPI_Barrier(MPI_COMM_WORLD);
}

/* collective, before a data path is used.  (1) No flux phase may READ a ghost row before the rows of its exchange
 * have landed: the field is constant in time and the landing arenas alternate, so such a read returns the right value
 * and no comparison of final states sees it.  The library's scaled-field mode (cfdp_gpu_scaled_check_begin:
 * var x 2, 2, 1/4 after every iteration, the flux of EVERY step compared with reference x 2^e on the device) makes the
 * values carry the iteration number -- the analogue of the reference's stage / flag lock-step asserts at every receive
 * (src/exchange_data_mpi.c:189,439, src/exchange_data_gaspi.c:389-416).  (2) Afterwards Sigma |sent rows| (weighted by
 * the position in the message) == Sigma |ghost rows| over all ranks.  (3) No device-side wait gave up.  *why names the
 * check that failed.  Restores the host and device fields.                                                         */
static int exchange_valid(comm_data *cd, solver_data *sd, cfdp_gpu *gpu, const char **why) {
  enum { STEPS = 64 };
  const size_t nall = (size_t)sd->nallpoints;
  double *var0 = cfdp_malloc(nall * NGRAD * sizeof(double));
  double *grad0 = cfdp_malloc(nall * NGRAD * 3 * sizeof(double));
  double *flux0 = cfdp_malloc(nall * NFLUX * sizeof(double));
  memcpy(var0, sd->var, nall * NGRAD * sizeof(double));
  memcpy(grad0, sd->grad, nall * NGRAD * 3 * sizeof(double));
  memcpy(flux0, sd->psd_flux, nall * NFLUX * sizeof(double));
  cfdp_ipc_set_wait_seconds(2.0); /* a broken mapping must not cost half a minute per iteration here */
  /* the reference flux: two exchanging iterations (both grad buffers / arenas hold delivered rows), every device
   * synced, a barrier -- every push has landed everywhere -- then one iteration that does not exchange */
  for (int it = 0; it < 2; it++) {
    compute_gradients_gg_gaspi_async(cd, sd, 0);
    compute_psd_flux(sd);
  }
  int ok = cfdp_gpu_sync(gpu) == 0;
  MPI_Barrier(MPI_COMM_WORLD);
  compute_gradients_gg_comm_free(cd, sd, 1);
  compute_psd_flux(sd);
  cfdp_scaled_check ev;
  memset(&ev, 0, sizeof ev);
  ok = ok && cfdp_gpu_scaled_check_begin(gpu) == 0;
  MPI_Barrier(MPI_COMM_WORLD);
  for (int it = 0; ok && it < STEPS; it++) {
    compute_gradients_gg_gaspi_async(cd, sd, it == STEPS - 1);
    compute_psd_flux(sd);
  }
  ok = ok && cfdp_gpu_scaled_check_end(gpu, &ev) == 0;
  cfdp_sync_fields_to_host(sd);
  double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, gs[5];
  for (int i = 0; i < cd->ncommdomains; i++) {
    const int p = cd->commpartner[i];
    for (int j = 0; j < cd->sendcount[p]; j++)
      for (int c = 0; c < NGRAD * 3; c++) s[0] += (j + 1.0) * fabs((&sd->grad[cd->sendindex[p][j]][0][0])[c]);
    for (int j = 0; j < cd->recvcount[p]; j++)
      for (int c = 0; c < NGRAD * 3; c++) s[1] += (j + 1.0) * fabs((&sd->grad[cd->recvindex[p][j]][0][0])[c]);
  }
  s[2] = g_use_ipc ? (double)(cfdp_gpu_ipc_error(gpu) != 0) : 0.0;
  s[3] = (double)ev.mismatches;
  s[4] = ok && ev.flux_checks >= STEPS - 1 ? 0.0 : 1.0; /* the mode did not run as it should */
  MPI_Allreduce(s, gs, 5, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
  const char *failed = NULL;
  if (gs[2] != 0.0) failed = "wait timeout (a partner's flag never arrived)";
  else if (gs[4] != 0.0) failed = "the scaled-field run did not complete";
  else if (!(gs[0] > 0.0)) failed = "nothing sent";
  else if (gs[3] != 0.0) failed = "stale read (a flux phase read a ghost row of an earlier exchange)";
  else if (!(fabs(gs[0] - gs[1]) <= 1e-9 * gs[0])) failed = "stale rows (rows did not arrive although every flag did)";
  if (ev.mismatches && failed && !strncmp(failed, "stale read", 10))
    fprintf(stderr, "[cfdp] rank %d: first stale read in iteration %d, point %d, component %d: saw %.17g, expected %.17g\n",
            cd->iProc, ev.first_iteration, ev.first_point, ev.first_component, ev.seen, ev.expected);
  if (why) *why = failed;
  const char *w = getenv("CFDP_IPC_WAIT_SECONDS");
  cfdp_ipc_set_wait_seconds(w && atof(w) > 0 ? atof(w) : 10.0);
  memcpy(sd->var, var0, nall * NGRAD * sizeof(double));
  memcpy(sd->grad, grad0, nall * NGRAD * 3 * sizeof(double));
  memcpy(sd->psd_flux, flux0, nall * NFLUX * sizeof(double));
  cfdp_sync_fields_to_device(sd);
  free(var0); free(grad0); free(flux0);
  return failed == NULL;
}

/* choose, set up and VALIDATE the data path between the ranks' GPUs (collective).  force_rccl: skip
 * the IPC path.  Returns 1 = xGMI write + notify, 0 = RCCL; exits when neither works.          */
int cfdp_mpi_attach(comm_data *cd, solver_data *sd, int force_rccl) {
  int G = 1, r = 0;
  MPI_Comm_size(MPI_COMM_WORLD, &G);
  MPI_Comm_rank(MPI_COMM_WORLD, &r);
  g_use_ipc = 0;
  if (G == 1 || cd->ndomains == 1) return 0;
  cfdp_gpu *gpu = cfdp_dropin_context(sd);
  int try_ipc = !force_rccl && !(getenv("CFDP_MPI_FORCE_RCCL") && atoi(getenv("CFDP_MPI_FORCE_RCCL")));
  if (try_ipc) { /* one node?  then the ranks can map each other's memory */
    MPI_Comm node;
    int nsize = 0;
    MPI_Comm_split_type(MPI_COMM_WORLD, MPI_COMM_TYPE_SHARED, r, MPI_INFO_NULL, &node);
    MPI_Comm_size(node, &nsize);
    MPI_Comm_free(&node);
    int all_one_node = nsize == G, agreed = 0;
    MPI_Allreduce(&all_one_node, &agreed, 1, MPI_INT, MPI_MIN, MPI_COMM_WORLD);
    try_ipc = agreed;
  }
  /* ranks that SHARE a device must not wait inside the fused pass: the boundary tiles of every rank would sit in the
   * device's workgroup slots, spinning, while the pass whose flags they wait for cannot get a slot (measured: 4 ranks
   * at 128^3 on one MI355X starve each other until the bounded waits give up).  One waiting workgroup per rank (the
   * wait kernel) cannot exhaust the device.  Sharing is found by comparing the PCI bus ids of the ranks' devices -- not
   * ordinals or device counts: a launcher that shows every rank ONE device (ROCR_VISIBLE_DEVICES per rank) makes every
   * count 1 and every ordinal 0.  A CFDP_IPC_WAIT_INKERNEL in the environment wins. */
  int wait_inkernel = -1, per_device = 1;
  {
    char mine[64], *all = cfdp_malloc((size_t)G * 64);
    memset(mine, 0, sizeof mine);
    if (cfdp_gpu_device_bus_id(cfdp_gpu_device(gpu), mine, (int)sizeof mine)) snprintf(mine, sizeof mine, "rank%d", r);
    MPI_Allgather(mine, 64, MPI_BYTE, all, 64, MPI_BYTE, MPI_COMM_WORLD);
    for (int a = 0; a < G; a++) {
      int n = 0;
      for (int b = 0; b < G; b++) n += !strncmp(all + (size_t)a * 64, all + (size_t)b * 64, 64);
      if (n > per_device) per_device = n;
    }
    free(all);
    if (per_device > 1 && !getenv("CFDP_IPC_WAIT_INKERNEL")) wait_inkernel = 0;
    if (r == 0 && try_ipc)
      printf("exchange: %d ranks, at most %d per device: the wait for an exchange runs %s\n", G, per_device,
             wait_inkernel == 0 ? "as a kernel of its own (ranks share a device)" : "inside the fused pass (or as the environment says)");
  }
  /* the memory modes of the landing block (cfdproxy_hip.h, CFDP_IPC_MODE), in the order they are tried; a mode the
   * environment names is the only one tried */
  static const char *const modes[3] = {"fine", "coarse", "split"}; /* fine first: coherent by definition, and no slower in loopback */
  static const char *const labels[3] = {"fine-grained landing block", "coarse-grained landing block",
                                        "fine-grained flags, coarse-grained arenas, explicit invalidate"};
  static const int mode_id[3] = {1, 0, 2}; /* cfdp_gpu_ipc_configure: 0 coarse, 1 fine, 2 split */
  const char *preset = getenv("CFDP_IPC_MODE");
  const char *fg0 = getenv("CFDP_IPC_FINEGRAINED");
  if (!(preset && *preset) && fg0 && atoi(fg0) != 0) preset = "fine";
  /* per memory mode: notification by counters (fire-and-forget atomic adds), then by flags; CFDP_IPC_NOTIFY names one.
   * Every attempt is configured by argument -- the process environment stays as the user left it, so the attach of a
   * further multigrid level or solver starts from the user's presets, not from the last rung tried here */
  const char *npre = getenv("CFDP_IPC_NOTIFY");
  /* attempts 6-8: the conservative rung on the same mappings -- push, notify and wait as kernels of their own with flags
   * (release / acquire at kernel boundaries) -- still ahead of RCCL, whose steps cannot be replayed from a hipGraph in this
   * ROCm (priced in loopback: 53 / 25 us per iteration against 73 / 65 us, dualgrid.384 / .192 partitions) */
  char tried[12][200]; /* the resolved configurations validated so far, this attach */
  int ntried = 0;
  /* attempts 9-11: MPI_Put's pattern on the same mappings (src/exchange_data_mpidma.c:93-127) -- pack kernel, one copy per
   * partner slice into its landing slice, the notify kernel: a shade above RCCL in the same table (0.52 / 0.16), replayed from
   * hipGraphs, and the one rung that does not depend on a KERNEL's stores reaching a peer's memory */
  for (int attempt = 0; try_ipc && attempt < 12; attempt++) {
    const int put = attempt >= 9;
    const int separate = attempt >= 6;
    const int mi = put ? attempt - 9 : (separate ? attempt - 6 : attempt / 2), counters = separate ? 0 : attempt % 2 == 0;
    if (preset && *preset && strcmp(preset, modes[mi])) continue;
    if (!separate && npre && *npre && strcmp(npre, counters ? "counter" : "flag")) continue;
    if (separate && getenv("CFDP_IPC_INKERNEL")) continue;
    (void)cfdp_gpu_ipc_configure(gpu, mode_id[mi], wait_inkernel, counters, put ? 2 : (separate ? 0 : -1));
    char what[200];
    snprintf(what, sizeof what, "%s, %s notification%s", labels[mi], counters ? "counter" : "flag",
             put ? ", copy-engine put" : (separate ? ", push / notify / wait as kernels of their own" : ""));
    if (!ipc_setup(gpu, r, G)) {
      if (r == 0) printf("exchange: HIP IPC setup failed (%s): %s\n", what, cfdp_gpu_last_error());
      ipc_teardown(gpu);
      continue;
    }
    { /* what the library RESOLVED (bit 3 of cfdp_gpu_ipc_mode: notification by counters needs the per-partner protocol, which
       * depends on a rank's own partition): the rung is named by what runs.  Ranks of different forms understand each other
       * (the sender states what its word advances by, csrc/gg_kernels.h); a rung that resolves to one tried before is skipped */
      const int mine = (cfdp_gpu_ipc_mode(gpu) >> 3) & 1;
      int lo = 0, hi = 0;
      MPI_Allreduce(&mine, &lo, 1, MPI_INT, MPI_MIN, MPI_COMM_WORLD);
      MPI_Allreduce(&mine, &hi, 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD);
      snprintf(what, sizeof what, "%s, notification by %s%s", labels[mi], lo == hi ? (hi ? "counters" : "flags") : "counters on some ranks, flags on others",
               put ? ", copy-engine put" : (separate ? ", push / notify / wait as kernels of their own" : ""));
      int dup = 0;
      for (int k = 0; k < ntried; k++) dup |= !strcmp(tried[k], what);
      if (dup) {
        if (r == 0) printf("exchange: skipped (%s): resolves to a configuration already tried\n", what);
        ipc_teardown(gpu);
        continue;
      }
      if (ntried < 12) snprintf(tried[ntried++], sizeof tried[0], "%s", what);
    }
    cfdp_attach_ipc(sd);
    g_use_ipc = 1;
    const char *why = NULL;
    if (exchange_valid(cd, sd, gpu, &why)) {
      snprintf(g_path, sizeof g_path, "xGMI write + notify (HIP IPC, %s), validated", what);
      if (r == 0) printf("exchange: %s\n", g_path);
      return 1;
    }
    if (r == 0) printf("exchange: xGMI write + notify FAILED its validation (%s): %s\n", what, why ? why : "?");
    g_use_ipc = 0;
    cfdp_detach_external(sd);
    ipc_teardown(gpu);
  }
  unsigned char id[128];
  memset(id, 0, sizeof id);
  int ok = cfdp_rccl_load(getenv("CFDP_RCCL_LIB")) == 0;
  if (ok && r == 0) ok = cfdp_rccl_unique_id(id) == 0;
  int all_ok = 0; /* every rank's own result counts: a rank whose RCCL did not load must not be masked by rank 0's */
  MPI_Allreduce(&ok, &all_ok, 1, MPI_INT, MPI_MIN, MPI_COMM_WORLD);
  if (!all_ok) {
    if (!ok) fprintf(stderr, "Error (rank %d): no data path between the ranks' GPUs: RCCL unavailable (%s)\n", r, cfdp_gpu_last_error());
    MPI_Abort(MPI_COMM_WORLD, EXIT_FAILURE);
  }
  MPI_Bcast(id, 128, MPI_BYTE, 0, MPI_COMM_WORLD);
  cfdp_attach_rccl(sd, id, G, r);
  const char *why = NULL;
  if (!exchange_valid(cd, sd, gpu, &why)) {
    if (r == 0) fprintf(stderr, "Error: the RCCL exchange failed its validation: %s\n", why ? why : "?");
    MPI_Abort(MPI_COMM_WORLD, EXIT_FAILURE);
  }
  snprintf(g_path, sizeof g_path, "RCCL send/recv, validated");
  if (r == 0) printf("exchange: %s\n", g_path);
  return 0;
}

static void hook_attach(comm_data *cd, solver_data *sd) { (void)cfdp_mpi_attach(cd, sd, 0); }

static void hook_barrier(void) { MPI_Barrier(MPI_COMM_WORLD); }

/* 1 if a device-side wait gave up on ANY rank since the exchange was set up (collective) */
int cfdp_mpi_exchange_failed(solver_data *sd) {
  int e = g_use_ipc ? cfdp_gpu_ipc_error(cfdp_dropin_context(sd)) != 0 : 0, any = 0;
  MPI_Allreduce(&e, &any, 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD);
  return any;
}

static void hook_finalize(comm_data *cd) {
  int inited = 0, finalized = 0;
  MPI_Initialized(&inited);
  MPI_Finalized(&finalized);
  if (!inited || finalized) return;
  if (g_use_ipc && cd && cd->group) {
    cfdp_gpu *gpu = cfdp_group_context(cd);
    if (gpu) ipc_teardown(gpu);
    g_use_ipc = 0;
  }
  MPI_Barrier(MPI_COMM_WORLD);
  if (g_own_init) MPI_Finalize();
}

__attribute__((constructor)) static void cfdp_mpi_register(void) {
  static const cfdp_mpi_hooks hooks = {hook_init, hook_tables, hook_attach, hook_barrier, hook_finalize};
  cfdp_register_mpi_hooks(&hooks);
}
