/*
 * cfdproxy_hip.h -- the thin C ABI between the C host side and the gfx950 HIP kernels.
 *
 * Plain pointers, sizes and opaque handles only.  `stream` arguments are hipStream_t
 * values passed as void* (NULL = the context's own stream), so a caller that owns its
 * streams (a C driver, or PyTorch used as plumbing) can order work itself.
 *
 * Every function returns 0 on success and a non-zero code on failure;
 * cfdp_gpu_last_error() describes the last failure of the calling thread.  The drop-in
 * layer (cfdproxy_dropin.h) turns a failure into message + exit(), the reference's
 * convention (src/error_handling.h:25-30).
 *
 * Reference interfaces each entry point replaces:
 *   cfdp_gpu_upload_plan      <- init_threads()                       src/threads.c:730-788
 *   cfdp_gpu_gradients        <- private_compute_gradients_gg() over all colours of all
 *                                threads                               src/gradients.c:25-147
 *   cfdp_gpu_flux             <- private_compute_psd_flux()            src/flux.c:111-190
 *   cfdp_gpu_pack             <- exchange_dbl_copy_in[_local]()        src/threads.c:791-813,842-854
 *   cfdp_gpu_unpack           <- exchange_dbl_copy_out[_local]()       src/threads.c:816-839,857-869
 *   cfdp_gpu_send_ptr/recv_ptr<- cd->sendbuf[i] / cd->recvbuf[i]       src/exchange_data_mpi.c:27-76
 *   cfdp_gpu_rank_gradients / <- exchange_dbl_mpi_bulk_sync / _async, exchange_dbl_gaspi_*
 *   cfdp_gpu_rank_flux           (in-process ranks, peer copies over xGMI; the write +
 *                                notify pattern of src/exchange_data_gaspi.c:105-151)
 *                                                                      src/exchange_data_mpi.c:199-543
 */
#ifndef CFDPROXY_HIP_H
#define CFDPROXY_HIP_H

#include "cfdproxy_host.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cfdp_gpu cfdp_gpu; /* one partition resident on one device */

/* which tiles a gradient launch covers */
enum { CFDP_TILES_ALL = 0, CFDP_TILES_BOUNDARY = 1, CFDP_TILES_INTERIOR = 2 };
/* pseudo-flux semantics */
enum {
  CFDP_FLUX_CONSISTENT = 0, /* every owned point accumulates +/- flux of all its faces     */
  CFDP_FLUX_REFERENCE = 1   /* the reference's 1-thread result on owned points: flux.c uses
                               the README class numbering (src/flux.c:177-188 vs
                               src/rangelist.c:719-736), so the "+" side of a face whose two
                               ends are both owned is never added                          */
};
/* kernel variants (lanes per point); 0 = library default */
enum { CFDP_GRAD_DEFAULT = 0, CFDP_GRAD_L1 = 1, CFDP_GRAD_L2 = 2, CFDP_GRAD_L4 = 4, CFDP_GRAD_L8 = 8 };

int  cfdp_gpu_device_count(void);
/* PCI bus id of a device ("0000:c1:00.0", len >= 16): equal in every process that sees the same physical device whatever
 * its ordinal there -- ranks compare it to find out whether they share a device                                       */
int  cfdp_gpu_device_bus_id(int device, char *buf, int len);
int  cfdp_gpu_device(const cfdp_gpu *g); /* the ordinal the context was created on; -1 for a null context */
const char *cfdp_gpu_last_error(void);

int  cfdp_gpu_create(int device, cfdp_gpu **out);
void cfdp_gpu_destroy(cfdp_gpu *g);

/* the plan with its two heavy stages done by HIP kernels (csrc/plan_kernels.hip) -- the reference's
 * init_threads() preprocessing (src/rangelist.c:500-764, src/thread_comm.c:27-432) on the device: `which` bit 0 =
 * point->face CSR (atomics + scan + per-point sort), bit 1 = per-tile blobs (one workgroup per tile: first-touch
 * numbering by prefix sums and an LDS hash).  Tile growth (BFS), tile order, renumbering and pack lists stay on
 * the host.  The result is bit-identical to cfdp_plan_build()'s; free it with cfdp_plan_free().
 * stage_seconds (optional): [0] stage 1, [1] stage 5 (-1: fell back to the host stage), transfers included. */
int  cfdp_plan_build_gpu(const solver_data *sd, const comm_data *cd, const cfdp_plan_opts *opts, int device, int which,
                         cfdp_plan **out, double *stage_seconds);

/* copy the tiled mesh to the device and allocate the fields (var, grad, psd_flux, send
 * arena).  The plan may be freed afterwards.                                             */
int  cfdp_gpu_upload_plan(cfdp_gpu *g, const cfdp_plan *plan);
/* optional: use caller-owned device memory (16-byte aligned) for grad [nall*21 doubles] / the
 * send arena [nsend*21 doubles] (e.g. buffers registered with a communication library).
 * Device layout of grad (csrc/gg_kernels.h has the full statement): with g0..g8 the row-major 3x3
 * velocity-gradient block of a row and g9 its tenth double, a row is kept -- on the device and on the
 * wire alike -- as  [ g0 g4 g8 g1+g3 g2+g6 g5+g7 | g3 g6 g7 g9 | doubles 10..20 ]  =  [ A1 | A2 | B ]:
 * the first 48 bytes are all the flux loop reads.  Owned rows are split, ghost rows stay whole (168
 * bytes, message order):
 *   [A1: nown x 6][ghost rows: (nall-nown) x 21][A2: nown x 4][B: nown x 11]
 * cfdp_gpu_recv_ptr() points into the ghost block, cfdp_gpu_send_ptr() into the send arena: whole rows
 * in that stored form -- a transport moves them as opaque bytes.  cfdp_gpu_get_grad/_set_grad convert
 * from and to the reference's grad[nall][7][3] in file numbering; the three upper off-diagonals of the
 * block come back as (g1+g3)-g3, (g2+g6)-g6, (g5+g7)-g7: within one rounding of the pair's sum (1e-16 of
 * the scale the 1e-10 tolerance is written against), the same value wherever a row is seen.  (A row
 * taken off the device and put back keeps its sums to within one more rounding.)                     */
int  cfdp_gpu_bind_grad(cfdp_gpu *g, void *dev_grad);
int  cfdp_gpu_bind_sendbuf(cfdp_gpu *g, void *dev_sendbuf);

/* Fused iterations.  Both face loops read the same tile data (normals + incidence lists, more
 * than half of either kernel's HBM traffic).  With fusion on, the iteration-level entry points
 * (cfdp_gpu_step_pre/_post, cfdp_gpu_rank_gradients/_rank_flux, cfdp_gpu_iteration_group,
 * cfdp_gpu_run_iterations) DEFER the flux of iteration i and compute it in the same pass over
 * the tiles as the gradients of iteration i+1 (grad is double-buffered; a deferred flux is
 * flushed by cfdp_gpu_sync, the get/set calls and the eager launch calls).  Values are
 * bit-identical to the separate kernels.  Default: off.  cfdp_gpu_bind_grad_alt: caller-owned
 * memory for the second grad buffer (same size and layout as cfdp_gpu_bind_grad's); the two
 * buffers swap roles every fused iteration -- cfdp_gpu_grad_ptr/_recv_ptr always refer to the
 * one holding (receiving) the latest gradients.                                            */
int  cfdp_gpu_set_fusion(cfdp_gpu *g, int on);
int  cfdp_gpu_bind_grad_alt(cfdp_gpu *g, void *dev_grad);

/* host <-> device fields, FILE numbering on the host side: var [nall][7], grad
 * [nall][7][3], psd_flux [nall][3] (ghost rows of psd_flux are not computed)             */
int  cfdp_gpu_set_var(cfdp_gpu *g, const double *var);
int  cfdp_gpu_set_grad(cfdp_gpu *g, const double *grad);
int  cfdp_gpu_set_flux(cfdp_gpu *g, const double *psd_flux);
int  cfdp_gpu_get_grad(cfdp_gpu *g, double *grad);
int  cfdp_gpu_get_flux(cfdp_gpu *g, double *psd_flux);

/* launches (asynchronous) */
int  cfdp_gpu_set_variant(cfdp_gpu *g, int grad_lanes, int flux_lanes);
int  cfdp_gpu_gradients(cfdp_gpu *g, int which_tiles, void *stream);
int  cfdp_gpu_flux(cfdp_gpu *g, int mode, void *stream);
int  cfdp_gpu_pack(cfdp_gpu *g, void *stream);   /* grad rows of send points -> send arena */
int  cfdp_gpu_unpack(cfdp_gpu *g, const void *dev_recvbuf, void *stream); /* staging path  */
int  cfdp_gpu_sync(cfdp_gpu *g);
void *cfdp_gpu_stream(cfdp_gpu *g, int which /*0 main, 1 comm*/);

/* exchange geometry: partner slot s in [0, npartners) */
int  cfdp_gpu_npartners(const cfdp_gpu *g);
int  cfdp_gpu_partner_rank(const cfdp_gpu *g, int s);
void *cfdp_gpu_send_ptr(cfdp_gpu *g, int s, size_t *bytes);  /* slice of the send arena    */
void *cfdp_gpu_recv_ptr(cfdp_gpu *g, int s, size_t *bytes);  /* ghost rows of grad         */
void *cfdp_gpu_grad_ptr(cfdp_gpu *g);
void *cfdp_gpu_var_ptr(cfdp_gpu *g);

/* in-process ranks: one halo exchange among the G partitions of this process; sends of
 * rank a to rank b are device-to-device (peer) copies into b's ghost rows.  `bulk` != 0:
 * after the full gradient; else boundary tiles -> pack -> copies overlap interior tiles. */
int  cfdp_gpu_iteration_group(cfdp_gpu **ranks, int G, int with_exchange, int overlap,
                              int with_flux, int flux_mode);
/* the two phases of one rank's iteration (what the drop-in compute_gradients_gg_* and
 * compute_psd_flux enqueue): gradients [+ pack + peer copies], then wait-for-halo + flux */
int  cfdp_gpu_rank_gradients(cfdp_gpu **ranks, int G, int a, int with_exchange, int overlap);
int  cfdp_gpu_rank_flux(cfdp_gpu **ranks, int G, int b, int with_flux, int flux_mode);
/* the same phase 1 in two parts, for G host threads driving G devices ("thread t drives device t"): _launch enqueues rank
 * a's own kernels, _send its copies into the partners -- called after EVERY rank's _launch (a barrier between the
 * threads), and every rank's _send precedes any rank's cfdp_gpu_rank_flux (a second barrier)                          */
int  cfdp_gpu_rank_gradients_launch(cfdp_gpu **ranks, int G, int a, int with_exchange, int overlap);
int  cfdp_gpu_rank_gradients_send(cfdp_gpu **ranks, int G, int a);
/* hipDeviceEnablePeerAccess between the devices of the group's ranks (copies and stores then go over xGMI directly);
 * *npairs (optional) = ordered device pairs enabled                                                                  */
int  cfdp_gpu_enable_peer_access(cfdp_gpu **ranks, int G, int *npairs);
int  cfdp_gpu_sync_group(cfdp_gpu **ranks, int G);

/* one rank per process: the two brackets of an iteration around the caller's transport.
 * Between them the caller enqueues its sends (from cfdp_gpu_send_ptr) and receives (into
 * cfdp_gpu_recv_ptr) on cfdp_gpu_stream(g, 1); pre() has made that stream wait for the
 * pack, post() makes the flux wait for it.  Replaces the send_fn/exch_fn call-backs of
 * the colour iterator (src/rangelist.c:838-889) by stream order.                         */
int  cfdp_gpu_step_pre(cfdp_gpu *g, int with_exchange, int overlap);
int  cfdp_gpu_step_post(cfdp_gpu *g, int with_flux, int flux_mode);

/* one rank per process with the exchange issued from this library: grouped ncclSend/ncclRecv
 * (RCCL over xGMI) on the context's comm stream between the two brackets -- one call per
 * iteration (cfdp_gpu_step_rccl) or per batch of iterations (cfdp_gpu_run_steps_rccl).
 * Replaces exchange_dbl_mpi_send/_post_recv and the MPI_Waitany loop
 * (src/exchange_data_mpi.c:96-166,199-543).
 *   cfdp_rccl_load       resolve RCCL at run time: `libpath` = the librccl.so the process
 *                        already uses (PyTorch ships one), NULL/"" = the system library
 *   cfdp_rccl_unique_id  128-byte ncclUniqueId (rank 0; the caller broadcasts it)
 *   cfdp_gpu_rccl_init   ncclCommInitRank; rank_of_partner[s] = communicator rank of partner
 *                        slot s (NULL: the partition's partner ranks are communicator ranks) */
int  cfdp_rccl_load(const char *libpath);
int  cfdp_rccl_unique_id(void *id128);
int  cfdp_gpu_rccl_init(cfdp_gpu *g, const void *id128, int nranks, int rank, const int *rank_of_partner);
/* MEASUREMENT / PLUMBING ONLY: a communicator of ONE rank whose partner slots all map to rank 0 exchanges with ITSELF
 * (how the RCCL fall-back is priced and its plumbing tested on a 1-GPU box); a send then meets a receive of its own
 * length -- both are cut to the shorter of the two.  Off by default: an exchange on a one-rank communicator is refused
 * with a message, not silently truncated.                                                                          */
int  cfdp_gpu_rccl_allow_self_exchange(cfdp_gpu *g, int on);
int  cfdp_gpu_rccl_finalize(cfdp_gpu *g);
int  cfdp_gpu_rccl_nranks(const cfdp_gpu *g); /* ncclCommCount of the context's communicator; 0 without one */
int  cfdp_gpu_exchange_rccl(cfdp_gpu *g);   /* between cfdp_gpu_step_pre and cfdp_gpu_step_post */
int  cfdp_gpu_step_rccl(cfdp_gpu *g, int with_exchange, int overlap, int with_flux, int flux_mode);
/* drop-in layer, one rank per process (e.g. MPI-launched, bin/hybrid.f6.hip.mpi): the context
 * init_threads() built for `sd`, and its RCCL communicator -- afterwards compute_gradients_gg_*
 * exchange with the other processes through it (replaces init_mpi_requests / the GASPI segment
 * setup, src/exchange_data_mpi.c:27-76, src/exchange_data_gaspi.c:38-103)                     */
cfdp_gpu *cfdp_dropin_context(solver_data *sd);
void cfdp_attach_rccl(solver_data *sd, const void *unique_id128, int nranks, int rank);
/* the same for the xGMI write + notify exchange: call after cfdp_gpu_ipc_export / _connect /
 * _ready on cfdp_dropin_context(sd)                                                          */
void cfdp_attach_ipc(solver_data *sd);
/* undo cfdp_attach_rccl / cfdp_attach_ipc (a transport that failed its validation is being replaced);
 * the context of the rank `cd` stands for in this process                                       */
void cfdp_detach_external(solver_data *sd);
cfdp_gpu *cfdp_group_context(comm_data *cd);
int  cfdp_gpu_run_steps_rccl(cfdp_gpu *g, int steps, int with_exchange, int overlap, int with_flux,
                             int flux_mode);

/* one rank per process on one node, exchange by xGMI write + notify -- the analogue of the
 * reference's best variant, gaspi_write_notify + gaspi_notify_waitsome
 * (src/exchange_data_gaspi.c:105-151,190-305): the packing kernel writes each partner's rows
 * straight into that partner's landing arena (mapped through a HIP IPC handle), a second kernel
 * raises the iteration counter in the partner's flag word, the receiver's stream polls its own
 * flag words before the ghost rows are read.  No communication library in the iteration, and a
 * run of iterations is one hipGraph replay (cfdp_gpu_run_steps_ipc).
 *   cfdp_gpu_ipc_export   allocate this rank's block [256-byte header | arena 0 | arena 1] and
 *                         return its 64-byte IPC handle; *land_bytes = size of one arena.  A
 *                         partner's rows for partner slot t land at header + parity*land_bytes +
 *                         recv_off[t]*168, its arrival counter is the int at 4*t.
 *   cfdp_gpu_ipc_connect  for my partner slot: the partner's handle and where, in ITS block, my
 *                         rows land (both parities) and my arrival counter lives
 *   cfdp_gpu_ipc_ready    switch the context over (the ghost block is then the landing arenas)
 *   cfdp_gpu_ipc_error    1 if a wait for a partner gave up (bounded polling)
 * Memory of the block, CFDP_IPC_MODE (the hosts try fine, coarse, split in this order and keep the first that passes the
 * scaled-field validation; without the variable a context allocates coarse): "fine" (everything fine-grained: coherent
 * between devices by definition; CFDP_IPC_FINEGRAINED=1 is the older spelling), "coarse" (hipMalloc; system-scope loads
 * and write-through stores in the kernels), "split" (the flag words in a small fine-grained block of their own, the
 * arenas coarse-grained, an explicit cache invalidate once a tile has seen its partners' flags).
 *   cfdp_gpu_ipc_export_flags   the handle of the block that holds this rank's flag words (the main block's handle
 *                               again unless the mode is "split")
 *   cfdp_gpu_ipc_connect_flags  after cfdp_gpu_ipc_connect, before _ready: my arrival counter at partner `slot` lives
 *                               at flag_off in the block of THAT handle
 * Notification (src/threads.c:268-311, src/exchange_data_gaspi.c:389-416): the boundary tile that completes partner k's
 * rows raises k's flag at once, and a boundary tile waits only for the partners it exchanges with -- when every boundary
 * tile reads ghost rows only of partners it holds send points for (checked at _ready; else, or with
 * CFDP_IPC_PER_PARTNER=0, the last boundary tile raises all flags and every tile waits for all).
 *   cfdp_gpu_ipc_mode     bit 0 the fused pass pushes / notifies, bit 1 its tiles wait themselves, bit 2 per-partner
 *                         notification, bit 3 notification by counters, bits 4-5 memory mode (0 coarse, 1 fine,
 *                         2 split), bit 6 the copy-engine put rung; -1 without a block                                */
/* A rank's block starts with CFDP_IPC_HEADER_BYTES of flag / counter words; landing arena 0 follows, then arena 1.
 * Notification (CFDP_IPC_NOTIFY, cfdp_gpu_ipc_configure): "counter" (default where the per-partner protocol holds) -- a
 * partner's word counts the boundary tiles that have completed their rows for it, raised by fire-and-forget system-scope
 * atomic adds (nothing returns to the tile; the reference's notification travels with the write as well,
 * src/exchange_data_gaspi.c:134-145), a waiter compares it with tiles-per-exchange x exchanges; "flag" -- the tile that
 * completes a partner's rows stores the exchange number (two dependent device-scope atomics decide which tile that is).
 * cfdp_gpu_ipc_ready writes into every partner's header what this rank's word advances by per exchange -- its boundary tiles
 * that count towards that partner when it notifies by counters, 1 when it stores its exchange number -- and every wait
 * compares a word with exchanges x THAT, whatever the waiting rank's own form: neighbours that resolved to different forms
 * (the per-partner protocol depends on a rank's own partition) understand each other.  The hosts must meet (a barrier, a
 * collective) between _ready on every rank and the first exchanging step.
 *   cfdp_gpu_ipc_configure  per context, by argument instead of through the environment (-1 = environment / default):
 *                           memory_mode 0 coarse | 1 fine | 2 split; wait_inkernel 1 | 0 (ranks sharing a device: 0);
 *                           notify 1 counters | 0 flags; push_inkernel 1 the fused pass pushes and notifies itself |
 *                           0 push, notify and wait are kernels of their own (release / acquire at kernel boundaries:
 *                           the conservative rung) | 2 the copy-engine put: the send arena packed by a kernel, one
 *                           hipMemcpyAsync per partner slice into its landing slice, the notify kernel behind the copies
 *                           (MPI_Put's pattern, src/exchange_data_mpidma.c:93-127).  Takes effect at the next _export /
 *                           _ready.                                                                                   */
#define CFDP_IPC_HEADER_BYTES 8192
int  cfdp_gpu_ipc_header_bytes(void);
size_t cfdp_gpu_ipc_flag_offset(int slot); /* flag_off of cfdp_gpu_ipc_connect[_flags]: one cache line per partner slot */
int  cfdp_gpu_ipc_configure(cfdp_gpu *g, int memory_mode, int wait_inkernel, int notify, int push_inkernel);
int  cfdp_gpu_ipc_export(cfdp_gpu *g, void *handle64, size_t *land_bytes);
int  cfdp_gpu_ipc_connect(cfdp_gpu *g, int slot, const void *partner_handle64, size_t land_off0,
                          size_t land_off1, size_t flag_off);
int  cfdp_gpu_ipc_export_flags(cfdp_gpu *g, void *handle64);
int  cfdp_gpu_ipc_connect_flags(cfdp_gpu *g, int slot, const void *partner_flags_handle64, size_t flag_off);
int  cfdp_gpu_ipc_mode(const cfdp_gpu *g);
/* how the steps of cfdp_gpu_run_steps_ipc have run on this context: replayed from hipGraphs / launched from the streams
 * (lead-in steps, odd remainders, short runs) / captures that were abandoned (0 in every schedule the library selects) */
int  cfdp_gpu_ipc_graph_stats(cfdp_gpu *g, long *steps_replayed, long *steps_streamed, long *captures_failed);
/* measurement only: partner slot `slot` is this rank itself (its own arenas, its own flag word): the cost of the protocol
 * with a partner that is never late; the ghost rows then hold this rank's own send rows                              */
int  cfdp_gpu_ipc_connect_loopback(cfdp_gpu *g, int slot);
int  cfdp_gpu_ipc_ready(cfdp_gpu *g);
int  cfdp_gpu_ipc_enable(cfdp_gpu *g, int on);   /* keep the mappings, use / do not use them */
int  cfdp_gpu_ipc_disconnect(cfdp_gpu *g);
int  cfdp_gpu_ipc_error(cfdp_gpu *g);
int  cfdp_ipc_set_wait_seconds(double seconds);   /* bound of the device-side waits (default 10 s) */
int  cfdp_gpu_step_ipc_pre(cfdp_gpu *g, int with_exchange, int overlap);  /* gradients + exchange ... */
int  cfdp_gpu_step_ipc_post(cfdp_gpu *g, int with_flux, int flux_mode);    /* ... then the flux         */
int  cfdp_gpu_step_ipc(cfdp_gpu *g, int with_exchange, int overlap, int with_flux, int flux_mode);
int  cfdp_gpu_run_steps_ipc(cfdp_gpu *g, int steps, int with_exchange, int overlap, int with_flux,
                            int flux_mode, int use_graph);

/* Scaled-field validation of an exchange -- the analogue of the reference's stage / flag lock-step asserts at every
 * receive (src/exchange_data_mpi.c:189,439, src/exchange_data_gaspi.c:389-416).  The benchmark's field is constant in
 * time, so a ghost row READ one exchange too early (double-buffered landing arenas: the row of two exchanges ago, same
 * address, same value) is invisible to any comparison of final states.  Between _begin and _end every step entry point
 * (cfdp_gpu_step_post, cfdp_gpu_step_ipc[_post], cfdp_gpu_run_steps_ipc / _rccl, cfdp_gpu_rank_flux, the drop-in
 * compute_psd_flux) ends with one more kernel: it compares the flux the step produced with reference * 2^e bit for bit and
 * then multiplies var by 2, 2, 1/4, 2, 2, 1/4, ... (exact), so iteration k's gradients, ghost rows and flux are those of
 * the first iteration times 2^((k-1) mod 3) and a term taken from a row of two exchanges ago is off by 2x or 4x.  The
 * state lives on the device: the steps replay from hipGraphs as usual.
 *   _begin  the flux the context holds now becomes the reference: the caller has run one iteration whose exchange is
 *           known to be complete (device sync on every rank, a barrier, one more step WITHOUT exchange)
 *   _end    compares the last iteration's deferred flux too, restores var exactly, returns the evidence
 * While the mode is on every step must exchange and compute the flux.                                                 */
typedef struct cfdp_scaled_check {
  int iterations;       /* steps run in the mode                                                        */
  int flux_checks;      /* flux fields compared (one per step that produced one)                        */
  int mismatches;       /* flux values that were not reference * 2^e (saturates)                        */
  int first_iteration;  /* the iteration (1-based, in the mode) whose flux held the first mismatch found, or 0 */
  int first_point;      /* ... its point, file numbering, or -1                                         */
  int first_component;
  double seen, expected;
  int var_mismatches;   /* elements of var that were not var(begin) * 2^(iterations mod 3) at _end: the mode's own
                           bookkeeping -- must be 0 for the verdict to mean anything                          */
} cfdp_scaled_check;
int  cfdp_gpu_scaled_check_begin(cfdp_gpu *g);
int  cfdp_gpu_scaled_check_end(cfdp_gpu *g, cfdp_scaled_check *out);

/* measurement: `iters` back-to-back launches bracketed by HIP events on the context's
 * main stream; average milliseconds per launch (gradient over all tiles; flux)           */
int  cfdp_gpu_time_kernels(cfdp_gpu *g, int iters, int flux_mode, float *ms_grad, float *ms_flux);
/* average milliseconds of the fused pass (flux(i) + gradients(i+1), all tiles); fusion on  */
int  cfdp_gpu_time_fused(cfdp_gpu *g, int iters, int flux_mode, float *ms_fused);
/* diagnostics (needs lib/libcfdproxy_diag.so): shader-clock stamps (s_memtime: one clock per XCD) of the last of `passes` fused
 * passes.  `stamps` takes 24 words per tile: [ntiles][8] -- start, indices here, loads landed, flux done, var rows in place,
 * gradients done, stores acknowledged, and the tile's place (XCC_ID register << 32 | HW_ID register) -- then [ntiles][4 waves][4]
 * -- the wave's own pieces landed, through its flux phase, through its gradient phase, through the tile (rows pushed, tile counted) */
int  cfdp_gpu_debug_phase_stamps(cfdp_gpu *g, int passes, unsigned long long *stamps);
/* the same for steps of the write + notify schedule (after cfdp_gpu_ipc_ready), with or without the exchange in the pass */
int  cfdp_gpu_debug_phase_stamps_ipc(cfdp_gpu *g, int passes, int with_exchange, unsigned long long *stamps);
/* the schedule of an exchange step without the exchange itself (the two brackets only), from one
 * hipGraph or from the streams: average milliseconds per step                             */
int  cfdp_gpu_time_schedule(cfdp_gpu *g, int steps, int with_exchange, int overlap, int use_graph, float *ms_step);
/* K full iterations (gradients [+flux]) replayed from hipGraphs -- whole chunks (50 fused passes
 * / 25 iterations, the reference's NITER, src/hybrid.f6.c:72) plus one graph for the remainder, so
 * any K runs without per-kernel stream launches; ms_total: device time of the K iterations
 * (HIP events around them), or NULL: no event pair, the call returns when the stream is through  */
int  cfdp_gpu_run_iterations(cfdp_gpu *g, int iters, int with_flux, int flux_mode,
                             int use_graph, float *ms_total);
/* the data-movement floor of the fused pass: the same kernel without its two face loops (every load and every store of
 * the pass; a diagnostic instantiation), timed as cfdp_gpu_time_fused times the real one; ms_pass = milliseconds per
 * pass.  grad / flux hold one correct iteration afterwards.                                                       */
int  cfdp_gpu_time_fused_movement(cfdp_gpu *g, int iters, float *ms_pass);
/* which kernel forms ran: every launch of a face-loop kernel made by the calling thread since the last call is appended to a
 * log, "form@first_tile+tiles" separated by blanks (e.g. "fused_split<6,4,3,4>listed@0+4124 flux_dma<3,2>@4124+5") --
 * what bench.py prints beside a roofline figure so that a number says which instantiation it belongs to.  Copies the log
 * into buf (truncated to len - 1 characters), clears it, returns the number of launches it covered.  The first call switches
 * the log on (and returns nothing); buf = NULL switches it off again.                                                 */
int  cfdp_gpu_kernel_forms(char *buf, size_t len);
/* capture + instantiate the graphs cfdp_gpu_run_iterations(g, iters, ...) will replay; nothing
 * executes (keeps the capture out of a caller's timed region)                                  */
int  cfdp_gpu_prepare_iterations(cfdp_gpu *g, int iters, int with_flux, int flux_mode);
/* instantiate every cached graph (those of cfdp_gpu_run_iterations and of cfdp_gpu_run_steps_ipc) again from the graph
 * it was captured as; syncs the device first.  For a caller that times a SHORT run on an idle device: an executable
 * graph instantiated before other work went through the device starts 60-100 us later than one instantiated just before
 * its launch (measured); replays queued behind running work do not see that.  ~50 us per graph.
 * The FIRST replay of a freshly instantiated graph pays the runtime's set-up for it (~15 us, and a variable amount): a
 * caller that times a short run replays it once, untimed, between this call and its timed replay (EXPERIMENTS.md E.12). */
int  cfdp_gpu_refresh_graphs(cfdp_gpu *g);

/* multigrid "3V cycle" (documentation/CFD-Proxy.pdf p.3; levels = the -lvl files of
 * src/hybrid.f6.c:38-47, no transfer operators in the reference): `sweeps` iterations
 * (gradients + flux) on levels[0] (finest) .. levels[nlevels-1], then back up to levels[0];
 * one partition per level, all on one device, no exchange.  The whole cycle is ONE hipGraph
 * (use_graph) replayed `cycles` times; average milliseconds per cycle.                     */
int  cfdp_gpu_vcycle(cfdp_gpu **levels, int nlevels, int sweeps, int cycles, int flux_mode,
                     int use_graph, float *ms_per_cycle);

/* sizes for callers that allocate */
int  cfdp_gpu_counts(const cfdp_gpu *g, int *nown, int *nall, int *nsend, int *nrecv);

#ifdef __cplusplus
}
#endif
#endif
