/*
 * cfdproxy_dropin.h -- the drop-in boundary of the MI355X-native CFD-Proxy hot path.
 *
 * CFD-Proxy has no plugin/FFI layer: the "boundary" is the set of plain C structs and
 * functions that its driver (src/hybrid.f6.c:54-88) and harness (src/solver.c:35-314)
 * call.  This header declares exactly those, with the reference's names, argument
 * meaning and error behaviour (message + exit()), so that a host program written
 * against the reference headers recompiles against this one unchanged.  Each
 * declaration cites the reference interface it replaces.
 *
 * What is behind it is new: a NetCDF-classic reader of our own (no libnetcdf), a
 * domain merger (N dualgrid domains -> G GPUs), a graph tiler, and hand-written
 * gfx950 HIP kernels reached through the thin C ABI of cfdproxy_hip.h.
 *
 * Struct fields that exist in the reference keep their name, type and order; new
 * fields are only ever appended (marked "ext").
 */
#ifndef CFDPROXY_DROPIN_H
#define CFDPROXY_DROPIN_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants (reference src/solver_data.h:9-10, src/flux.h:7-9) ------------------- */
#define NGRAD 7 /* equations whose gradient is reconstructed         */
#define NFLUX 3 /* pseudo-flux components                            */
#define IVX 0
#define IVY 1
#define IVZ 2

/* ---- MPI / GASPI opaque types --------------------------------------------------------
 * reference src/comm_data.h:4-11 pulls in <mpi.h> (and typedefs the GASPI types when
 * GASPI is absent).  The GPU path needs neither; when CFDP_WITH_MPI is not defined the
 * MPI handle types are opaque placeholders so that the struct keeps its field list.   */
#ifdef CFDP_WITH_MPI
#include <mpi.h>
#else
typedef int MPI_Request;
typedef struct { int cfdp_opaque[5]; } MPI_Status;
#endif
typedef unsigned long gaspi_offset_t;
typedef unsigned short gaspi_notification_id_t;

/* 64-byte aligned counter (reference src/solver_data.h:12-15) */
typedef struct { int global __attribute__((aligned(64))); } counter_t;

/* per-thread private face copies (reference src/solver_data.h:22-26) */
typedef struct {
  int (*fpoint)[2];
  double (*fnormal)[3];
} solver_data_local;

/* colour descriptor (reference src/solver_data.h:28-63).  The GPU path replaces colours
 * by point tiles (cfdp_plan) but keeps the type so `solver_data.fcolor` still compiles. */
typedef struct RangeList_t {
  struct RangeList_t *succ;
  int start, stop, ftype;
  int nall_points_of_color;   int *all_points_of_color;
  int nfirst_points_of_color; int *first_points_of_color;
  int nlast_points_of_color;  int *last_points_of_color;
  int nsendcount; int *sendpartner; int *sendcount; int **sendindex; int **sendoffset;
  int nrecvcount; int *recvpartner; int *recvcount; int **recvindex; int **recvoffset;
  int tid;
} RangeList;

/* mesh + fields of one partition (reference src/solver_data.h:66-81) */
typedef struct {
  int nfaces;
  int nallfaces;
  int nownpoints;              /* points [0,nownpoints) are owned                       */
  int nallpoints;              /* points [nownpoints,nallpoints) are ghosts (addpoints) */
  int ncolors;
  int (*fpoint)[2];            /* face -> (p0,p1)                                       */
  double (*fnormal)[3];        /* face normal (area-weighted)                           */
  double *pvolume;             /* dual-cell volume per point                            */
  double (*var)[NGRAD];        /* primitive variables                                   */
  double (*grad)[NGRAD][3];    /* Green-Gauss gradients (the parity quantity)           */
  double (*psd_flux)[NFLUX];   /* pseudo viscous flux                                   */
  RangeList *fcolor;
  int niter;
  /* ext: */
  void *gpu;                   /* cfdp_solver* attached by init_threads(); NULL before  */
} solver_data;

/* halo topology of one partition (reference src/comm_data.h:15-55) */
typedef struct {
  int nProc;
  int iProc;
  int ndomains;
  int ncommdomains;
  int nownpoints;
  int naddpoints;
  int *addpoint_owner;         /* [naddpoints] owner rank of ghost j                    */
  int *addpoint_id;            /* [naddpoints] owner-local id of ghost j                */
  int *commpartner;            /* [ncommdomains]                                        */
  int *sendcount;              /* [ndomains], indexed by rank                           */
  int *recvcount;              /* [ndomains], indexed by rank                           */
  int **recvindex;             /* [ndomains][recvcount[k]] local ghost ids              */
  int **sendindex;             /* [ndomains][sendcount[k]] local own ids                */
  int nreq;
  MPI_Request *req;
  MPI_Status *stat;
  double **recvbuf;
  double **sendbuf;
  gaspi_offset_t *remote_recv_offset;
  gaspi_offset_t *local_recv_offset;
  gaspi_offset_t *local_send_offset;
  gaspi_notification_id_t *notification;
  volatile counter_t *recv_flag;
  volatile counter_t *send_flag;
  volatile int recv_stage;
  volatile int send_stage;
  volatile int comm_stage;
  /* ext: in-process rank group (one process drives G GPUs; rank g <-> peers[g])        */
  void *group;                 /* cfdp_group* shared by the G comm_data of a process    */
} comm_data;

/* ---- loader, libnetcdf's own entry points: the eight calls the reference makes
 * (src/hybrid.f6.c:65,91 + ERR() of src/error_handling.h:4-10: nc_open, nc_close, nc_strerror;
 * src/read_netcdf.c:25,28,38,41,53,56: nc_inq_dimid, nc_inq_dimlen, nc_inq_varid, nc_get_var_int,
 * nc_get_var_double), with libnetcdf's signatures and return convention (0 = NC_NOERR, else a code
 * nc_strerror() explains), backed by our own NetCDF-classic reader.  The reference's hybrid.f6.c
 * and read_netcdf.c link against this library unchanged (include/compat/ holds forwarding headers
 * under the reference's header names).  Read-only: `mode` is ignored.                          */
#ifndef NC_NOWRITE
#define NC_NOWRITE 0
#endif
#ifndef NC_NOERR
#define NC_NOERR 0
#endif
int nc_open(const char *path, int mode, int *ncidp);
int nc_close(int ncid);
const char *nc_strerror(int ncerr);
int nc_inq_dimid(int ncid, const char *name, int *idp);
int nc_inq_dimlen(int ncid, int dimid, size_t *lenp);
int nc_inq_varid(int ncid, const char *name, int *varidp);
int nc_get_var_int(int ncid, int varid, int *ip);
int nc_get_var_double(int ncid, int varid, double *ip);

/* ---- loader: reference src/read_netcdf.h:4-6.  `ncid` is a handle from nc_open() /
 * cfdp_nc_open().  Failure: message + exit(2), like ERR() in reference
 * src/error_handling.h:4-10.  cfdp_nc_open/_close = nc_open/nc_close with that ERR() built in. */
int  cfdp_nc_open(const char *path);
void cfdp_nc_close(int ncid);
void get_nc_double(int ncid, const char *name, double *array);
void get_nc_int(int ncid, const char *name, int *array);
int  get_nc_val(int ncid, const char *name);      /* value = a DIMENSION length         */

/* ---- reference src/solver_data.h:84-85 ---------------------------------------------- */
void read_solver_data(int ncid, solver_data *sd);
void init_solver_data(solver_data *sd, int NITER);

/* ---- reference src/comm_data.h:58-61 ------------------------------------------------- */
void init_communication(int argc, char *argv[], comm_data *cd);
void read_communication_data(int ncid, comm_data *cd);
void compute_communication_tables(comm_data *cd);
void free_communication_ressources(comm_data *cd);

/* ---- reference src/rangelist.h:17-20: preprocessing entry point.  Here: builds the GPU
 * tiling (cfdp_plan) for this partition and uploads it to the device.                  */
void init_threads(comm_data *cd, solver_data *sd, int NTHREADS);

/* ---- who enqueues a compute_gradients_gg_* / compute_psd_flux call (host/call_election.c).  The reference calls
 * them from EVERY thread of one `omp parallel` region (src/solver.c:45-55) and elects its first / last thread
 * inside (src/threads.c:142-179); here one caller per call enqueues the GPU work:
 *   CFDP_CALLS_AUTO  (default) outside a parallel region the caller performs every call it makes; inside a team of
 *                    T > 1 threads every thread must make every call, the first to arrive performs it -- a team that
 *                    does not (calls from omp single sections) is detected from the attendance of earlier calls and
 *                    the run stops with a message
 *   CFDP_CALLS_TEAM  the same without the attendance check
 *   CFDP_CALLS_EVERY no election: every call is performed (hosts that call from ONE thread at a time inside a region)
 * Environment: CFDP_CALL_MODE=auto|team|every (read at the first call unless cfdp_set_call_mode was called).
 * The team is the caller's OpenMP team as the HOST's OpenMP runtime sees it (the first one in the process's global
 * symbol order -- a host may bring a different runtime than this library links); cfdp_set_call_team(T) states the team
 * size for hosts whose callers are not an OpenMP team (T pthreads that all make every call); 0 = ask OpenMP again. */
enum { CFDP_CALLS_AUTO = 0, CFDP_CALLS_TEAM = 1, CFDP_CALLS_EVERY = 2 };
void cfdp_set_call_mode(int mode);
void cfdp_set_call_team(int nthreads);

/* ---- reference src/threads.h:13-24: callback types of the colour iterator.  Kept for
 * source compatibility; the GPU path orders pack/exchange by stream events instead.    */
typedef void (*send_fn)(RangeList *color, comm_data *cd, double *data, int dim2);
typedef void (*exch_fn)(comm_data *cd, double *data, int dim2, int final);

/* ---- reference src/gradients.h:7-25.  All variants enqueue the HIP gradient kernel on
 * the partition's device.  comm_free does no exchange; *_bulk_sync run gradient -> pack
 * -> xGMI exchange -> unpack in order; the async/early_recv variants run boundary tiles
 * first and overlap pack+exchange with the interior tiles on a second stream.          */
void compute_gradients_gg_comm_free(comm_data *cd, solver_data *sd, int final);
void compute_gradients_gg_mpi_bulk_sync(comm_data *cd, solver_data *sd, int final);
void compute_gradients_gg_mpi_early_recv(comm_data *cd, solver_data *sd, int final);
void compute_gradients_gg_mpi_async(comm_data *cd, solver_data *sd, int final);
void compute_gradients_gg_gaspi_bulk_sync(comm_data *cd, solver_data *sd, int final);
void compute_gradients_gg_gaspi_async(comm_data *cd, solver_data *sd, int final);
void compute_gradients_gg_mpifence_bulk_sync(comm_data *cd, solver_data *sd, int final);
void compute_gradients_gg_mpifence_async(comm_data *cd, solver_data *sd, int final);
void compute_gradients_gg_mpipscw_bulk_sync(comm_data *cd, solver_data *sd, int final);
void compute_gradients_gg_mpipscw_async(comm_data *cd, solver_data *sd, int final);

/* ---- the three priming calls the reference's harness makes before a sample of the early-receive /
 * one-sided variants (src/solver.c:87,106,183,220; src/exchange_data_mpi.h:37,
 * src/exchange_data_mpidma.h:40-42): pre-posting receives, opening a fence / PSCW epoch.  Here a partner
 * writes into this rank's ghost block (or landing arena) at any time -- it is always "posted" and there is
 * no window to open -- so they are no-ops, kept so that src/solver.c links unchanged.             */
void exchange_dbl_mpi_post_recv(comm_data *cd, int dim2);
void mpidma_async_win_fence(int assertion);
void mpidma_async_post_start(void);

/* ---- reference src/flux.h:12 -------------------------------------------------------- */
void compute_psd_flux(solver_data *sd);

/* ---- reference src/solver.h:7: the timing harness (25 samples x NITER iterations,
 * median in seconds per NITER iterations, reference src/solver.c:32-33,302-311).       */
void test_solver(comm_data *cd, solver_data *sd, int NTHREADS);

/* ---- the helpers the reference's main() and harness take from its util/error modules:
 * f_exist (src/error_handling.h:52, used at src/hybrid.f6.c:64), now (src/util.h:27: wall-clock
 * seconds, src/solver.c:43), sort_median (src/util.h:19; `end` is EXCLUSIVE as in src/util.c:61-78,
 * so test_solver's call sorts samples 0..23 -- kept, it is what the reference prints).         */
int f_exist(char *fname);
double now(void);
void sort_median(double *begin, double *end);

/* ---- ext: moving results across the boundary (the reference never reads a value back;
 * parity needs it).  Copies device grad / psd_flux into sd->grad / sd->psd_flux in FILE
 * numbering.  Ghost rows hold what the halo exchange delivered.                        */
void cfdp_sync_fields_to_host(solver_data *sd);
/* ext: push host sd->var (and sd->grad/psd_flux initial values) to the device.         */
void cfdp_sync_fields_to_device(solver_data *sd);

/* ---- ext: in-process rank group.  The reference is one MPI rank per partition; here one
 * process may drive G partitions (one per GPU).  The G comm_data share a group; a "send"
 * is a peer copy into the partner's ghost rows.                                         */
typedef struct cfdp_group cfdp_group;
cfdp_group *cfdp_group_create(int G, solver_data **sds, comm_data **cds);
/* build sendindex of raw (un-merged) partitions from the partners' ghost tables: the
 * MPI_Send/Recv of reference src/comm_data.c:203-249 without MPI                        */
void cfdp_group_link_raw(int G, comm_data **cds);
void cfdp_group_set_flux_mode(cfdp_group *grp, int mode);
void cfdp_group_destroy(cfdp_group *grp);
/* extension: the published "3V multigrid cycle" (documentation/CFD-Proxy.pdf p.3) over the
 * rank groups of several -lvl levels (finest first), `sweeps` iterations per level down and
 * up; prints median seconds per cycle in the style of test_solver's TIMINGS block
 * (src/solver.c:296-311).  Every level must have been through init_threads().              */
void cfdp_test_vcycle(int nlevels, cfdp_group **levels, int sweeps, int ncycles);

/* ---- ext: multi-process hooks.  The reference is one MPI rank per process; the library itself has no
 * MPI dependency.  lib/libcfdproxy_mpi.so (host/dropin_mpi.c, built when an MPI is installed) registers
 * these from a constructor, which gives the kept entry points their MPI meaning -- so the reference's
 * main() AND its harness run unchanged under mpiexec, one rank per GPU:
 *   init      in init_communication: MPI_Init_thread, nProc/iProc      (src/comm_data.c:257-307)
 *   tables    in compute_communication_tables: the sendindex exchange  (src/comm_data.c:203-249)
 *   attach    at the end of init_threads: the validated GPU-to-GPU data path (xGMI write + notify or RCCL)
 *   barrier   around test_solver's samples                              (src/solver.c:44,56)
 *   finalize  in free_communication_ressources                          (src/comm_data.c:504-521)    */
typedef struct {
  void (*init)(int *argc, char ***argv, comm_data *cd);
  void (*tables)(comm_data *cd);
  void (*attach)(comm_data *cd, solver_data *sd);
  void (*barrier)(void);
  void (*finalize)(comm_data *cd);
} cfdp_mpi_hooks;
void cfdp_register_mpi_hooks(const cfdp_mpi_hooks *hooks);
const cfdp_mpi_hooks *cfdp_get_mpi_hooks(void);

#ifdef __cplusplus
}
#endif
#endif /* CFDPROXY_DROPIN_H */
