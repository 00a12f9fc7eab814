/*
 * hybrid_f6_hip_mpi.c -- the driver launched the way the reference is: one MPI rank per
 * process (reference src/hybrid.f6.c:27-101, init_communication src/comm_data.c:257-307),
 * here one rank per GPU:
 *
 *     mpiexec -n G hybrid.f6.hip.mpi -lvl [1-4] GRID_PREFIX [--flux-ref] [--var one|hash]
 *                                    [--cluster] [--rccl] [--dry-run]
 *
 * MPI is the control plane only: rank/size, the broadcast of the ncclUniqueId and the exchange
 * of the (domain, idx) request lists -- what create_recvsend_index does with MPI_Send/Recv
 * (src/comm_data.c:203-249) -- or, when all ranks share a node (and --rccl is not given), of
 * the HIP IPC handles of the landing arenas.  The data path of an iteration is then xGMI write +
 * notify from kernels (cfdp_attach_ipc), otherwise RCCL over xGMI issued by the library
 * (cfdp_attach_rccl); compute_gradients_gg_* / compute_psd_flux are the reference's entry
 * points.  Rank r merges the N/G domain files that fall to it and times the same three
 * variants as test_solver, with MPI_Barrier + device sync around every sample
 * (src/solver.c:42-58).  --dry-run stops before the GPU is touched and checks the halo tables
 * (usable on a machine without GPUs).
 */
#define CFDP_WITH_MPI 1
#include "cfdproxy_hip.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define N_MEDIAN 25
#define N_VARIANT 3

static int cmp_double(const void *a, const void *b) {
  double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

int main(int argc, char *argv[]) {
  int provided = 0;
  MPI_Init_thread(&argc, &argv, MPI_THREAD_SERIALIZED, &provided);
  int G = 1, r = 0;
  MPI_Comm_size(MPI_COMM_WORLD, &G);
  MPI_Comm_rank(MPI_COMM_WORLD, &r);
  if (argc < 4 || strcmp(argv[1], "-lvl") != 0) {
    if (r == 0)
      printf("Usage: mpiexec -n G %s -lvl [1-4] GRID_PREFIX [--flux-ref] [--var one|hash] [--cluster] [--dry-run]\n",
             argv[0]);
    MPI_Finalize();
    return EXIT_FAILURE;
  }
  const int lvl = atoi(argv[2]);
  const char *prefix = argv[3];
  int flux_ref = 0, var_hash = 0, cluster = 0, dry = 0, force_rccl = 0;
  for (int i = 4; i < argc; i++) {
    if (!strcmp(argv[i], "--flux-ref")) flux_ref = 1;
    else if (!strcmp(argv[i], "--var") && i + 1 < argc) var_hash = !strcmp(argv[++i], "hash");
    else if (!strcmp(argv[i], "--cluster")) cluster = 1;
    else if (!strcmp(argv[i], "--dry-run")) dry = 1;
    else if (!strcmp(argv[i], "--rccl")) force_rccl = 1;
  }
  char fname[4096];
  snprintf(fname, sizeof fname, "%s_domain_%d_lvl_%d", prefix, 0, lvl);
  int ncid = cfdp_nc_open(fname);
  const int N = get_nc_val(ncid, "ndomains");
  cfdp_nc_close(ncid);
  if (G > N) {
    if (r == 0) fprintf(stderr, "Error: %d ranks for %d domains\n", G, N);
    MPI_Abort(MPI_COMM_WORLD, EXIT_FAILURE);
  }

  /* ---- domains -> ranks (every rank needs the same map) ---- */
  if (cluster && G > 1) {
    int *map = malloc((size_t)N * sizeof(int));
    if (r == 0) {
      int *xadj, *adj, *wgt;
      cfdp_domain_graph(prefix, lvl, N, &xadj, &adj, &wgt);
      long cut = cfdp_cluster_domains(N, G, xadj, adj, wgt, map);
      printf("%d domains clustered onto %d ranks, %ld halo points between ranks\n", N, G, cut);
      free(xadj); free(adj); free(wgt);
    }
    MPI_Bcast(map, N, MPI_INT, 0, MPI_COMM_WORLD);
    cfdp_set_domain_map(map, N, G);
    free(map);
  }

  /* ---- load and merge this rank's domains ---- */
  int *ids = malloc((size_t)N * sizeof(int));
  const int count = cfdp_rank_domain_list(r, N, G, ids);
  solver_data *ds = calloc((size_t)count, sizeof(solver_data));
  comm_data *dc = calloc((size_t)count, sizeof(comm_data));
  for (int i = 0; i < count; i++) {
    init_communication(argc, argv, &dc[i]);
    cfdp_load_domain(prefix, ids[i], lvl, &ds[i], &dc[i]);
  }
  solver_data sd;
  comm_data cd;
  cfdp_merge_info *info = NULL;
  cfdp_merge_domains(count, ids, ds, dc, N, G, r, &sd, &cd, &info);
  for (int i = 0; i < count; i++) { cfdp_free_solver_data(&ds[i]); cfdp_free_comm_data(&dc[i]); }
  free(ds); free(dc); free(ids);

  /* ---- request lists: tell every partner which of its points this rank needs
   * (the MPI index exchange of src/comm_data.c:203-249) ---- */
  int *want = calloc((size_t)G, sizeof(int)), *asked = calloc((size_t)G, sizeof(int));
  for (int i = 0; i < info->npartners; i++) want[info->partner[i]] = info->want_off[i + 1] - info->want_off[i];
  MPI_Alltoall(want, 1, MPI_INT, asked, 1, MPI_INT, MPI_COMM_WORLD);
  MPI_Request *req = malloc((size_t)(2 * G) * sizeof(MPI_Request));
  int **inbox = calloc((size_t)G, sizeof(int *)), **outbox = calloc((size_t)G, sizeof(int *));
  int nreq = 0;
  for (int p = 0; p < G; p++)
    if (asked[p] > 0) {
      inbox[p] = malloc((size_t)asked[p] * 2 * sizeof(int));
      MPI_Irecv(inbox[p], 2 * asked[p], MPI_INT, p, 77, MPI_COMM_WORLD, &req[nreq++]);
    }
  for (int i = 0; i < info->npartners; i++) {
    const int p = info->partner[i], off = info->want_off[i], n = want[p];
    outbox[p] = malloc((size_t)n * 2 * sizeof(int));
    memcpy(outbox[p], info->ghost_domain + off, (size_t)n * sizeof(int));
    memcpy(outbox[p] + n, info->ghost_idx + off, (size_t)n * sizeof(int));
    MPI_Isend(outbox[p], 2 * n, MPI_INT, p, 77, MPI_COMM_WORLD, &req[nreq++]);
  }
  MPI_Waitall(nreq, req, MPI_STATUSES_IGNORE);
  long nsend = 0, nrecv = sd.nallpoints - sd.nownpoints;
  for (int p = 0; p < G; p++)
    if (asked[p] > 0) {
      cfdp_merge_set_send(&cd, info, p, asked[p], inbox[p], inbox[p] + asked[p]);
      nsend += asked[p];
    }
  for (int p = 0; p < G; p++) { free(inbox[p]); free(outbox[p]); }
  free(inbox); free(outbox); free(req); free(want); free(asked);
  long tot[2] = {nsend, nrecv}, all[2] = {0, 0};
  MPI_Allreduce(tot, all, 2, MPI_LONG, MPI_SUM, MPI_COMM_WORLD);
  printf("rank %d/%d: %d domains, %d own + %d ghost points, %d faces, %d partners, sends %ld rows\n", r, G, count,
         sd.nownpoints, sd.nallpoints - sd.nownpoints, sd.nfaces, cd.ncommdomains, nsend);
  if (all[0] != all[1]) {
    fprintf(stderr, "Error: %ld rows sent but %ld ghost rows expected\n", all[0], all[1]);
    MPI_Abort(MPI_COMM_WORLD, EXIT_FAILURE);
  }
  if (dry) {
    MPI_Barrier(MPI_COMM_WORLD);
    if (r == 0) printf("halo tables consistent: %ld rows per iteration\n*** SUCCESS (dry run)\n", all[0]);
    MPI_Finalize();
    return 0;
  }

  /* ---- GPU side: plan, upload, communicator ---- */
  if (var_hash) cfdp_fill_var(sd.var, NULL, sd.nallpoints, CFDP_VAR_HASH, 1, 1, 1);
  compute_communication_tables(&cd);
  init_threads(&cd, &sd, 0);
  cfdp_gpu *gpu = cfdp_dropin_context(&sd);
  if (flux_ref) cfdp_group_set_flux_mode((cfdp_group *)cd.group, CFDP_FLUX_REFERENCE);
  int use_ipc = 0;
  if (G > 1 && !force_rccl) { /* one node? then the ranks can map each other's memory */
    MPI_Comm node;
    int nsize = 0;
    MPI_Comm_split_type(MPI_COMM_WORLD, MPI_COMM_TYPE_SHARED, r, MPI_INFO_NULL, &node);
    MPI_Comm_size(node, &nsize);
    MPI_Comm_free(&node);
    use_ipc = nsize == G;
  }
  if (use_ipc) {
    /* every rank publishes {handle, arena size, partner list, receive offsets}; rank r's rows for
     * its partner p land in p's block at header + parity*arena + recv_off_p[slot of r] rows */
    enum { MAXP = 48 };
    typedef struct { unsigned char handle[64]; long land; int np, partner[MAXP], recv_off[MAXP + 1]; } ipc_info;
    ipc_info mine, *all_info = malloc((size_t)G * sizeof(ipc_info));
    memset(&mine, 0, sizeof mine);
    size_t land = 0;
    int ok = cfdp_gpu_ipc_export(gpu, mine.handle, &land) == 0 && cfdp_gpu_npartners(gpu) <= MAXP;
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
    for (int s = 0; ok && s < mine.np; s++) {
      const ipc_info *pi = &all_info[mine.partner[s]];
      int t = -1;
      for (int i = 0; i < pi->np; i++)
        if (pi->partner[i] == r) t = i;
      const size_t base = 256 + (size_t)pi->recv_off[t < 0 ? 0 : t] * NGRAD * 3 * sizeof(double);
      ok = t >= 0 && cfdp_gpu_ipc_connect(gpu, s, pi->handle, base, base + (size_t)pi->land, 4 * (size_t)t) == 0;
    }
    ok = ok && cfdp_gpu_ipc_ready(gpu) == 0;
    int all_ok = 0;
    MPI_Allreduce(&ok, &all_ok, 1, MPI_INT, MPI_MIN, MPI_COMM_WORLD); /* also: nobody pushes before everybody is ready */
    free(all_info);
    if (all_ok) {
      cfdp_attach_ipc(&sd);
      if (r == 0) printf("exchange: xGMI write + notify (HIP IPC)\n");
    } else {
      if (r == 0) printf("exchange: HIP IPC setup failed (%s), using RCCL\n", ok ? "another rank" : cfdp_gpu_last_error());
      MPI_Barrier(MPI_COMM_WORLD);
      cfdp_gpu_ipc_disconnect(gpu);
      use_ipc = 0;
    }
  }
  if (G > 1 && !use_ipc) {
    unsigned char id[128];
    if (cfdp_rccl_load(getenv("CFDP_RCCL_LIB"))) { fprintf(stderr, "Error: %s\n", cfdp_gpu_last_error()); MPI_Abort(MPI_COMM_WORLD, 1); }
    if (r == 0 && cfdp_rccl_unique_id(id)) { fprintf(stderr, "Error: %s\n", cfdp_gpu_last_error()); MPI_Abort(MPI_COMM_WORLD, 1); }
    MPI_Bcast(id, 128, MPI_BYTE, 0, MPI_COMM_WORLD);
    cfdp_attach_rccl(&sd, id, G, r);
    if (r == 0) printf("exchange: RCCL send/recv\n");
  }

  /* ---- test_solver across processes ---- */
  typedef void (*grad_fn)(comm_data *, solver_data *, int);
  const char *names[N_VARIANT] = {"comm_free", use_ipc ? "exchange_dbl_xgmi_notify_bulk_sync" : "exchange_dbl_rccl_bulk_sync",
                                  use_ipc ? "exchange_dbl_xgmi_notify_async" : "exchange_dbl_rccl_async"};
  grad_fn fns[N_VARIANT] = {compute_gradients_gg_comm_free, compute_gradients_gg_mpi_bulk_sync,
                            compute_gradients_gg_gaspi_async};
  const int nvar = G == 1 ? 1 : N_VARIANT;
  double median[N_VARIANT][N_MEDIAN];
  for (int k = 0; k < N_MEDIAN; k++) {
    for (int v = 0; v < nvar; v++) {
      cfdp_gpu_sync(gpu);
      MPI_Barrier(MPI_COMM_WORLD);
      double t = -MPI_Wtime();
      for (int i = 0; i < sd.niter; i++) {
        fns[v](&cd, &sd, i == sd.niter - 1);
        compute_psd_flux(&sd);
      }
      cfdp_gpu_sync(gpu);
      MPI_Barrier(MPI_COMM_WORLD);
      t += MPI_Wtime();
      median[v][k] = t;
    }
    if (r == 0) { printf("."); fflush(stdout); }
  }
  if (r == 0) {
    printf("\n\n*** SETUP\n");
    printf("                                 nProc: %d\n", G);
    printf("                                 NITER: %d\n", sd.niter);
    printf("                              N_MEDIAN: %d\n", N_MEDIAN);
    printf("\n*** TIMINGS\n");
    for (int v = 0; v < nvar; v++) {
      qsort(median[v], N_MEDIAN, sizeof(double), cmp_double);
      printf("%38s: %10.6f\n", names[v], median[v][(N_MEDIAN - 1) / 2]);
    }
  }
  /* every sent row must have arrived: sum of |ghost rows| == sum of |packed send rows| */
  cfdp_sync_fields_to_host(&sd);
  double sums[2] = {0.0, 0.0}, gs[2] = {0.0, 0.0};
  for (int i = 0; i < cd.ncommdomains; i++) {
    const int p = cd.commpartner[i];
    for (int j = 0; j < cd.sendcount[p]; j++)
      for (int c = 0; c < NGRAD * 3; c++) sums[0] += fabs((&sd.grad[cd.sendindex[p][j]][0][0])[c]);
  }
  for (int q = sd.nownpoints; q < sd.nallpoints; q++)
    for (int c = 0; c < NGRAD * 3; c++) sums[1] += fabs((&sd.grad[q][0][0])[c]);
  MPI_Allreduce(sums, gs, 2, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
  const int ok = G == 1 || fabs(gs[0] - gs[1]) <= 1e-9 * (gs[0] > 1e-300 ? gs[0] : 1e-300);
  if (r == 0) printf("\nexchange check: sent %.12e received %.12e %s\n", gs[0], gs[1], ok ? "ok" : "MISMATCH");
  if (use_ipc) {
    int e = cfdp_gpu_ipc_error(gpu), any = 0;
    MPI_Allreduce(&e, &any, 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD);
    if (any && r == 0) printf("exchange: a device-side wait for a partner timed out\n");
    MPI_Barrier(MPI_COMM_WORLD); /* nobody unmaps a block a partner may still write to */
    cfdp_gpu_ipc_disconnect(gpu);
    MPI_Barrier(MPI_COMM_WORLD);
  }
  free_communication_ressources(&cd);
  MPI_Barrier(MPI_COMM_WORLD);
  if (r == 0) printf(ok ? "*** SUCCESS\n" : "*** FAILURE\n");
  MPI_Finalize();
  return ok ? 0 : EXIT_FAILURE;
}
