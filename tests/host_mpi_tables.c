/*
 * host_mpi_tables.c -- TEST HOST (compiled by tests/test_multirank.py, run under mpiexec on the CPU).
 *
 * The MPI side of the drop-in boundary without a GPU: init_communication (-> MPI_Init_thread through the hook of
 * libcfdproxy_mpi.so), the reference's loader calls, compute_communication_tables (-> the sendindex exchange of
 * src/comm_data.c:203-249).  One rank per domain file, as the reference runs.  Check: the points this rank will
 * send to partner k, in message order, are -- by GLOBAL lattice id -- exactly the ghosts k expects from this rank,
 * in k's file order (src/comm_data.c:161-174).
 *
 *   mpiexec -n N host_mpi_tables PREFIX LVL NX NY NZ
 */
#define CFDP_WITH_MPI 1
#include "cfdproxy_host.h"

#include <stdio.h>
#include <stdlib.h>

int main(int argc, char *argv[]) {
  if (argc < 6) return 1;
  comm_data cd;
  solver_data sd;
  int ncid, retval;
  init_communication(argc, argv, &cd);
  char fname[4096];
  snprintf(fname, sizeof fname, "%s_domain_%d_lvl_%d", argv[1], cd.iProc, atoi(argv[2]));
  if ((retval = nc_open(fname, NC_NOWRITE, &ncid))) { printf("Error: %s\n", nc_strerror(retval)); return 2; }
  read_solver_data(ncid, &sd);
  init_solver_data(&sd, 25);
  read_communication_data(ncid, &cd);
  compute_communication_tables(&cd);
  if (cd.nProc != cd.ndomains) { fprintf(stderr, "rank %d: nProc %d != ndomains %d\n", cd.iProc, cd.nProc, cd.ndomains); return 3; }

  cfdp_gen_params gp = {atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), cd.ndomains, 7, 1, 1, 0, 1, 20241};
  int *gid = malloc((size_t)sd.nallpoints * sizeof(int));
  if (cfdp_gen_global_ids(&gp, cd.iProc, gid) != sd.nallpoints) { fprintf(stderr, "rank %d: generator disagrees with the file\n", cd.iProc); return 4; }
  int bad = 0, nreq = 0;
  MPI_Request *req = malloc((size_t)(2 * cd.ncommdomains + 1) * sizeof(MPI_Request));
  int **mine = calloc((size_t)cd.ncommdomains + 1, sizeof(int *)), **theirs = calloc((size_t)cd.ncommdomains + 1, sizeof(int *));
  for (int i = 0; i < cd.ncommdomains; i++) {
    const int k = cd.commpartner[i];
    if (!cd.sendindex || (cd.sendcount[k] > 0 && !cd.sendindex[k])) { fprintf(stderr, "rank %d: no sendindex for partner %d\n", cd.iProc, k); return 5; }
    mine[i] = malloc((size_t)(cd.recvcount[k] + 1) * sizeof(int));
    theirs[i] = malloc((size_t)(cd.sendcount[k] + 1) * sizeof(int));
    for (int j = 0; j < cd.recvcount[k]; j++) mine[i][j] = gid[cd.recvindex[k][j]]; /* the ghosts I expect from k */
    MPI_Isend(mine[i], cd.recvcount[k], MPI_INT, k, 99, MPI_COMM_WORLD, &req[nreq++]);
    MPI_Irecv(theirs[i], cd.sendcount[k], MPI_INT, k, 99, MPI_COMM_WORLD, &req[nreq++]);
  }
  MPI_Waitall(nreq, req, MPI_STATUSES_IGNORE);
  long rows = 0;
  for (int i = 0; i < cd.ncommdomains; i++) {
    const int k = cd.commpartner[i];
    for (int j = 0; j < cd.sendcount[k]; j++) {
      const int p = cd.sendindex[k][j];
      if (p < 0 || p >= sd.nownpoints || gid[p] != theirs[i][j]) bad++;
    }
    rows += cd.sendcount[k];
  }
  int allbad = 0;
  long allrows = 0;
  MPI_Allreduce(&bad, &allbad, 1, MPI_INT, MPI_SUM, MPI_COMM_WORLD);
  MPI_Allreduce(&rows, &allrows, 1, MPI_LONG, MPI_SUM, MPI_COMM_WORLD);
  if (cd.iProc == 0) printf("ranks %d  send rows %ld  mismatches %d\n%s\n", cd.nProc, allrows, allbad, allbad || !allrows ? "*** FAILURE" : "*** SUCCESS");
  if ((retval = nc_close(ncid))) return 2;
  free_communication_ressources(&cd); /* MPI_Finalize through the hook */
  return allbad || !allrows ? 1 : 0;
}
