/*
 * comm_tables.c -- halo index tables of a partition (no HIP, no MPI).
 *
 *   compute_communication_tables  <- reference src/comm_data.c:446-502
 *                                    (create_recvsend_index :116-255, offsets :309-443)
 *   cfdp_group_link_raw           <- the MPI_Send/Recv index exchange of :203-249, done by
 *                                    reading the partners' tables when all ranks live in
 *                                    one process
 */
#include "cfdproxy_host.h"
#include "host_util.h"

#include <string.h>

static const cfdp_mpi_hooks *g_hooks = NULL;
void cfdp_register_mpi_hooks(const cfdp_mpi_hooks *hooks) { g_hooks = hooks; }
const cfdp_mpi_hooks *cfdp_get_mpi_hooks(void) { return g_hooks; }

void compute_communication_tables(comm_data *cd) {
  /* reference: create_recvsend_index (MPI exchange of ghost ids) + offset tables + buffer
   * allocation (src/comm_data.c:446-502).  Merged partitions arrive with recvindex built
   * (domain_merge.c) and sendindex linked (cfdp_group_link()); a raw single file gets its
   * recvindex here exactly like src/comm_data.c:161-174.  Buffers live on the device.   */
  CFDP_ASSERT(cd != NULL);
  if (cd->ndomains == 1) return;
  CFDP_ASSERT(cd->naddpoints != 0 && cd->addpoint_owner != NULL && cd->addpoint_id != NULL);
  CFDP_ASSERT(cd->commpartner != NULL && cd->sendcount != NULL && cd->recvcount != NULL);
  if (!cd->recvindex) {
    cd->recvindex = cfdp_calloc((size_t)cd->ndomains, sizeof(int *));
    for (int i = 0; i < cd->ncommdomains; i++) {
      int k = cd->commpartner[i], count = 0;
      if (cd->recvcount[k] <= 0) continue;
      cd->recvindex[k] = cfdp_malloc((size_t)cd->recvcount[k] * sizeof(int));
      for (int j = 0; j < cd->naddpoints; j++)
        if (cd->addpoint_owner[j] == k) cd->recvindex[k][count++] = cd->nownpoints + j;
      CFDP_ASSERT(count == cd->recvcount[k]);
    }
  }
  if (!cd->sendindex) cd->sendindex = cfdp_calloc((size_t)cd->ndomains, sizeof(int *));
  /* one rank per process: the partners tell each other which points they need (src/comm_data.c:203-249);
   * only for raw single-file partitions -- merged ones get their send lists from the merger's requests */
  int raw = 1;
  for (int i = 0; i < cd->ncommdomains; i++)
    if (cd->sendindex[cd->commpartner[i]]) raw = 0;
  if (g_hooks && g_hooks->tables && raw && cd->nProc > 1) g_hooks->tables(cd);
}


/* link the send side of G in-process partitions: what rank s receives from r (ghost
 * (owner-local id) lists) becomes r's sendindex[s] -- the MPI_Send/Recv of
 * src/comm_data.c:203-249 done by reading the partner's tables directly               */
void cfdp_group_link_raw(int G, comm_data **cds) {
  for (int r = 0; r < G; r++)
    for (int s = 0; s < G; s++) {
      if (s == r || !cds[s]->recvcount || cds[s]->recvcount[r] <= 0) continue;
      comm_data *me = cds[r], *other = cds[s];
      int n = other->recvcount[r];
      CFDP_ASSERT(me->sendcount[s] == n);
      free(me->sendindex[s]);
      me->sendindex[s] = cfdp_malloc((size_t)n * sizeof(int));
      for (int j = 0; j < n; j++) {
        int ghost = other->recvindex[r][j] - other->nownpoints;
        me->sendindex[s][j] = other->addpoint_id[ghost];
      }
    }
}

