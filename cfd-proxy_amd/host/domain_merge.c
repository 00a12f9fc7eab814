/*
 * domain_merge.c -- N dualgrid domains -> one partition per GPU rank.
 *
 * The reference runs one MPI rank per domain file (ASSERT(ndomains == nProc),
 * src/comm_data.c:94) and learns which of its points a neighbour needs by exchanging
 * index lists over MPI (create_recvsend_index, src/comm_data.c:116-255).  On a node of
 * G GPUs each rank instead owns N/G whole domains (BASELINE.json configs: 12/1, 48/4,
 * 192/8, 384/8).  Merging turns halo points whose owner lives on the same GPU into
 * ordinary interior points (the ghost (owner, idx) pair of src/comm_data.c:108-112 is
 * resolved to the owner's row), removes the second copy of every face that crosses an
 * internal domain boundary, and leaves ghosts only for points owned by other ranks,
 * grouped by partner rank in message order so that a received message is a contiguous
 * block of ghost rows (zero-copy unpack).
 */
#include "cfdproxy_host.h"
#include "host_util.h"

#include <string.h>

/* Optional explicit domain -> rank map (cfdp_set_domain_map).  Default: contiguous blocks of
 * domain ids, which is right when the ids are spatially coherent (our generator numbers RCB
 * leaves in order); for files whose numbering is not, cfdp_cluster_domains() builds a map from
 * the commpartner graph.  Set once during setup, before any merge.                          */
static int *g_domain_map = NULL;
static int g_domain_map_n = 0, g_domain_map_g = 0;

void cfdp_set_domain_map(const int *rank_of_domain, int N, int G) {
  free(g_domain_map);
  g_domain_map = NULL;
  g_domain_map_n = g_domain_map_g = 0;
  if (!rank_of_domain) return;
  CFDP_ASSERT(N >= 1 && G >= 1);
  g_domain_map = cfdp_malloc((size_t)N * sizeof(int));
  for (int d = 0; d < N; d++) {
    CFDP_ASSERT(rank_of_domain[d] >= 0 && rank_of_domain[d] < G);
    g_domain_map[d] = rank_of_domain[d];
  }
  g_domain_map_n = N;
  g_domain_map_g = G;
}

int cfdp_domain_rank(int domain, int N, int G) {
  if (g_domain_map && g_domain_map_n == N && g_domain_map_g == G) return g_domain_map[domain];
  int base = N / G, rem = N % G;
  int cut = rem * (base + 1);
  if (domain < cut) return domain / (base + 1);
  return rem + (domain - cut) / base;
}

void cfdp_rank_domains(int r, int N, int G, int *first, int *count) {
  int base = N / G, rem = N % G;
  *first = r * base + (r < rem ? r : rem);
  *count = base + (r < rem ? 1 : 0);
}

int cfdp_rank_domain_list(int r, int N, int G, int *ids) {
  int n = 0;
  for (int d = 0; d < N; d++)
    if (cfdp_domain_rank(d, N, G) == r) ids[n++] = d; /* ascending, as cfdp_merge_domains wants */
  return n;
}

/* Greedy graph growing on the domain graph (CSR xadj/adj, edge weight = halo points exchanged):
 * G clusters of N/G (+1) domains.  A cluster starts from the unassigned domain with the fewest
 * unassigned neighbours (a corner of what is left) and repeatedly takes the unassigned domain
 * with the heaviest connection to it, so clusters are compact and the halo between ranks small.
 * Returns the total weight of the edges cut between ranks.                                  */
long cfdp_cluster_domains(int N, int G, const int *xadj, const int *adj, const int *wgt, int *rank_of_domain) {
  CFDP_ASSERT(N >= 1 && G >= 1 && G <= N);
  long *gain = cfdp_calloc((size_t)N, sizeof(long));
  for (int d = 0; d < N; d++) rank_of_domain[d] = -1;
  int assigned = 0;
  for (int r = 0; r < G; r++) {
    const int target = N / G + (r < N % G ? 1 : 0);
    for (int d = 0; d < N; d++) gain[d] = 0;
    int size = 0;
    while (size < target && assigned < N) {
      int best = -1;
      if (size > 0) { /* heaviest connection to the growing cluster */
        for (int d = 0; d < N; d++)
          if (rank_of_domain[d] < 0 && gain[d] > 0 && (best < 0 || gain[d] > gain[best])) best = d;
      }
      if (best < 0) { /* seed (or the cluster's component is exhausted): a corner of the rest */
        int best_deg = 0;
        for (int d = 0; d < N; d++) {
          if (rank_of_domain[d] >= 0) continue;
          int deg = 0;
          for (int e = xadj[d]; e < xadj[d + 1]; e++) deg += rank_of_domain[adj[e]] < 0;
          if (best < 0 || deg < best_deg) { best = d; best_deg = deg; }
        }
      }
      rank_of_domain[best] = r;
      size++;
      assigned++;
      for (int e = xadj[best]; e < xadj[best + 1]; e++) gain[adj[e]] += wgt ? wgt[e] : 1;
    }
  }
  CFDP_ASSERT(assigned == N);
  long cut = 0;
  for (int d = 0; d < N; d++)
    for (int e = xadj[d]; e < xadj[d + 1]; e++)
      if (rank_of_domain[adj[e]] != rank_of_domain[d]) cut += wgt ? wgt[e] : 1;
  free(gain);
  return cut / 2;
}

/* the domain graph of a set of dualgrid files: neighbours = commpartner, weight = recvcount
 * (src/comm_data.c:79-112).  Arrays are malloc'd; the caller frees them.                     */
int cfdp_domain_graph(const char *prefix, int lvl, int N, int **pxadj, int **padj, int **pwgt) {
  int *xadj = cfdp_calloc((size_t)N + 1, sizeof(int));
  int cap = 16 * N + 16, n = 0;
  int *adj = cfdp_malloc((size_t)cap * sizeof(int)), *wgt = cfdp_malloc((size_t)cap * sizeof(int));
  for (int d = 0; d < N; d++) {
    solver_data sd;
    comm_data cd;
    memset(&cd, 0, sizeof cd);
    cd.nProc = N; cd.iProc = d;
    cfdp_load_domain(prefix, d, lvl, &sd, &cd);
    for (int i = 0; i < cd.ncommdomains; i++) {
      const int k = cd.commpartner[i];
      if (n == cap) {
        cap *= 2;
        adj = realloc(adj, (size_t)cap * sizeof(int));
        wgt = realloc(wgt, (size_t)cap * sizeof(int));
      }
      adj[n] = k;
      wgt[n] = cd.recvcount ? cd.recvcount[k] : 1;
      n++;
    }
    xadj[d + 1] = n;
    cfdp_free_solver_data(&sd);
    cfdp_free_comm_data(&cd);
  }
  *pxadj = xadj; *padj = adj; *pwgt = wgt;
  return 0;
}

typedef struct { int rank, domain, idx; } ext_key;

static int ext_cmp(const void *a, const void *b) {
  const ext_key *x = (const ext_key *)a, *y = (const ext_key *)b;
  if (x->rank != y->rank) return x->rank < y->rank ? -1 : 1;
  if (x->domain != y->domain) return x->domain < y->domain ? -1 : 1;
  if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
  return 0;
}

static int ext_find(const ext_key *keys, int n, ext_key k) {
  int lo = 0, hi = n - 1;
  while (lo <= hi) {
    int mid = (lo + hi) / 2;
    int c = ext_cmp(&keys[mid], &k);
    if (c == 0) return mid;
    if (c < 0) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

int cfdp_merge_domains(int ndl, const int *domain_ids, const solver_data *sds,
                       const comm_data *cds, int N, int G, int r, solver_data *out_sd,
                       comm_data *out_cd, cfdp_merge_info **pinfo) {
  CFDP_ASSERT(ndl >= 1 && G >= 1 && r >= 0 && r < G && N >= 1);
  cfdp_merge_info *info = cfdp_calloc(1, sizeof(*info));
  info->G = G; info->r = r; info->ndomains_total = N; info->ndom_local = ndl;
  info->domain_ids = cfdp_malloc((size_t)ndl * sizeof(int));
  info->own_offset = cfdp_malloc((size_t)(ndl + 1) * sizeof(int));
  info->local2merged = cfdp_calloc((size_t)ndl, sizeof(int *));
  int *dom2local = cfdp_malloc((size_t)N * sizeof(int));
  for (int d = 0; d < N; d++) dom2local[d] = -1;
  info->own_offset[0] = 0;
  size_t next_total = 0;
  for (int dl = 0; dl < ndl; dl++) {
    int d = domain_ids[dl];
    CFDP_ASSERT(d >= 0 && d < N && dom2local[d] < 0);
    CFDP_ASSERT(dl == 0 || d > domain_ids[dl - 1]);
    CFDP_ASSERT(cfdp_domain_rank(d, N, G) == r);
    info->domain_ids[dl] = d;
    dom2local[d] = dl;
    info->own_offset[dl + 1] = info->own_offset[dl] + sds[dl].nownpoints;
    next_total += (size_t)(sds[dl].nallpoints - sds[dl].nownpoints);
  }
  const int nown = info->own_offset[ndl];

  /* external ghosts: unique (rank, domain, idx), sorted => grouped by partner rank */
  ext_key *keys = cfdp_malloc((next_total ? next_total : 1) * sizeof(ext_key));
  size_t nk = 0;
  for (int dl = 0; dl < ndl; dl++) {
    const int nadd = sds[dl].nallpoints - sds[dl].nownpoints;
    if (nadd) CFDP_ASSERT(cds[dl].addpoint_owner && cds[dl].addpoint_id);
    for (int j = 0; j < nadd; j++) {
      int e = cds[dl].addpoint_owner[j];
      CFDP_ASSERT(e >= 0 && e < N && e != domain_ids[dl]);
      if (dom2local[e] >= 0) continue;
      ext_key k = {cfdp_domain_rank(e, N, G), e, cds[dl].addpoint_id[j]};
      keys[nk++] = k;
    }
  }
  qsort(keys, nk, sizeof(ext_key), ext_cmp);
  int nghost = 0;
  for (size_t i = 0; i < nk; i++)
    if (i == 0 || ext_cmp(&keys[i], &keys[i - 1]) != 0) keys[nghost++] = keys[i];
  const int nall = nown + nghost;
  info->nghost = nghost;
  info->ghost_domain = cfdp_malloc((size_t)(nghost ? nghost : 1) * sizeof(int));
  info->ghost_idx = cfdp_malloc((size_t)(nghost ? nghost : 1) * sizeof(int));
  for (int j = 0; j < nghost; j++) {
    info->ghost_domain[j] = keys[j].domain;
    info->ghost_idx[j] = keys[j].idx;
  }

  /* file numbering -> merged numbering */
  for (int dl = 0; dl < ndl; dl++) {
    const int no = sds[dl].nownpoints, na = sds[dl].nallpoints;
    int *m = cfdp_malloc((size_t)na * sizeof(int));
    for (int i = 0; i < no; i++) m[i] = info->own_offset[dl] + i;
    for (int j = 0; j < na - no; j++) {
      int e = cds[dl].addpoint_owner[j], idx = cds[dl].addpoint_id[j];
      if (dom2local[e] >= 0) {
        CFDP_ASSERT(idx >= 0 && idx < sds[dom2local[e]].nownpoints);
        m[no + j] = info->own_offset[dom2local[e]] + idx;
      } else {
        ext_key k = {cfdp_domain_rank(e, N, G), e, idx};
        int pos = ext_find(keys, nghost, k);
        CFDP_ASSERT(pos >= 0);
        m[no + j] = nown + pos;
      }
    }
    info->local2merged[dl] = m;
  }

  /* faces: keep every face with an owned end exactly once */
  size_t nf_max = 0;
  for (int dl = 0; dl < ndl; dl++) nf_max += (size_t)sds[dl].nfaces;
  memset(out_sd, 0, sizeof(*out_sd));
  out_sd->fpoint = cfdp_malloc(nf_max * 2 * sizeof(int));
  out_sd->fnormal = cfdp_malloc(nf_max * 3 * sizeof(double));
  size_t nf = 0;
  for (int dl = 0; dl < ndl; dl++) {
    const int d = domain_ids[dl], no = sds[dl].nownpoints;
    const int *m = info->local2merged[dl];
    for (int f = 0; f < sds[dl].nfaces; f++) {
      int a = sds[dl].fpoint[f][0], b = sds[dl].fpoint[f][1];
      int oa = a < no ? d : cds[dl].addpoint_owner[a - no];
      int ob = b < no ? d : cds[dl].addpoint_owner[b - no];
      int keep;
      if (oa != d && ob != d) keep = 0;                 /* ghost-ghost: no owned end here */
      else if (oa == d && ob == d) keep = 1;            /* interior of the domain         */
      else {
        int e = (oa == d) ? ob : oa;                    /* the foreign owner              */
        keep = (dom2local[e] < 0) || (d < e);           /* internal cut: lower id keeps   */
      }
      if (!keep) { info->nfaces_dropped++; continue; }
      out_sd->fpoint[nf][0] = m[a];
      out_sd->fpoint[nf][1] = m[b];
      memcpy(out_sd->fnormal[nf], sds[dl].fnormal[f], 3 * sizeof(double));
      nf++;
    }
    info->nfaces_in += sds[dl].nfaces;
  }
  out_sd->fpoint = realloc(out_sd->fpoint, (nf ? nf : 1) * 2 * sizeof(int));
  out_sd->fnormal = realloc(out_sd->fnormal, (nf ? nf : 1) * 3 * sizeof(double));
  out_sd->nfaces = out_sd->nallfaces = (int)nf;
  out_sd->nownpoints = nown;
  out_sd->nallpoints = nall;
  out_sd->ncolors = 1;
  out_sd->pvolume = cfdp_malloc((size_t)nall * sizeof(double));
  out_sd->var = cfdp_malloc((size_t)nall * NGRAD * sizeof(double));
  out_sd->grad = cfdp_malloc((size_t)nall * NGRAD * 3 * sizeof(double));
  out_sd->psd_flux = cfdp_malloc((size_t)nall * NFLUX * sizeof(double));
  for (int dl = 0; dl < ndl; dl++) {
    const int *m = info->local2merged[dl];
    for (int i = 0; i < sds[dl].nallpoints; i++) out_sd->pvolume[m[i]] = sds[dl].pvolume[i];
  }
  init_solver_data(out_sd, sds[0].niter ? sds[0].niter : 25);

  /* halo topology of the merged partition */
  memset(out_cd, 0, sizeof(*out_cd));
  out_cd->nProc = G; out_cd->iProc = r; out_cd->ndomains = G;
  out_cd->nownpoints = nown; out_cd->naddpoints = nghost;
  out_cd->sendcount = cfdp_calloc((size_t)G, sizeof(int));
  out_cd->recvcount = cfdp_calloc((size_t)G, sizeof(int));
  out_cd->sendindex = cfdp_calloc((size_t)G, sizeof(int *));
  out_cd->recvindex = cfdp_calloc((size_t)G, sizeof(int *));
  out_cd->addpoint_owner = cfdp_malloc((size_t)(nghost ? nghost : 1) * sizeof(int));
  out_cd->addpoint_id = cfdp_malloc((size_t)(nghost ? nghost : 1) * sizeof(int));
  for (int j = 0; j < nghost; j++) {
    out_cd->addpoint_owner[j] = keys[j].rank;
    out_cd->addpoint_id[j] = keys[j].idx;
    out_cd->recvcount[keys[j].rank]++;
  }
  int np = 0;
  for (int s = 0; s < G; s++) np += out_cd->recvcount[s] > 0;
  info->npartners = np;
  info->partner = cfdp_malloc((size_t)(np ? np : 1) * sizeof(int));
  info->want_off = cfdp_calloc((size_t)np + 1, sizeof(int));
  out_cd->ncommdomains = np;
  out_cd->commpartner = cfdp_malloc((size_t)(np ? np : 1) * sizeof(int));
  np = 0;
  int pos = 0;
  for (int s = 0; s < G; s++) {
    if (!out_cd->recvcount[s]) continue;
    info->partner[np] = s;
    out_cd->commpartner[np] = s;
    info->want_off[np] = pos;
    out_cd->recvindex[s] = cfdp_malloc((size_t)out_cd->recvcount[s] * sizeof(int));
    for (int j = 0; j < out_cd->recvcount[s]; j++) out_cd->recvindex[s][j] = nown + pos + j;
    pos += out_cd->recvcount[s];
    np++;
  }
  info->want_off[np] = pos;
  CFDP_ASSERT(pos == nghost);
  free(keys);
  free(dom2local);
  *pinfo = info;
  return 0;
}

int cfdp_merge_set_send(comm_data *out_cd, const cfdp_merge_info *info, int s, int count,
                        const int *want_domain, const int *want_idx) {
  CFDP_ASSERT(s >= 0 && s < info->G && s != info->r && count >= 0);
  free(out_cd->sendindex[s]);
  out_cd->sendindex[s] = cfdp_malloc((size_t)(count ? count : 1) * sizeof(int));
  out_cd->sendcount[s] = count;
  for (int j = 0; j < count; j++) {
    int dl = -1;
    for (int i = 0; i < info->ndom_local; i++)
      if (info->domain_ids[i] == want_domain[j]) { dl = i; break; }
    CFDP_ASSERT(dl >= 0);
    int nown_d = info->own_offset[dl + 1] - info->own_offset[dl];
    CFDP_ASSERT(want_idx[j] >= 0 && want_idx[j] < nown_d);
    out_cd->sendindex[s][j] = info->own_offset[dl] + want_idx[j];
  }
  /* a partner that only receives from us still has to be in the partner list */
  int known = 0;
  for (int i = 0; i < out_cd->ncommdomains; i++) known |= (out_cd->commpartner[i] == s);
  if (!known && count > 0) {
    int n = out_cd->ncommdomains;
    out_cd->commpartner = realloc(out_cd->commpartner, (size_t)(n + 1) * sizeof(int));
    int i = n;
    while (i > 0 && out_cd->commpartner[i - 1] > s) { out_cd->commpartner[i] = out_cd->commpartner[i - 1]; i--; }
    out_cd->commpartner[i] = s;
    out_cd->ncommdomains = n + 1;
  }
  return 0;
}

void cfdp_merge_link_group(int G, comm_data **cds, cfdp_merge_info **infos) {
  for (int s = 0; s < G; s++) {           /* s requests ...            */
    const cfdp_merge_info *is = infos[s];
    for (int i = 0; i < is->npartners; i++) {
      int r = is->partner[i];             /* ... points owned by r     */
      int off = is->want_off[i], cnt = is->want_off[i + 1] - off;
      cfdp_merge_set_send(cds[r], infos[r], s, cnt, is->ghost_domain + off, is->ghost_idx + off);
    }
  }
}

void cfdp_merge_scatter(const cfdp_merge_info *info, int dl, int npoints_d, int rowlen,
                        const double *merged, double *out) {
  const int *m = info->local2merged[dl];
  for (int i = 0; i < npoints_d; i++)
    memcpy(out + (size_t)i * rowlen, merged + (size_t)m[i] * rowlen, (size_t)rowlen * sizeof(double));
}

void cfdp_merge_info_free(cfdp_merge_info *info) {
  if (!info) return;
  for (int dl = 0; dl < info->ndom_local; dl++) free(info->local2merged[dl]);
  free(info->local2merged);
  free(info->domain_ids); free(info->own_offset);
  free(info->ghost_domain); free(info->ghost_idx);
  free(info->partner); free(info->want_off);
  free(info);
}
