/*
 * hybrid_f6_hip.c -- the driver, with the reference's command line
 * (reference src/hybrid.f6.c:27-101):
 *
 *     hybrid.f6.hip -lvl [1-4] GRID_PREFIX [--gpus G] [--flux-ref] [--var one|hash] [--cluster]
 *     hybrid.f6.hip -vcycle LMAX GRID_PREFIX [--gpus G] [--sweeps S] [--cycles C]   (extension)
 *
 * Same call order as the reference main(): init_communication, open
 * "<prefix>_domain_<d>_lvl_<l>", read_solver_data, init_solver_data,
 * read_communication_data, compute_communication_tables, init_threads, test_solver,
 * free_communication_ressources.  Difference: the reference starts one MPI rank per
 * domain; this process reads ALL domains, merges N/G of them per GPU rank and drives the
 * G ranks itself (peer copies over xGMI instead of MPI/GASPI messages).
 * -vcycle: the "3V multigrid cycle" of the published plots (documentation/CFD-Proxy.pdf p.3):
 * levels 1..LMAX are loaded side by side and S (default 3) iterations run on every level,
 * finest to coarsest and back (cfdp_test_vcycle).
 */
#include "cfdproxy_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct level {
  int G;
  solver_data *sd;
  comm_data *cd;
  cfdp_group *grp;
} level;

/* the reference main()'s sequence for one -lvl level, for all G in-process ranks */
static level load_level(int argc, char *argv[], const char *prefix, int lvl, int G_req, int flux_ref,
                        int var_hash, int cluster) {
  level L;
  char fname[4096];
  snprintf(fname, sizeof fname, "%s_domain_%d_lvl_%d", prefix, 0, lvl);
  int ncid = cfdp_nc_open(fname);
  const int N = get_nc_val(ncid, "ndomains"); /* number of domains: every file carries it */
  cfdp_nc_close(ncid);
  int ndev = cfdp_gpu_device_count();
  if (ndev <= 0) { fprintf(stderr, "Error: no HIP device\n"); exit(EXIT_FAILURE); }
  int G = G_req;
  if (G <= 0) G = ndev < N ? ndev : N;
  if (G > N) G = N;

  if (cluster && G > 1) {
    /* domain ids need not be spatially coherent: group the domains of a rank along the
     * commpartner graph of the files instead of by blocks of ids */
    int *xadj, *adj, *wgt, *map = malloc((size_t)N * sizeof(int));
    cfdp_domain_graph(prefix, lvl, N, &xadj, &adj, &wgt);
    long cut = cfdp_cluster_domains(N, G, xadj, adj, wgt, map);
    cfdp_set_domain_map(map, N, G);
    printf("level %d: %d domains clustered onto %d ranks, %ld halo points between ranks\n", lvl, N, G, cut);
    free(xadj); free(adj); free(wgt); free(map);
  } else {
    cfdp_set_domain_map(NULL, 0, 0);
  }
  solver_data *sd = calloc((size_t)G, sizeof(solver_data));
  comm_data *cd = calloc((size_t)G, sizeof(comm_data));
  solver_data **sdp = calloc((size_t)G, sizeof(void *));
  comm_data **cdp = calloc((size_t)G, sizeof(void *));
  cfdp_merge_info **infos = calloc((size_t)G, sizeof(void *));
  for (int r = 0; r < G; r++) {
    int *ids = malloc((size_t)N * sizeof(int));
    const int count = cfdp_rank_domain_list(r, N, G, ids);
    solver_data *ds = calloc((size_t)count, sizeof(solver_data));
    comm_data *dc = calloc((size_t)count, sizeof(comm_data));
    for (int i = 0; i < count; i++) {
      init_communication(argc, argv, &dc[i]);
      cfdp_load_domain(prefix, ids[i], lvl, &ds[i], &dc[i]);
    }
    cfdp_merge_domains(count, ids, ds, dc, N, G, r, &sd[r], &cd[r], &infos[r]);
    for (int i = 0; i < count; i++) { cfdp_free_solver_data(&ds[i]); cfdp_free_comm_data(&dc[i]); }
    free(ds); free(dc); free(ids);
    sdp[r] = &sd[r];
    cdp[r] = &cd[r];
    printf("level %d rank %d: %d domains, %d own + %d ghost points, %d faces\n", lvl, r, count,
           sd[r].nownpoints, sd[r].nallpoints - sd[r].nownpoints, sd[r].nfaces);
  }
  cfdp_merge_link_group(G, cdp, infos);
  cfdp_group *grp = cfdp_group_create(G, sdp, cdp);
  if (flux_ref) cfdp_group_set_flux_mode(grp, CFDP_FLUX_REFERENCE);
  for (int r = 0; r < G; r++) {
    if (var_hash) cfdp_fill_var(sd[r].var, NULL, sd[r].nallpoints, CFDP_VAR_HASH, 1, 1, 1);
    compute_communication_tables(&cd[r]);
    init_threads(&cd[r], &sd[r], 0);
  }
  free(sdp); free(cdp); free(infos);
  L.G = G; L.sd = sd; L.cd = cd; L.grp = grp;
  return L;
}

static void free_level(level *L) {
  for (int r = 0; r < L->G; r++) free_communication_ressources(&L->cd[r]);
  cfdp_group_destroy(L->grp);
}

int main(int argc, char *argv[]) {
  const int vcycle = argc >= 2 && !strcmp(argv[1], "-vcycle");
  if (argc < 4 || (strcmp(argv[1], "-lvl") != 0 && !vcycle)) {
    printf("Usage: %s -lvl [1-4] GRID_PREFIX [--gpus G] [--flux-ref] [--var one|hash] [--cluster]\n"
           "       %s -vcycle LMAX GRID_PREFIX [--gpus G] [--sweeps S] [--cycles C]\n", argv[0], argv[0]);
    exit(EXIT_FAILURE);
  }
  const int lvl = atoi(argv[2]);
  const char *prefix = argv[3];
  int G = 0, flux_ref = 0, var_hash = 0, sweeps = 3, cycles = 10, cluster = 0;
  for (int i = 4; i < argc; i++) {
    if (!strcmp(argv[i], "--gpus") && i + 1 < argc) G = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--flux-ref")) flux_ref = 1;
    else if (!strcmp(argv[i], "--var") && i + 1 < argc) var_hash = !strcmp(argv[++i], "hash");
    else if (!strcmp(argv[i], "--cluster")) cluster = 1;
    else if (!strcmp(argv[i], "--sweeps") && i + 1 < argc) sweeps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--cycles") && i + 1 < argc) cycles = atoi(argv[++i]);
  }
  if (!vcycle) {
    level L = load_level(argc, argv, prefix, lvl, G, flux_ref, var_hash, cluster);
    test_solver(&L.cd[0], &L.sd[0], 0);
    free_level(&L);
  } else {
    if (lvl < 1 || lvl > 16 || sweeps < 1 || cycles < 1) { fprintf(stderr, "Error: bad -vcycle arguments\n"); exit(EXIT_FAILURE); }
    level *Ls = calloc((size_t)lvl, sizeof(level));
    cfdp_group **groups = calloc((size_t)lvl, sizeof(void *));
    for (int l = 0; l < lvl; l++) {
      Ls[l] = load_level(argc, argv, prefix, l + 1, G, flux_ref, var_hash, cluster);
      groups[l] = Ls[l].grp;
    }
    cfdp_test_vcycle(lvl, groups, sweeps, cycles);
    for (int l = 0; l < lvl; l++) free_level(&Ls[l]);
    free(groups); free(Ls);
  }
  printf("*** SUCCESS\n");
  return 0;
}
