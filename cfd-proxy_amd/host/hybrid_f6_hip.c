/*
 * hybrid_f6_hip.c -- the driver, with the reference's command line
 * (reference src/hybrid.f6.c:27-101):
 *
 *     hybrid.f6.hip -lvl [1-4] GRID_PREFIX [--gpus G] [--flux-ref] [--dump FILE]
 *
 * Same call order as the reference main(): init_communication, open
 * "<prefix>_domain_<d>_lvl_<l>", read_solver_data, init_solver_data,
 * read_communication_data, compute_communication_tables, init_threads, test_solver,
 * free_communication_ressources.  Difference: the reference starts one MPI rank per
 * domain; this process reads ALL domains, merges N/G of them per GPU rank and drives the
 * G ranks itself (peer copies over xGMI instead of MPI/GASPI messages).
 */
#include "cfdproxy_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char *argv[]) {
  if (argc < 4 || strcmp(argv[1], "-lvl") != 0) {
    printf("Usage: %s -lvl [1-4] GRID_PREFIX [--gpus G] [--flux-ref] [--var one|hash]\n", argv[0]);
    exit(EXIT_FAILURE);
  }
  const int lvl = atoi(argv[2]);
  const char *prefix = argv[3];
  int G = 0, flux_ref = 0, var_hash = 0;
  for (int i = 4; i < argc; i++) {
    if (!strcmp(argv[i], "--gpus") && i + 1 < argc) G = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--flux-ref")) flux_ref = 1;
    else if (!strcmp(argv[i], "--var") && i + 1 < argc) var_hash = !strcmp(argv[++i], "hash");
  }
  /* number of domains: every file carries it */
  char fname[4096];
  snprintf(fname, sizeof fname, "%s_domain_%d_lvl_%d", prefix, 0, lvl);
  int ncid = cfdp_nc_open(fname);
  const int N = get_nc_val(ncid, "ndomains");
  cfdp_nc_close(ncid);
  int ndev = cfdp_gpu_device_count();
  if (ndev <= 0) { fprintf(stderr, "Error: no HIP device\n"); exit(EXIT_FAILURE); }
  if (G <= 0) G = ndev < N ? ndev : N;
  if (G > N) G = N;

  solver_data *sd = calloc((size_t)G, sizeof(solver_data));
  comm_data *cd = calloc((size_t)G, sizeof(comm_data));
  solver_data **sdp = calloc((size_t)G, sizeof(void *));
  comm_data **cdp = calloc((size_t)G, sizeof(void *));
  cfdp_merge_info **infos = calloc((size_t)G, sizeof(void *));
  for (int r = 0; r < G; r++) {
    int first, count;
    cfdp_rank_domains(r, N, G, &first, &count);
    solver_data *ds = calloc((size_t)count, sizeof(solver_data));
    comm_data *dc = calloc((size_t)count, sizeof(comm_data));
    int *ids = malloc((size_t)count * sizeof(int));
    for (int i = 0; i < count; i++) {
      ids[i] = first + i;
      init_communication(argc, argv, &dc[i]);
      cfdp_load_domain(prefix, ids[i], lvl, &ds[i], &dc[i]);
    }
    cfdp_merge_domains(count, ids, ds, dc, N, G, r, &sd[r], &cd[r], &infos[r]);
    for (int i = 0; i < count; i++) { cfdp_free_solver_data(&ds[i]); cfdp_free_comm_data(&dc[i]); }
    free(ds); free(dc); free(ids);
    sdp[r] = &sd[r];
    cdp[r] = &cd[r];
    printf("rank %d: %d domains, %d own + %d ghost points, %d faces\n", r, count,
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
  test_solver(&cd[0], &sd[0], 0);
  for (int r = 0; r < G; r++) free_communication_ressources(&cd[r]);
  cfdp_group_destroy(grp);
  printf("*** SUCCESS\n");
  return 0;
}
