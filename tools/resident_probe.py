"""development helper: tile-resident iterations (one launch for K iterations, cfdp_gpu_set_resident) against the fused
one-launch-per-pass path replayed from hipGraphs, on partitions whose tiles are all co-resident: rank 0 of the
strong-scaling decompositions (iterations without exchange) and small whole meshes (coarse V-cycle levels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
from cfd_proxy_amd import multigpu as mg


def rank0(name, world):
    cfg = mg.bench_config(name, world)
    gp = pkg.gen_params(*cfg["dims"], ndomains=cfg["ndomains"])
    parts = [mg.build_rank_partition(gp, cfg["ndomains"], world, r, via_files=False)[0] for r in range(world)]
    reqs = [{int(k): (v[0], v[1]) for k, v in pkg.merge_requests(p).items()} for p in parts]
    mg.exchange_requests(parts[0], 0, world, None, all_requests=reqs)
    return parts[0]


def whole(n):
    d = pkg.gen_domain(pkg.gen_params(n, ndomains=1), 0)
    pkg.fill_var(d, None, pkg.VAR_HASH)
    return d


cases = [("dualgrid.48 rank 0 of 4", lambda: rank0("dualgrid.48", 4)), ("dualgrid.192 rank 0 of 8", lambda: rank0("dualgrid.192", 8)),
         ("32^3 whole", lambda: whole(32)), ("16^3 whole", lambda: whole(16)), ("40x40x40 whole", lambda: whole(40))]
K = int(os.environ.get("K", "1000"))
for label, make in cases:
    part = make()
    g = pkg.GpuPartition(part)
    g.set_fusion(True)
    ok, why = g.resident_qualifies()
    g.run_iterations(200)
    t_graph = min(g.run_iterations(K) for _ in range(3)) / K * 1e3
    line = f"{label:26s} own {part.nown:7d} tiles {g.stats['ntiles']:5d}  fused passes from hipGraphs {t_graph:6.2f} us/iteration"
    if ok:
        g.set_resident(1)
        g.run_iterations(200)
        t_res = min(g.run_iterations(K) for _ in range(3)) / K * 1e3
        line += f"   tile-resident {t_res:6.2f} us/iteration  ({t_graph / t_res:4.2f}x)"
    else:
        line += f"   tile-resident: does not qualify ({why})"
    print(line, flush=True)
    g.close()
