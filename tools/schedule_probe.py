"""development helper: cost of the step schedules themselves (no exchange) on rank 0 of the 2- and
8-rank decompositions: all tiles in one launch / boundary tiles before the interior tiles on one
stream (bulk) / boundary tiles beside the interior tiles on two streams; hipGraph vs stream launches"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
from cfd_proxy_amd import multigpu as mg
for world in (2, 8):
    dims, nd = mg.bench_mesh(world)
    gp = pkg.gen_params(*dims, ndomains=nd)
    parts = [mg.build_rank_partition(gp, nd, world, r, via_files=False)[0] for r in range(world)]
    reqs = [{int(k): (v[0], v[1]) for k, v in pkg.merge_requests(p).items()} for p in parts]
    mg.exchange_requests(parts[0], 0, world, None, all_requests=reqs)
    g = pkg.GpuPartition(parts[0]); g.set_fusion(True)
    for graph in (True, False):
        for name, ex, ov in (("no exchange (one launch)", False, False), ("bulk (all tiles, then pack)", True, False), ("overlap (two streams)", True, True)):
            g.time_schedule(100, ex, ov, graph)
            us = g.time_schedule(200, ex, ov, graph) * 1e3
            print("world %d %-6s %-30s %.1f us/step" % (world, "graph" if graph else "stream", name, us), flush=True)
    g.close()
