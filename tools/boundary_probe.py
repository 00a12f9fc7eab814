"""development helper: what does the boundary/interior tile split of a multi-rank partition cost?
rank 0 of the N=2 and N=8 bench decompositions, iterations without exchange, on one GPU"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
from cfd_proxy_amd import multigpu as mg
for world in (1, 2, 8):
    dims, nd = mg.bench_mesh(world)
    gp = pkg.gen_params(*dims, ndomains=nd)
    if world == 1:
        part, st = mg.build_rank_partition(gp, nd, 1, 0, via_files=False)
    else:
        # rank 0 and its send lists: requests of all ranks are needed -> build every rank's merge (host only)
        parts = [mg.build_rank_partition(gp, nd, world, r, via_files=False)[0] for r in range(world)]
        reqs = [{int(k): (v[0], v[1]) for k, v in pkg.merge_requests(p).items()} for p in parts]
        mg.exchange_requests(parts[0], 0, world, None, all_requests=reqs)
        part = parts[0]
    if True:
        g = pkg.GpuPartition(part)
        g.set_fusion(True)
        g.run_iterations(50)
        ms = g.run_iterations(500) / 500
        print("world", world, "own", part.nown, "ghost", part.nall - part.nown, "partners", part.partners,
              "tiles", g.stats["ntiles"], "boundary tiles", g.stats["nbtiles"], "iteration %.1f us" % (ms * 1e3), flush=True)
        g.close()
