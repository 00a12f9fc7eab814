"""development helper: where does the fixed cost of a timed region go? (N=1 workload)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
pkg = load_package()
from cfd_proxy_amd import multigpu as mg
gp = pkg.gen_params(64, ndomains=12)
part, st = mg.build_rank_partition(gp, 12, 1, 0, via_files=False)
solver = mg.RankSolver(part, 0, 1, 0, None)
g = solver.gpu
g.run_iterations(300)
for steps in (51, 101, 501, 2001):
    for rep in range(3):
        solver.synchronize(); solver.synchronize()
        t0 = time.perf_counter()
        ms_dev = g.run_iterations(steps)
        t1 = time.perf_counter()
        solver.synchronize()
        t2 = time.perf_counter()
        print("steps %5d: device events %.1f us/step | host call %.1f us/step | incl. sync %.1f us/step | fixed part %.0f us" % (
            steps, ms_dev * 1e3 / steps, (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6, (t2 - t0) * 1e6 - steps * 37.9), flush=True)
solver.close()
