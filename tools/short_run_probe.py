"""K = 20 iterations behind a sync, in the sustained clock state (the driver's flags): ONE graph of 20 passes against 20 + 2
launches from the stream -- wall time per step (what bench.py reports) and device time (HIP events).  The graph wins by
0.6 us/step of wall time; what a short run pays over the pass itself is launch + sync latency, 1.3 us/step
(profiles/r05_short_run_graph_vs_stream.log)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
pkg = load_package()
from cfd_proxy_amd import multigpu as mg
gp = pkg.gen_params(64, ndomains=12)
part, _ = mg.build_rank_partition(gp, 12, 1, 0, via_files=False)
g = pkg.GpuPartition(part)
g.set_fusion(True)
K = 20
def once(ug):
    g.run_iterations(5000, True, 0, use_graph=True)
    g.prepare_iterations(K, True, 0)
    g.refresh_graphs()
    g.sync(); torch.cuda.synchronize()
    t = time.perf_counter()
    ms_dev = g.run_iterations(K, True, 0, use_graph=ug)
    g.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t) / K * 1e6, ms_dev / K * 1e3
for rep in range(3):
    for ug in (True, False):
        w = sorted(once(ug) for _ in range(7))
        print(f"use_graph={ug}: wall us/step min {w[0][0]:.2f} median {w[3][0]:.2f}; device {w[3][1]:.2f}", flush=True)
g.close()
