"""K = 20 timed iterations (the driver's flags): does it matter whether the graphs of the timed run are captured before or
after the W = 5 warm-up iterations?  (The capture + instantiate idles the GPU for milliseconds right in front of the
timed region when it comes last.)"""
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
K, W = 20, 5
def once(order):
    if order == "capture last":
        g.run_iterations(W, True, 0, use_graph=True)
        g.prepare_iterations(K, True, 0)
    else:
        g.prepare_iterations(K, True, 0)
        g.run_iterations(W, True, 0, use_graph=False)
    g.sync(); torch.cuda.synchronize()
    t = time.perf_counter()
    ms_dev = g.run_iterations(K, True, 0, use_graph=True)
    g.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t) / K * 1e6, ms_dev / K * 1e3
for rep in range(3):
    for order in ("capture last", "capture first"):
        time.sleep(0.5)  # the GPU idles between measurements, as it does after the set-up of a bench run
        wall, dev = once(order)
        print(f"{order:14s}: wall {wall:6.2f} us/step, device {dev:6.2f} us/step", flush=True)
g.close()
