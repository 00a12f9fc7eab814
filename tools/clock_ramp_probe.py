"""K = 20 timed iterations behind a barrier + sync (the driver's flags): how does the time per step depend on the
continuous load the GPU has seen right before the sync (W warm-up steps), and on the idle time between the sync and the
timed launch?  Separates 'the chip has not reached its sustained clock' from 'a short graph costs more per step'."""
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
g.prepare_iterations(K, True, 0)
def once(warm, idle_us):
    if warm:
        g.run_iterations(warm, True, 0, use_graph=True)
    g.prepare_iterations(K, True, 0)  # as bench.py does: the warm-up's graph took the slot (a no-op when it did not)
    g.sync(); torch.cuda.synchronize()
    if idle_us:
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e6 < idle_us:
            pass
    t = time.perf_counter()
    ms_dev = g.run_iterations(K, True, 0, use_graph=True)
    g.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t) / K * 1e6, ms_dev / K * 1e3
print("warm-up steps | idle us | wall us/step (median of 5) | device us/step")
for warm in (5, 50, 500, 5000, 50000):
    for idle in (0, 200, 5000):
        w, d = [], []
        for rep in range(5):
            time.sleep(0.2)
            a, b = once(warm, idle)
            w.append(a); d.append(b)
        w.sort(); d.sort()
        print(f"{warm:6d} | {idle:5d} | {w[2]:6.2f} | {d[2]:6.2f}", flush=True)
# the same 20 steps as the tail of a long run: K steps timed by events inside continuous load
ms = g.run_iterations(20000, True, 0, use_graph=True)
print(f"long run: {ms / 20000 * 1e3:6.2f} us/step", flush=True)
g.close()
