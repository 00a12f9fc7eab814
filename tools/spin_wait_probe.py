"""How the HOST waits for the device, priced on the driver's K = 20 run (one graph of 20 passes behind a sync, sustained clock
state): HIP's default wait against hipSetDeviceFlags(hipDeviceScheduleSpin / Yield / BlockingSync), and -- when started with
HSA_ENABLE_INTERRUPT=0 in the environment -- the runtime polling its signals instead of sleeping on an interrupt.
Wall time per step (what bench.py reports) beside device time (HIP events): the difference is launch + sync latency."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
pkg = load_package()
from cfd_proxy_amd import multigpu as mg
gp = pkg.gen_params(64, ndomains=12)
part, _ = mg.build_rank_partition(gp, 12, 1, 0, via_files=False)
g = pkg.GpuPartition(part)
g.set_fusion(True)
K = int(os.environ.get("K", "20"))
hip = ctypes.CDLL("libamdhip64.so")
def once():
    g.run_iterations(5000, True, 0, use_graph=True)
    g.prepare_iterations(K, True, 0)
    g.refresh_graphs()
    g.sync(); torch.cuda.synchronize()
    t = time.perf_counter()
    ms_dev = g.run_iterations(K, True, 0, use_graph=True)
    g.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t) / K * 1e6, ms_dev / K * 1e3
print("HSA_ENABLE_INTERRUPT =", os.environ.get("HSA_ENABLE_INTERRUPT"), flush=True)
FLAGS = {"auto": 0, "spin": 1, "yield": 2, "blocking": 4}
for rep in range(2):
    for name in os.environ.get("MODES", "auto,spin,yield,blocking,auto,spin").split(","):
        rc = hip.hipSetDeviceFlags(ctypes.c_uint(FLAGS[name]))
        w = sorted(once() for _ in range(7))
        print(f"{name:9s} rc={rc}: wall us/step min {w[0][0]:.2f} median {w[3][0]:.2f} max {w[-1][0]:.2f}; device median {w[3][1]:.2f}", flush=True)
g.close()
