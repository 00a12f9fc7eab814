"""The driver's K = 20 run as ONE graph against a short head graph + the rest (CFDP_GRAPH_HEAD = passes in the head; read once
per process, so one process per setting: HEADS=0,1,2,0,1,2 alternates them on one box).  Wall time per step (what bench.py
reports), device time (HIP events) and a checksum of the final gradients and flux (the pair must compute what the one
graph computes, bit for bit)."""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "HEADS" in os.environ:
    heads = os.environ.pop("HEADS").split(",")
    for h in heads:
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, CFDP_GRAPH_HEAD=h), check=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
pkg = load_package()
from cfd_proxy_amd import multigpu as mg
gp = pkg.gen_params(64, ndomains=12)
part, _ = mg.build_rank_partition(gp, 12, 1, 0, via_files=False)
g = pkg.GpuPartition(part)
g.set_fusion(True)
K = int(os.environ.get("K", "20"))
def once():
    g.run_iterations(5000, True, 0, use_graph=True)
    g.prepare_iterations(K, True, 0)
    g.refresh_graphs()
    g.sync(); torch.cuda.synchronize()
    t = time.perf_counter()
    ms_dev = g.run_iterations(K, True, 0, use_graph=True)
    g.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t) / K * 1e6, ms_dev / K * 1e3
w = sorted(once() for _ in range(9))
g.pull_fields()
h = hashlib.sha256(g.dom.grad.tobytes() + g.dom.psd_flux.tobytes()).hexdigest()[:16]
print(f"CFDP_GRAPH_HEAD={os.environ.get('CFDP_GRAPH_HEAD')}: K={K} wall us/step min {w[0][0]:.2f} median {w[4][0]:.2f} max {w[-1][0]:.2f}; "
      f"device median {w[4][1]:.2f}; values {h}", flush=True)
g.close()
