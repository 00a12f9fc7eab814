"""The distribution of bench.py's 20-step timed region (one partition): the same sequence -- 0.25 s of conditioning, graphs
instantiated again, sync, K steps, sync -- repeated REPS times in one process, for several ways of preparing the region
(MODES): refresh = bench.py's; primer = refresh, then the timed graph launched once (untimed) and a sync; none = no refresh;
noevents = primer, and the timed run without the HIP event pair around it (cfdp_gpu_run_iterations with ms_total = NULL).
(A build with hipGraphUpload behind every instantiation was measured with the refresh mode too: worse, not kept.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
from cfd_proxy_amd import multigpu as mg
gp = pkg.gen_params(64, ndomains=12)
part, _ = mg.build_rank_partition(gp, 12, 1, 0, via_files=False)
g = pkg.GpuPartition(part)
g.set_fusion(True)
K, REPS = int(os.environ.get("K", "20")), int(os.environ.get("REPS", "24"))
COND = int(os.environ.get("COND", "6400"))
def region(mode):
    g.prepare_iterations(K, True, 0)
    g.run_iterations(COND + 1, True, 0, use_graph=True)
    if mode != "none":
        g.refresh_graphs()
    if mode in ("primer", "noevents"):
        g.run_iterations(K, True, 0, use_graph=True)
    g.sync(); torch.cuda.synchronize()
    t = time.perf_counter()
    g.run_iterations(K, True, 0, use_graph=True, device_time=mode != "noevents")
    g.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t) / K * 1e6
if os.environ.get("BREAKDOWN"):  # where the region's wall time goes: the call (launch + stream sync), the context's sync, torch's
    rows = []
    for _ in range(REPS):
        g.prepare_iterations(K, True, 0)
        g.run_iterations(COND + 1, True, 0, use_graph=True)
        g.refresh_graphs()
        g.run_iterations(K, True, 0, use_graph=True, device_time=False)
        g.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.run_iterations(K, True, 0, use_graph=True, device_time=False)
        t1 = time.perf_counter()
        g.sync()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        rows.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6))
    a = np.median(np.array(rows), axis=0)
    print(f"median us per region of {K} steps: run_iterations (launch + stream sync) {a[0]:.1f}, context sync {a[1]:.1f}, torch.cuda.synchronize {a[2]:.1f}", flush=True)
    g.close()
    sys.exit(0)
for mode in os.environ.get("MODES", "refresh,primer,none,refresh,primer").split(","):
    w = np.array([region(mode) for _ in range(REPS)])
    print(f"{mode:8s}: us/step min {w.min():.2f} p25 {np.percentile(w,25):.2f} median {np.median(w):.2f} p75 {np.percentile(w,75):.2f} max {w.max():.2f}; "
          f"above 38: {int((w > 38).sum())} of {REPS};  " + " ".join(f"{x:.1f}" for x in w), flush=True)
g.close()
