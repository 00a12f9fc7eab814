"""development helper: wall time vs device time of run_iterations(K) for small K (what a --steps 20 bench run sees)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
m = load_package()
gp = m.gen_params(64, ndomains=1); dom = m.gen_domain(gp, 0); m.fill_var(dom, None, m.VAR_HASH)
part = m.GpuPartition(dom); part.set_fusion(True)
part.run_iterations(200)
for K in (20, 21, 50, 100, 1000):
    for rep in range(3):
        part.prepare_iterations(K)
        part.sync()
        t = time.perf_counter()
        ms = part.run_iterations(K)
        t1 = time.perf_counter()
        part.sync()
        t2 = time.perf_counter()
        print(f"K {K:5d} device {ms*1e3/K:7.2f} us/it  call {1e6*(t1-t)/K:7.2f} us/it  call+sync {1e6*(t2-t)/K:7.2f} us/it", flush=True)
part.close()
