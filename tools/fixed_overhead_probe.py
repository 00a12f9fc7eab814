"""development helper: what a short timed run (the driver's --steps 20 --warmup 5) pays.  Device time per iteration
of run_iterations(20) after (A) a graph capture + sync (the GPU idles for the milliseconds the capture takes),
(B) 200 iterations of work, (C) 5 iterations of work; and of longer runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
m = load_package()
gp = m.gen_params(64, ndomains=1); dom = m.gen_domain(gp, 0); m.fill_var(dom, None, m.VAR_HASH)
part = m.GpuPartition(dom); part.set_fusion(True)
part.run_iterations(200)
part.prepare_iterations(20); part.prepare_iterations(5)
for rep in range(3):
    part.prepare_iterations(20); part.sync(); time.sleep(0.005)
    a = part.run_iterations(20) * 1e3 / 20
    part.run_iterations(200); b = part.run_iterations(20) * 1e3 / 20
    part.sync(); time.sleep(0.005)
    part.run_iterations(5); c = part.run_iterations(20) * 1e3 / 20
    part.sync(); time.sleep(0.005)
    part.run_iterations(5); part.sync(); c2 = part.run_iterations(20) * 1e3 / 20
    print(f"K=20 device us/it: after idle {a:.2f}   right after 200 iterations {b:.2f}   right after 5 iterations {c:.2f}   after 5 iterations + sync {c2:.2f}", flush=True)
for K in (50, 100, 1000):
    part.prepare_iterations(K); part.sync()
    print(f"K={K}: {part.run_iterations(K) * 1e3 / K:.2f} us/it", flush=True)
part.close()
