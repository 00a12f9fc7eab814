"""development helper: host vs device time of the plan's two heavy stages"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
m = load_package()
for n in (64, 128):
    gp = m.gen_params(n, ndomains=1); dom = m.gen_domain(gp, 0)
    for which in (0, 3, 3):
        t = time.time(); p = m.Plan(dom, device_stages=which); dt = time.time() - t
        print(f"n {n} device_stages {which}: plan {dt:.3f} s  stage seconds (csr, blobs) {p.stage_seconds}", flush=True)
        p.free()
    dom.free()
