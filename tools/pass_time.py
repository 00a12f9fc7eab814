"""time of the fused pass (event-timed, 7 x 200 passes: min / median / max), its movement-only floor and the two separate
kernels, 64^3 and 128^3 -- the quick A/B of a kernel change.  python tools/pass_time.py [64 128]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
m = load_package()
for n in [int(a) for a in sys.argv[1:]] or [64, 128]:
    dom = m.gen_domain(m.gen_params(n, ndomains=1), 0); m.fill_var(dom, None, m.VAR_HASH)
    part = m.GpuPartition(dom); part.set_fusion(True)
    ts = sorted(part.time_fused(200) for _ in range(7))
    tg, tf = part.time_kernels(200, m.FLUX_CONSISTENT)
    print(f"n {n}: fused pass {ts[0]*1e3:.2f} / {ts[3]*1e3:.2f} / {ts[6]*1e3:.2f} us; gradient kernel {tg*1e3:.2f} us, flux kernel {tf*1e3:.2f} us", flush=True)
    part.close()
