"""the fused pass with workgroups that stay (CFDP_FUSED_PERSIST=1) against the one-tile-per-workgroup form: values
(bit-identical gradients and flux after K iterations) and time per pass, 64^3 and 128^3"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
m = load_package()
for n in [int(a) for a in sys.argv[1:]] or [64, 128]:
    dom = m.gen_domain(m.gen_params(n, ndomains=1), 0); m.fill_var(dom, None, m.VAR_HASH)
    res = {}
    for persist in (0, 1):
        os.environ["CFDP_FUSED_PERSIST"] = str(persist)
        part = m.GpuPartition(dom); part.set_fusion(True)
        dom.grad[:] = 0.0; dom.psd_flux[:] = 0.0; part.push_fields()
        part.run_iterations(7, True, m.FLUX_CONSISTENT, use_graph=True)
        part.pull_fields()
        g, f = dom.grad[: dom.nown].copy(), dom.psd_flux[: dom.nown].copy()
        ts = sorted(part.time_fused(200) for _ in range(5))
        long = part.run_iterations(4001, True, m.FLUX_CONSISTENT, use_graph=True) / 4001 * 1e3
        res[persist] = (g, f, ts[2] * 1e3, long)
        part.close()
        print(f"n {n} persist {persist}: fused pass {ts[0]*1e3:.2f} / {ts[2]*1e3:.2f} / {ts[4]*1e3:.2f} us (min/med/max of 5 x 200), long run {long:.2f} us/iteration", flush=True)
    same = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    print(f"n {n}: values identical: {same}; finite: {np.isfinite(res[1][0]).all() and np.isfinite(res[1][1]).all()}; |grad| max {np.abs(res[1][0]).max():.3e}", flush=True)
    assert same
