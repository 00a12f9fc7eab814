"""GPU exploration (development helper, not part of the product): smoke(), then kernel timings over
tile size x lanes-per-point on the level-2 and level-1 stand-in meshes.  Parity is the tests' job."""
import sys, time, json, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package, smoke
m = load_package()
out = open(os.environ.get("EXPLORE_OUT", "gpurun_out/explore.log"), "a")
def log(*a):
    s = " ".join(str(x) for x in a); print(s, flush=True); out.write(s + "\n"); out.flush()
if not os.environ.get("CFDP_DEBUG_ABLATE"): smoke()
sizes = [int(x) for x in os.environ.get("SIZES", "64,128").split(",")]
tps = [int(x) for x in os.environ.get("TPS", "64,128,256").split(",")]
lanes = [int(x) for x in os.environ.get("LANES", "1,2,4,8").split(",")]
for n in sizes:
    gp = m.gen_params(n, ndomains=1); dom = m.gen_domain(gp, 0)
    m.fill_var(dom, None, m.VAR_HASH)
    var = dom.var.copy()
    bg = m.algo_bytes_grad(dom.nfaces, dom.nown, 0); bf = m.algo_bytes_flux(dom.nfaces, dom.nown, 0)
    for tp in tps:
        for L in lanes:
            if tp * abs(L) > 1024: continue
            t0 = time.time()
            try:
                part = m.GpuPartition(dom, tile_points=tp, grad_lanes=L, flux_lanes=8)
            except Exception as e:
                log("n", n, "tp", tp, "L", L, "FAILED", e); continue
            t_up = time.time() - t0
            iters = 50 if n <= 64 else 20
            for pipe in [int(x) for x in os.environ.get("PIPES", "0,1,2,4").split(",")]:
                part.set_pipeline(pipe)
                mg, mf = part.time_kernels(iters)
                mg, mf = part.time_kernels(iters)
                err = ""
                log("n", n, "tp", tp, "L", L, "pipe", pipe, "grad %.1f us %.0f GB/s (%.1f%% of 8TB/s)" % (mg * 1e3, bg / mg / 1e6, bg / mg / 1e6 / 80),
                    "flux %.1f us %.0f GB/s" % (mf * 1e3, bf / mf / 1e6), "lds", part.stats["lds_grad"], err)
            if L == 4 and not os.environ.get("NO_FUSED"):
                part.set_pipeline(0)
                part.set_fusion(True)
                mfu = part.time_fused(iters); mfu = part.time_fused(iters)
                it = 100
                part.run_iterations(it); t_f = part.run_iterations(it) / it
                part.set_fusion(False)
                part.run_iterations(it); t_s = part.run_iterations(it) / it
                log("n", n, "tp", tp, "fused pass %.1f us (%.0f GB/s algorithmic = %.1f%% of 8TB/s)" % (mfu * 1e3, (bg + bf) / mfu / 1e6, (bg + bf) / mfu / 1e6 / 80),
                    "iteration fused %.1f us, separate %.1f us" % (t_f * 1e3, t_s * 1e3))
            part.close()
    dom.free()
