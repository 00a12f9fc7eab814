"""profile helper: run the gradient (and flux) kernel a few times on one mesh (for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
m = load_package()
n = int(os.environ.get("N", "128")); tp = int(os.environ.get("TP", "128")); L = int(os.environ.get("L", "4"))
iters = int(os.environ.get("ITERS", "5"))
irr = os.environ.get("IRREGULAR", "0") != "0"  # the generator's irregular option (random tetrahedralisation + hubs, scrambled numbering)
gp = m.gen_params(n, ndomains=1, connectivity=m.CONN_IRREGULAR if irr else 7, numbering=1 if irr else 0)
dom = m.gen_domain(gp, 0); m.fill_var(dom, None, m.VAR_HASH)
part = m.GpuPartition(dom, tile_points=tp, grad_lanes=L, flux_lanes=8)
g, f = part.time_kernels(iters)
print("irregular" if irr else "lattice", "n", n, "tp", tp, "L", L, "grad us", g * 1e3, "flux us", f * 1e3, flush=True)
if os.environ.get("FUSED", "1") != "0":
    part.set_fusion(True)
    print("fused pass us", part.time_fused(iters) * 1e3, flush=True)
part.close()
