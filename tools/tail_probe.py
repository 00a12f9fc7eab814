"""fused pass time versus tile count (64 x 64 x nz lattices): the fixed cost of a launch -- its first round
runs in lockstep -- and the steady-state cost per tile."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
m = load_package()
for nz in (16, 32, 48, 56, 60, 64, 68, 72, 80, 96, 128):
    gp = m.gen_params(64, 64, nz, ndomains=1); dom = m.gen_domain(gp, 0); m.fill_var(dom, None, m.VAR_HASH)
    part = m.GpuPartition(dom); part.set_fusion(True)
    part.time_fused(50)
    us = min(part.time_fused(300) for _ in range(3)) * 1e3
    nt = (part.counts()["nown"] + 63) // 64
    print(nz, "tiles", nt, "fused us %.2f" % us, "ns/tile %.2f" % (us * 1e3 / max(nt, 1)), flush=True)
    part.close(); dom.free()
