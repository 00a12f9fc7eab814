"""3V-cycle timing on one GPU (development helper): levels n^3, (n/2)^3, ... ; hipGraph vs stream."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
m = load_package()
out = {}
for top, nlev in ((128, 4), (64, 3)):
    doms, parts = [], []
    for l in range(nlev):
        n = top >> l
        dom = m.gen_domain(m.gen_params(n, ndomains=1), 0); m.fill_var(dom, None, m.VAR_HASH)
        part = m.GpuPartition(dom); part.set_fusion(True)
        doms.append(dom); parts.append(part)
    res = {}
    for graph in (True, False):
        m.vcycle(parts, 3, 5, use_graph=graph)
        res["hipgraph" if graph else "stream"] = m.vcycle(parts, 3, 50, use_graph=graph)
    per_level = []
    for part in parts:
        part.run_iterations(25); per_level.append(part.run_iterations(100) / 100)
    res["levels"] = [top >> l for l in range(nlev)]
    res["iteration_ms_per_level"] = per_level
    res["sum_of_levels_ms"] = sum(3 * t * (1 if l == nlev - 1 else 2) for l, t in enumerate(per_level))
    out[f"{top}^3 x {nlev} levels"] = res
    for p in parts: p.close()
    for d in doms: d.free()
print(json.dumps(out, indent=1))
