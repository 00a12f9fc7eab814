"""development helper: kernel times on a genuinely unstructured mesh (Delaunay edges of random points, ~15.5 incidences
per point, degrees up to ~60) of the size of the level-2 stand-in -- which kernels the capacities select, how fast.
CONFIGS = ';'-separated lists of NAME=value environment settings applied before a plan is built (tiler experiments)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from __graft_entry__ import load_package
from unstructured import delaunay_mesh
m = load_package()
n = int(os.environ.get("NPTS", "262144"))
if os.environ.get("LATTICE") or os.environ.get("IRREGULAR"):  # the lattice stand-in of that size instead (regression check of a
    # tiler change), or the generator's irregular option (the mesh of bench.py's irregular_mesh block)
    nx = int(os.environ.get("LATTICE") or os.environ["IRREGULAR"])
    dom = m.gen_domain(m.gen_params(nx, ndomains=1, connectivity=m.CONN_IRREGULAR if os.environ.get("IRREGULAR") else 7,
                                    numbering=1 if os.environ.get("IRREGULAR") else 0), 0)
    n = dom.nown
    m.fill_var(dom, None, m.VAR_HASH)
    nfaces = dom.nfaces
    print("lattice", nx, "^3:", n, "points", nfaces, "faces", flush=True)
else:
    t = time.time(); xyz, fp, fn, vol, var = delaunay_mesh(n); print("mesh", n, "points", len(fp), "faces", time.time() - t, "s", flush=True)
    dom = m.domain_from_arrays(fp, fn, vol, n, var=var)
    nfaces = len(fp)
bg = m.algo_bytes_grad(nfaces, n, 0); bf = m.algo_bytes_flux(nfaces, n, 0)
tps = [int(x) for x in os.environ.get("TPS", "64").split(",")]
for cfg in os.environ.get("CONFIGS", "").split(";"):
    sets = [kv.split("=", 1) for kv in cfg.split() if "=" in kv]
    for k, v in sets:
        os.environ[k] = v
    for tp in tps:
        part = m.GpuPartition(dom, tile_points=tp)
        st = part.stats
        g, f = part.time_kernels(50); g, f = part.time_kernels(200)
        part.set_fusion(True)
        fu = part.time_fused(50); fu = part.time_fused(400)
        try:
            mv = part.time_fused_movement(200)
        except Exception as e:  # no movement-only form at this capacity
            mv = float("nan")
        print(f"[{cfg.strip() or 'default'}] tp {tp}: tiles {st['ntiles']} dup {st['nfaces_dup']/st['nfaces_used']:.3f} halo/tile {st['nhalo']/st['ntiles']:.1f} "
              f"lds_grad {st['lds_grad']} grad {g*1e3:.1f} us ({bg/g/1e6/8000:.3f})  flux {f*1e3:.1f} us ({bf/f/1e6/8000:.3f})  "
              f"fused {fu*1e3:.1f} us ({(bg+bf)/fu/1e6/8000:.3f} of 8 TB/s, 8d bytes)  movement {mv*1e3:.1f} us", flush=True)
        part.close()
    for k, _ in sets:
        os.environ.pop(k, None)
