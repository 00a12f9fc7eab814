"""development helper: kernel times on a genuinely unstructured mesh (Delaunay edges of random points, ~15.5 faces
per point, degrees up to ~40) of the size of the level-2 stand-in -- which kernels the capacities select, how fast"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from __graft_entry__ import load_package
from unstructured import delaunay_mesh
m = load_package()
n = int(os.environ.get("NPTS", "262144"))
t = time.time(); xyz, fp, fn, vol, var = delaunay_mesh(n); print("mesh", n, "points", len(fp), "faces", time.time() - t, "s", flush=True)
dom = m.domain_from_arrays(fp, fn, vol, n, var=var)
bg = m.algo_bytes_grad(len(fp), n, 0); bf = m.algo_bytes_flux(len(fp), n, 0)
for tp in [int(x) for x in os.environ.get("TPS", "64,48,32").split(",")]:
    part = m.GpuPartition(dom, tile_points=tp)
    st = part.stats
    g, f = part.time_kernels(50); g, f = part.time_kernels(50)
    part.set_fusion(True)
    fu = part.time_fused(50); fu = part.time_fused(50)
    print(f"tp {tp}: tiles {st['ntiles']} dup {st['nfaces_dup']/st['nfaces_used']:.3f} halo/tile {st['nhalo']/st['ntiles']:.1f} lds_grad {st['lds_grad']} "
          f"grad {g*1e3:.1f} us ({bg/g/1e6/8000:.3f})  flux {f*1e3:.1f} us  fused {fu*1e3:.1f} us ({(bg+bf)/fu/1e6/8000:.3f} of 8 TB/s, 8d bytes)", flush=True)
    part.close()
