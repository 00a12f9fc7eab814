"""VERDICT r3 item 4, the last traffic lever -- its UPPER BOUND, measured before anything is built: the normals of a
face cut by two tiles stored with ONE tile only (CFDP_EXP_OWNED_NORMALS=1: the other tile's blob simply lacks them, so
the values are wrong and the fetch from the neighbour's blob is FREE).  If even this does not lower the data-movement
floor of the fused pass by 5 % at 128^3, the real thing (a 4-byte index per cut face + a gather through L2) cannot.
    python tools/cut_normals_probe.py [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CFDP_PLAN_DEVICE"] = "1"  # tile blobs by the host stage (the experiment lives there)
from __graft_entry__ import load_package
pkg = load_package()
for n in [int(x) for x in sys.argv[1:]] or [64, 128]:
    gp = pkg.gen_params(n, ndomains=1)
    dom = pkg.gen_domain(gp, 0)
    pkg.fill_var(dom, None, pkg.VAR_HASH)
    for exp in ("0", "1", "0", "1"):
        os.environ["CFDP_EXPERIMENTS"] = "1"
        os.environ["CFDP_EXP_OWNED_NORMALS"] = exp
        part = pkg.GpuPartition(dom)
        part.set_fusion(True)
        iters = 400 if n <= 64 else 100
        part.time_fused(iters)
        fu = min(part.time_fused(iters) for _ in range(3))
        mv = min(part.time_fused_movement(iters) for _ in range(3))
        print(f"n={n} owned_normals={exp}: blob {part.stats['blob_bytes'] / 1e6:8.2f} MB, faces stored {part.stats['nfaces_dup']}, "
              f"fused pass {fu * 1e3:7.2f} us, movement floor {mv * 1e3:7.2f} us", flush=True)
        part.close()
    dom.free()
