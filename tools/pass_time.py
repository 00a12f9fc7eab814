"""time of the fused pass (event-timed, 7 x 200 passes: min / median / max), the two separate kernels and -- A/B/A/B in one
process, same box -- the forms of the fused pass (CFDP_FUSED_SPLIT=1: phase-split, 36 KiB, 4 workgroups per CU; 2: part-A rows
through registers, P(g) rows in LDS, 32 KiB, 5 workgroups per CU), 64^3 and 128^3 -- the quick A/B of a kernel change.
    python tools/pass_time.py [64 128]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
m = load_package()
forms = [f for f in os.environ.get("FORMS", "1,2,1,2").split(",") if f]
for n in [int(a) for a in sys.argv[1:]] or [64, 128]:
    dom = m.gen_domain(m.gen_params(n, ndomains=1), 0); m.fill_var(dom, None, m.VAR_HASH)
    for form in forms:
        os.environ["CFDP_FUSED_SPLIT"] = "2" if form == "x" else form
        os.environ["CFDP_EXPERIMENTS"] = "1" if form == "x" else "0"  # x: timing experiment, values wrong (EXPERIMENTS.md D.2)
        os.environ["CFDP_EXP_SKIP_PRE"] = "1" if form == "x" else "0"
        part = m.GpuPartition(dom, tile_points=int(os.environ.get("TP", "0"))); part.set_fusion(True)
        it = 200 if n <= 64 else 60
        part.time_fused(4 * it)  # (the chip's clock settles)
        ts = sorted(part.time_fused(it) for _ in range(7))
        mv = min(part.time_fused_movement(it) for _ in range(3))
        tg, tf = part.time_kernels(it, m.FLUX_CONSISTENT)
        print(f"n {n} form {form} tiles {part.stats['ntiles']} x <= {part.stats['tile_points']} points: fused pass {ts[0]*1e3:.2f} / {ts[3]*1e3:.2f} / {ts[6]*1e3:.2f} us; movement floor {mv*1e3:.2f} us; "
              f"gradient kernel {tg*1e3:.2f} us, flux kernel {tf*1e3:.2f} us", flush=True)
        part.close()
    dom.free()
