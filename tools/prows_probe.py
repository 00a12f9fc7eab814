"""round 5, EXPERIMENTS.md D.2 -- UPPER BOUND, values wrong: the fused pass as if the gradient phase of the previous pass had
stored P(g) = -stress(g)/2 (6 doubles) per point, so that the flux phase stages 48-byte rows instead of 80-byte ones, skips its
pass over the staged rows (and the barrier behind it), and the shared row region shrinks to 3 pieces per thread: a 32-KiB tile
image, five workgroups per CU instead of four.  Not counted: the 48 bytes per point the gradient phase would have to write.
    python tools/prows_probe.py [64 128]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
m = load_package()
for n in [int(a) for a in sys.argv[1:]] or [64, 128]:
    dom = m.gen_domain(m.gen_params(n, ndomains=1), 0); m.fill_var(dom, None, m.VAR_HASH)
    for exp in ("0", "1", "2", "0", "1", "2"):  # 1: 48-byte rows; 2: the first 48 bytes of 80-byte rows
        os.environ["CFDP_EXPERIMENTS"] = "1"
        os.environ["CFDP_EXP_PROWS"] = exp
        part = m.GpuPartition(dom); part.set_fusion(True)
        it = 400 if n <= 64 else 100
        part.time_fused(4 * it)  # (the chip's clock settles)
        ts = sorted(part.time_fused(it) for _ in range(5))
        print(f"n {n} P rows {exp}: fused pass {ts[0]*1e3:.2f} / {ts[2]*1e3:.2f} / {ts[4]*1e3:.2f} us", flush=True)
        part.close()
    dom.free()
