"""diagnostics: where does a tile of the fused pass spend its life?  Shader-clock stamps of the phase boundaries of
every workgroup (thread 0) of one fused pass, averaged over tiles.  python tools/phase_stamps.py [64|128]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
m = load_package()
for n in [int(a) for a in sys.argv[1:]] or [64, 128]:
    dom = m.gen_domain(m.gen_params(n, ndomains=1), 0); m.fill_var(dom, None, m.VAR_HASH)
    part = m.GpuPartition(dom); part.set_fusion(True); part.time_fused(20)
    nt = part.stats["ntiles"]
    st = np.zeros(nt * 8, np.uint64)
    part.lib.cfdp_gpu_debug_phase_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    part._ck(part.lib.cfdp_gpu_debug_phase_stamps(part.h, 6, st.ctypes.data))
    st = st.reshape(nt, 8).astype(np.int64)
    ok = (st[:, :7] > 0).all(axis=1)
    d = np.diff(st[ok, :7], axis=1).astype(np.float64)
    names = ["start -> indices here", "-> blob + rows landed", "-> flux phase done", "-> var rows in place", "-> gradients done, stores issued", "-> stores acknowledged"]
    life = (st[ok, 6] - st[ok, 0]).astype(np.float64)
    span = float(st[ok, 6].max() - st[ok, 0].min())
    print(f"n {n}: {ok.sum()} tiles stamped; pass spans {span:.0f} clocks; mean tile lifetime {life.mean():.0f} clocks (p10 {np.percentile(life,10):.0f}, p90 {np.percentile(life,90):.0f})")
    for i, nm in enumerate(names):
        print(f"   {nm:36s} mean {d[:, i].mean():8.0f}  p10 {np.percentile(d[:, i],10):8.0f}  p90 {np.percentile(d[:, i],90):8.0f}  ({100*d[:, i].mean()/life.mean():4.1f} %)")
    part.close(); dom.free()
