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
    st = np.zeros(nt * 24, np.uint64)
    part.lib.cfdp_gpu_debug_phase_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    part._ck(part.lib.cfdp_gpu_debug_phase_stamps(part.h, 6, st.ctypes.data))
    wv = st[nt * 8:].reshape(nt, 4, 4).astype(np.int64)
    st = st[: nt * 8].reshape(nt, 8).astype(np.int64)
    ok = (st[:, :7] > 0).all(axis=1)
    d = np.diff(st[ok, :7], axis=1).astype(np.float64)
    names = ["start -> indices here", "-> blob + rows landed", "-> flux phase done", "-> var rows in place", "-> gradients done, stores issued", "-> stores acknowledged"]
    life = (st[ok, 6] - st[ok, 0]).astype(np.float64)
    span = float(st[ok, 6].max() - st[ok, 0].min())
    print(f"n {n}: {ok.sum()} tiles stamped; pass spans {span:.0f} clocks; mean tile lifetime {life.mean():.0f} clocks (p10 {np.percentile(life,10):.0f}, p90 {np.percentile(life,90):.0f})")
    for i, nm in enumerate(names):
        print(f"   {nm:36s} mean {d[:, i].mean():8.0f}  p10 {np.percentile(d[:, i],10):8.0f}  p90 {np.percentile(d[:, i],90):8.0f}  ({100*d[:, i].mean()/life.mean():4.1f} %)")
    # barrier skew: how long the first of a tile's four waves waits for the last at the end of a phase
    okw = ok & (wv[:, :, :3] > 0).all(axis=(1, 2))
    for i, nm in enumerate(["own pieces landed", "through the flux phase", "through the gradient phase"]):
        sk = (wv[okw, :, i].max(axis=1) - wv[okw, :, i].min(axis=1)).astype(np.float64)
        print(f"   skew of the 4 waves at '{nm}': mean {sk.mean():7.0f} clocks  p50 {np.percentile(sk,50):7.0f}  p90 {np.percentile(sk,90):7.0f}")
    # a wave's own time in the phases (from the barrier that opens a phase to the wave's arrival at the next)
    fl = (wv[okw, :, 1] - st[okw, 2][:, None]).astype(np.float64)
    gr = (wv[okw, :, 2] - st[okw, 4][:, None]).astype(np.float64)
    print(f"   a wave's flux phase: mean {fl.mean():.0f} clocks (fastest wave of a tile {fl.min(axis=1).mean():.0f}, slowest {fl.max(axis=1).mean():.0f});"
          f" gradient phase: mean {gr.mean():.0f} (fastest {gr.min(axis=1).mean():.0f}, slowest {gr.max(axis=1).mean():.0f})")
    # are the tiles resident on one CU in phase with each other?  gaps between consecutive tile STARTS on a CU: bursts
    # (several starts within a few hundred clocks, then a long gap) = phase-locked; even gaps (~lifetime/4) = drifted
    hw = st[ok, 7]
    cu_key = ((hw >> 32) & 0xF) * 4096 + (hw & 0xFFFF) // 256      # xcc, then se/sh/cu bits [15:8] of HW_ID
    starts = st[ok, 0]
    gaps = []
    for key in np.unique(cu_key):
        s0 = np.sort(starts[cu_key == key])
        if len(s0) > 8:
            gaps.append(np.diff(s0)[4:-4])
    gaps = np.concatenate(gaps).astype(np.float64)
    print(f"   {len(np.unique(cu_key))} CUs seen; gaps between consecutive tile starts on a CU: mean {gaps.mean():.0f} clocks, "
          f"p10 {np.percentile(gaps,10):.0f} p25 {np.percentile(gaps,25):.0f} p50 {np.percentile(gaps,50):.0f} p75 {np.percentile(gaps,75):.0f} p90 {np.percentile(gaps,90):.0f}; "
          f"share of gaps under 1000 clocks {100*(gaps<1000).mean():.0f} %")
    part.close(); dom.free()
