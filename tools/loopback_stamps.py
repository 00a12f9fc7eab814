"""development helper: the life of the tiles of ONE pass of the loopback measurement (tools/loopback_probe.py), with and without
the exchange riding in it -- when the boundary tiles and the interior tiles start and end relative to the first stamp of their
XCD (s_memtime: shader clock ticks, a clock per XCD), and how long their phases take.  Needs lib/libcfdproxy_diag.so."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
from cfd_proxy_amd import multigpu as mg

name = os.environ.get("CONFIG8", "dualgrid.384")
cfg = mg.bench_config(name, 8)
gp = pkg.gen_params(*cfg["dims"], ndomains=cfg["ndomains"])
parts = [mg.build_rank_partition(gp, cfg["ndomains"], 8, r, via_files=False)[0] for r in range(8)]
reqs = [{int(k): (v[0], v[1]) for k, v in pkg.merge_requests(p).items()} for p in parts]
mg.exchange_requests(parts[0], 0, 8, None, all_requests=reqs)
g = pkg.GpuPartition(parts[0])
g.set_fusion(True)
g.ipc_configure(memory_mode=mg.ipc_mode_attempts()[0], notify=os.environ.get("NOTIFY", "counter"))
g.ipc_export()
for s in range(len(g.partners())):
    g._ck(g.lib.cfdp_gpu_ipc_connect_loopback(g.h, s))
g.ipc_ready()
nt, nb = g.stats["ntiles"], g.stats["nbtiles"]
g.lib.cfdp_gpu_debug_phase_stamps_ipc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
names = ["indices", "loads landed", "flux done", "var rows in place", "gradients done", "stores acknowledged", "pushed + counted"]
for ex in (0, 1, 0, 1):
    g.run_steps_ipc(200, use_graph=2, with_exchange=bool(ex), overlap=True); g.sync()
    raw = np.zeros(nt * 24, np.uint64)
    g._ck(g.lib.cfdp_gpu_debug_phase_stamps_ipc(g.h, 6, ex, raw.ctypes.data))
    st = raw[: nt * 8].reshape(nt, 8).astype(np.int64)
    wv = raw[nt * 8:].reshape(nt, 4, 4).astype(np.int64)
    ok = (st[:, :7] > 0).all(axis=1) & (wv[:, :, 3] > 0).all(axis=1)
    xcc = ((st[:, 7] >> 32) & 0xF).astype(np.int64)  # XCC_ID register in the high word, HW_ID in the low one
    # the clock (s_memtime) is per XCD: tiles are placed on the time axis of their XCD, whose first tile starts at 0
    endraw = wv[:, :, 3].max(axis=1)
    st8 = np.concatenate([st[:, :7], wv[:, :, 3].max(axis=1, keepdims=True)], axis=1)  # .. , the last wave through push + count
    for x in np.unique(xcc[ok]):
        m = ok & (xcc == x)
        st8[m] -= st[m, 0].min()
    start, end = st8[:, 0] / 1e3, st8[:, 7] / 1e3  # kiloticks of the shader clock
    print(f"{name} {'WITH exchange' if ex else 'comm_free   '}: {ok.sum()} of {nt} tiles stamped on {len(np.unique(xcc[ok]))} XCDs, pass spans {end[ok].max():.1f} kiloticks "
          f"(XCD by XCD: {' '.join(f'{end[ok & (xcc == x)].max():.1f}' for x in np.unique(xcc[ok]))})")
    for what, sel in (("boundary", np.arange(nt) < nb), ("interior", np.arange(nt) >= nb)):
        m = ok & sel
        life = end[m] - start[m]
        ph = np.diff(st8[m], axis=1) / 1e3
        print(f"   {what} ({m.sum():5d}): start mean {start[m].mean():6.2f} p90 {np.percentile(start[m], 90):6.2f} max {start[m].max():6.2f}   "
              f"end mean {end[m].mean():6.2f} max {end[m].max():6.2f}   life mean {life.mean():6.2f} p90 {np.percentile(life, 90):6.2f} max {life.max():6.2f}")
        print("        phases (mean kiloticks): " + "  ".join(f"{n} {ph[:, i].mean():.2f}" for i, n in enumerate(names)))
    # CU by CU (one clock): how many tiles a CU holds at a time, how many it runs, how long it is busy
    key = (xcc << 16) | (st[:, 7] & 0xFF00)  # XCD | SE, SH, CU of HW_ID
    cus = np.unique(key[ok])
    conc, count, busy, bfirst = [], [], [], []
    for k in cus:
        m = np.where(ok & (key == k))[0]
        ev = sorted([(st[i, 0], 1) for i in m] + [(endraw[i], -1) for i in m])
        c = best = 0
        for _, d in ev:
            c += d; best = max(best, c)
        conc.append(best); count.append(len(m))
        busy.append((max(endraw[i] for i in m) - min(st[i, 0] for i in m)) / 1e3)
        bfirst.append(int((m < nb).sum()))
    conc, count, busy, bfirst = map(np.array, (conc, count, busy, bfirst))
    print(f"   {len(cus)} CUs: tiles at a time max {np.bincount(conc)[1:].tolist()} (CUs holding 1, 2, ... at most)  tiles per CU mean {count.mean():.1f} min {count.min()} max {count.max()}  "
          f"busy kiloticks mean {busy.mean():.1f} p10 {np.percentile(busy, 10):.1f} p90 {np.percentile(busy, 90):.1f} max {busy.max():.1f}  boundary tiles per CU max {bfirst.max()}", flush=True)
    for nbt in range(0, int(bfirst.max()) + 1):
        m = bfirst == nbt
        if m.any():
            print(f"      CUs with {nbt} boundary tiles: {m.sum():3d}, busy mean {busy[m].mean():.1f}, tiles per CU {count[m].mean():.1f}")
assert g.ipc_error() == 0
g.ipc_disconnect(); g.close()
