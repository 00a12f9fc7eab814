"""development helper: the loopback bound of the exchange protocol (tools/loopback_probe.py) under plan variants set through the
environment -- which part of a tiler change moved comm_free / with_exchange.  VARIANTS = ';'-separated 'NAME=v NAME=v' lists."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
from cfd_proxy_amd import multigpu as mg

def timed(g, steps, reps, **kw):
    g.run_steps_ipc(200 if steps > 100 else 3 * steps, use_graph=2, **kw); g.sync()
    best = 1e9
    for _ in range(reps):
        g.sync(); t = time.perf_counter(); g.run_steps_ipc(steps, use_graph=2, **kw); g.sync(); best = min(best, (time.perf_counter() - t) / steps)
    return best * 1e6

variants = os.environ.get("VARIANTS", "A=1;CFDP_DEGREE_SORT=0;CFDP_TILE_BUDGET=1;A=2").split(";")
for name in os.environ.get("CONFIGS8", "dualgrid.384,dualgrid.192").split(","):  # NAME or NAME:RANKS (8 ranks unless said)
    name, _, w = name.partition(":")
    world = int(w or 8)
    cfg = mg.bench_config(name, world)
    gp = pkg.gen_params(*cfg["dims"], ndomains=cfg["ndomains"])
    parts = [mg.build_rank_partition(gp, cfg["ndomains"], world, r, via_files=False)[0] for r in range(world)]
    reqs = [{int(k): (v[0], v[1]) for k, v in pkg.merge_requests(p).items()} for p in parts]
    mg.exchange_requests(parts[0], 0, world, None, all_requests=reqs)
    for var in variants:
        sets = [kv.split("=", 1) for kv in var.split() if "=" in kv]
        for k, v in sets:
            os.environ[k] = v
        g = pkg.GpuPartition(parts[0])
        g.set_fusion(True)
        g.ipc_configure(memory_mode=mg.ipc_mode_attempts()[0], notify="counter")
        g.ipc_export()
        for s in range(len(g.partners())):
            g._ck(g.lib.cfdp_gpu_ipc_connect_loopback(g.h, s))
        g.ipc_ready()
        pkg.kernel_forms()
        free = timed(g, 1000, 3, with_exchange=False, overlap=True)
        exch = timed(g, 1000, 3, with_exchange=True, overlap=True)
        forms = sorted(set(f.split("@")[0] for f in pkg.kernel_forms().split()))
        f20 = timed(g, 20, 9, with_exchange=False, overlap=True)
        e20 = timed(g, 20, 9, with_exchange=True, overlap=True)
        assert g.ipc_error() == 0
        print(f"{name} [{var.strip()}] tiles {g.stats['ntiles']} (boundary {g.stats['nbtiles']}) groups {g.stats['groups']}: steady {free:6.2f} / {exch:6.2f} -> {free / exch:5.3f}; "
              f"K=20 {f20:6.2f} / {e20:6.2f} -> {f20 / e20:5.3f}  {forms}", flush=True)
        g.ipc_disconnect()
        g.close()
        for k, _ in sets:
            os.environ.pop(k, None)
    for p in parts:
        p.free()
