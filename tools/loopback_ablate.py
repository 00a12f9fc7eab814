"""diagnostic: which part of the in-kernel exchange costs what (loopback, dualgrid.384 rank 0 of 8).  CFDP_DEBUG_ABLATE bits of the
pushing fused pass: 0x100 no wait, 0x200 no row pushes, 0x400 no counting / flags."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
from cfd_proxy_amd import multigpu as mg
name, world = os.environ.get("CFG", "dualgrid.384"), int(os.environ.get("WORLD", "8"))
cfg = mg.bench_config(name, world)
gp = pkg.gen_params(*cfg["dims"], ndomains=cfg["ndomains"])
parts = [mg.build_rank_partition(gp, cfg["ndomains"], world, r, via_files=False)[0] for r in range(world)]
reqs = [{int(k): (v[0], v[1]) for k, v in pkg.merge_requests(p).items()} for p in parts]
mg.exchange_requests(parts[0], 0, world, None, all_requests=reqs)
part = parts[0]
def timed(g, steps, **kw):
    g.run_steps_ipc(200, **kw); g.sync()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); g.run_steps_ipc(steps, **kw); g.sync(); best = min(best, (time.perf_counter() - t) / steps)
    return best * 1e6
for label, bits in (("everything", 0), ("no wait", 0x100), ("no row pushes", 0x200), ("no wait, no counting/flags", 0x500),
                    ("no wait, no pushes, no counting", 0x700), ("no wait, no per-slot atomics/flags", 0x900),
                    ("no wait, no drain+barrier", 0x1100), ("no wait, neither", 0x1900)):
    os.environ["CFDP_EXPERIMENTS"] = "1"
    os.environ["CFDP_DEBUG_ABLATE"] = str(bits)
    g = pkg.GpuPartition(part)
    g.set_fusion(True)
    g.ipc_export()
    for s in range(len(g.partners())):
        g._ck(g.lib.cfdp_gpu_ipc_connect_loopback(g.h, s))
    g.ipc_ready()
    free = timed(g, 1000, with_exchange=False, overlap=True)
    exch = timed(g, 1000, with_exchange=True, overlap=True)
    print(f"{name} {label:34s} comm_free {free:6.2f} us, with exchange {exch:6.2f} us", flush=True)
    g.ipc_disconnect(); g.close()
