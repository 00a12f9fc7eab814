"""What does the xGMI write + notify protocol itself cost per iteration when no partner is ever late?  Rank 0's partition of
the 2 / 4 / 8-rank decompositions on ONE GPU, every partner slot looped back to the rank's own landing arenas and flag words
(cfdp_gpu_ipc_connect_loopback: wrong ghost values, right traffic and right protocol): iterations with the exchange riding in
the fused pass (in-kernel wait, push, per-partner notify) against iterations without exchange.  The ratio is an UPPER bound
of the overlap efficiency a rank with a GPU of its own can reach (remote stores cross xGMI there, local ones do not)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
from cfd_proxy_amd import multigpu as mg

def timed(g, steps, **kw):
    g.run_steps_ipc(200, **kw); g.sync()  # (open chunks of 50: the steady state)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); g.run_steps_ipc(steps, **kw); g.sync(); best = min(best, (time.perf_counter() - t) / steps)
    return best * 1e6

for name, world in (("dualgrid.24", 2), ("dualgrid.48", 4), ("dualgrid.192", 8), ("dualgrid.384", 8)):
    cfg = mg.bench_config(name, world)
    gp = pkg.gen_params(*cfg["dims"], ndomains=cfg["ndomains"])
    parts = [mg.build_rank_partition(gp, cfg["ndomains"], world, r, via_files=False)[0] for r in range(world)]
    reqs = [{int(k): (v[0], v[1]) for k, v in pkg.merge_requests(p).items()} for p in parts]
    mg.exchange_requests(parts[0], 0, world, None, all_requests=reqs)
    part = parts[0]
    for variant, env, notify in (("per-partner counters, wait in the pass", {}, "counter"),
                                 ("per-partner flags, wait in the pass", {}, "flag"),
                                 ("one completion counter + flags, wait in the pass", {"CFDP_IPC_PER_PARTNER": "0"}, "flag"),
                                 ("per-partner counters, wait kernel", {"CFDP_IPC_WAIT_INKERNEL": "0"}, "counter"),
                                 ("push + notify + wait kernels", {"CFDP_IPC_INKERNEL": "0"}, "counter")):
        for k in ("CFDP_IPC_PER_PARTNER", "CFDP_IPC_WAIT_INKERNEL", "CFDP_IPC_INKERNEL"):
            os.environ.pop(k, None)
        os.environ.update(env)
        g = pkg.GpuPartition(part)
        g.set_fusion(True)
        g.ipc_configure(notify=notify)
        g.ipc_export()
        try:
            for s in range(len(g.partners())):
                g._ck(g.lib.cfdp_gpu_ipc_connect_loopback(g.h, s))
            g.ipc_ready()
        except Exception as e:
            print(name, "loopback not possible:", e); g.close(); break
        free = timed(g, 2000, with_exchange=False, overlap=True)
        exch = timed(g, 2000, with_exchange=True, overlap=True)
        assert g.ipc_error() == 0
        if not env:  # the driver's flags: K = 20 steps between two syncs (what a bench line records at N > 1)
            def short(ug, **kw):
                best = 1e9
                for _ in range(7):
                    g.sync(); t = time.perf_counter(); g.run_steps_ipc(20, use_graph=ug, **kw); g.sync(); best = min(best, (time.perf_counter() - t) / 20)
                return best * 1e6
            for ug, what in ((1, "open graph + the sync's flux and wait from the streams"), (2, "ONE closed graph")):
                f20, e20 = short(ug, with_exchange=False, overlap=True), short(ug, with_exchange=True, overlap=True)
                print(f"{name:13s} rank 0 of {world}: K = 20 between syncs, {what}: comm_free {f20:6.2f} us/step, with exchange {e20:6.2f} us/step  -> {f20 / e20:5.3f}", flush=True)
            print(f"{name:13s} graph replay: {g.ipc_graph_stats()}", flush=True)
        print(f"{name:13s} rank 0 of {world} ({part.nown} points, {len(g.partners())} partners, {g.stats['nbtiles']} boundary tiles of {g.stats['ntiles']}): "
              f"{variant:38s} comm_free {free:6.2f} us, with exchange {exch:6.2f} us  -> efficiency bound {free / exch:5.3f}", flush=True)
        g.ipc_disconnect()
        g.close()
    for p in parts:
        p.free()
