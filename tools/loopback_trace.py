"""development helper: where the time of a K = 20 batch goes in the loopback measurement (tools/loopback_probe.py), kernel by
kernel.  Run under `rocprofv3 --kernel-trace --output-format csv` with MODE=exch|free; then
`python tools/loopback_trace.py analyse <kernel_trace.csv>` prints, for the median batch, every kernel's duration and the gap
in front of it, and the batch's span from first kernel start to last kernel end."""
import os, sys, time


def run():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import load_package
    pkg = load_package()
    from cfd_proxy_amd import multigpu as mg
    name = os.environ.get("CONFIG8", "dualgrid.384")
    K = int(os.environ.get("K", "20")); reps = int(os.environ.get("REPS", "30"))
    exch = os.environ.get("MODE", "exch") == "exch"
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
    if os.environ.get("SWEEP"):  # not for the profiler: per-step time against the length of a batch, both modes
        import statistics
        for k in [int(x) for x in os.environ["SWEEP"].split(",")]:
            out = []
            for ex in (False, True):
                g.run_steps_ipc(max(3 * k, 200), use_graph=2, with_exchange=ex, overlap=True); g.sync()
                ts = []
                for _ in range(max(9, 2000 // k)):
                    g.sync(); t = time.perf_counter(); g.run_steps_ipc(k, use_graph=2, with_exchange=ex, overlap=True); g.sync()
                    ts.append((time.perf_counter() - t) / k * 1e6)
                out.append((min(ts), statistics.median(ts)))
            print(f"{name} batches of {k:5d}: comm_free best {out[0][0]:6.2f} median {out[0][1]:6.2f}   with exchange best {out[1][0]:6.2f} "
                  f"median {out[1][1]:6.2f}   ratio of bests {out[0][0] / out[1][0]:5.3f} of medians {out[0][1] / out[1][1]:5.3f}", flush=True)
        g.ipc_disconnect(); g.close()
        return
    g.run_steps_ipc(3 * K, use_graph=2, with_exchange=exch, overlap=True); g.sync()
    best = 1e9
    for _ in range(reps):
        g.sync(); t = time.perf_counter(); g.run_steps_ipc(K, use_graph=2, with_exchange=exch, overlap=True); g.sync()
        best = min(best, (time.perf_counter() - t) / K)
    assert g.ipc_error() == 0
    print(f"{name} K={K} {'with exchange' if exch else 'comm_free'}: best {best * 1e6:.2f} us/step (host clock, {reps} batches)", flush=True)
    g.ipc_disconnect(); g.close()


def analyse(path):
    import csv
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    batches, cur = [], []
    for s, e, n in rows:
        if cur and s - cur[-1][1] > 12000:  # a host sync + launch between batches: > 12 us of nothing
            batches.append(cur); cur = []
        cur.append((s, e, n))
    batches.append(cur)
    from collections import Counter
    size = Counter(len(b) for b in batches).most_common(1)[0][0]
    bs = [b for b in batches if len(b) == size]
    bs = bs[len(bs) // 4:]  # the later ones
    bs.sort(key=lambda b: b[-1][1] - b[0][0])
    b = bs[len(bs) // 2]
    print(f"{len(batches)} batches, {len(bs)} of {size} kernels used; spans min {(bs[0][-1][1] - bs[0][0][0]) / 1e3:.1f} "
          f"median {(b[-1][1] - b[0][0]) / 1e3:.1f} max {(bs[-1][-1][1] - bs[-1][0][0]) / 1e3:.1f} us")
    prev = None
    for s, e, n in b:
        short = n.split("(")[0][-70:]
        print(f"  gap {0 if prev is None else (s - prev) / 1e3:6.2f}  dur {(e - s) / 1e3:7.2f}  {short}")
        prev = e
    import statistics
    durs = [(e - s) / 1e3 for s, e, n in b[1:-1]]
    gaps = [(b[i][0] - b[i - 1][1]) / 1e3 for i in range(1, len(b))]
    print(f"  inner kernels: median dur {statistics.median(durs):.2f}, median gap {statistics.median(gaps):.2f}; first {(b[0][1] - b[0][0]) / 1e3:.2f}, last {(b[-1][1] - b[-1][0]) / 1e3:.2f}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "analyse":
        analyse(sys.argv[2])
    else:
        run()
