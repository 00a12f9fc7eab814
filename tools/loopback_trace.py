"""development helper: kernel timeline of K = 20 exchanging steps in loopback (dualgrid.384 rank 0 of 8)
usage on the GPU box:  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lbtrace -- python3 tools/loopback_trace.py run
                       python3 tools/loopback_trace.py parse gpurun_out/lbtrace"""
import csv, glob, os, sys
if sys.argv[1] == "run":
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
    g = pkg.GpuPartition(parts[0]); g.set_fusion(True); g.ipc_export()
    for s in range(len(g.partners())):
        g._ck(g.lib.cfdp_gpu_ipc_connect_loopback(g.h, s))
    g.ipc_ready()
    g.run_steps_ipc(200, with_exchange=True, overlap=True); g.sync()
    for rep in range(4):
        g.run_steps_ipc(20, with_exchange=bool(int(os.environ.get("EXCH", "1"))), overlap=True, use_graph=int(os.environ.get("UG", "2"))); g.sync()
    g.ipc_disconnect(); g.close()
else:
    f = sorted(glob.glob(os.path.join(sys.argv[2], "*", "*_kernel_trace.csv")), key=os.path.getmtime)[-1]
    rows = [r for r in csv.DictReader(open(f)) if "gg_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = rows[-26:]
    t0 = int(last[0]["Start_Timestamp"]); prev_end = None
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print(f"{r['Kernel_Name'][:58]:58s} start {(s-t0)/1e3:8.1f} us  dur {(e-s)/1e3:6.1f} us  gap before {gap:6.1f} us")
        prev_end = e
