"""development helper: kernel timeline of a K=20 run (run under rocprofv3 --kernel-trace; prints gaps and durations)
usage on the GPU box:  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/k20trace -- python3 tools/short_run_trace.py run
                       python3 tools/short_run_trace.py parse gpurun_out/k20trace"""
import csv, glob, os, sys
if sys.argv[1] == "run":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import load_package
    m = load_package()
    dom = m.gen_domain(m.gen_params(64, ndomains=1), 0); m.fill_var(dom, None, m.VAR_HASH)
    part = m.GpuPartition(dom); part.set_fusion(True)
    part.run_iterations(200)
    for rep in range(3):
        part.prepare_iterations(20); part.sync()
        print("K=20 device ms", part.run_iterations(20)); part.sync()
    part.close()
else:
    f = sorted(glob.glob(os.path.join(sys.argv[2], "*", "*_kernel_trace.csv")), key=os.path.getmtime)[-1]
    rows = [r for r in csv.DictReader(open(f)) if "gg_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = rows[-22:]
    t0 = int(last[0]["Start_Timestamp"])
    prev_end = None
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print(f"{r['Kernel_Name'][:40]:40s} start {(s-t0)/1e3:8.1f} us  dur {(e-s)/1e3:6.1f} us  gap before {gap:5.1f} us")
        prev_end = e
    print("first start -> last end:", (int(last[-1]["End_Timestamp"]) - t0) / 1e3, "us")
