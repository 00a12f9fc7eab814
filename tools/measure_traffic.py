"""Collect HBM traffic of the gradient / flux kernels with rocprofv3 PMC counters (run on the GPU box).

Separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md, rocprofv3
PMC slots), each with --kernel-trace only.  Corrections per that guide's HBM section:
  * FETCH_SIZE counts 64 B per 128-B request of a wide (16 B/lane) coalesced read on gfx950, i.e.
    exactly half the bytes: doubled here (all loads of these kernels are 16 B/lane: blob + var rows);
  * WRITE_SIZE is exact for coalesced streaming stores (checked against the known 168 B/point).
Writes profiles/<round>_traffic.json and the per-kernel counter averages as CSV.

    python tools/measure_traffic.py r01
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out_dir = os.path.join(ROOT, "gpurun_out", f"traffic_{tag}")
os.makedirs(out_dir, exist_ok=True)
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
result = {}
rows_out = []
for n, label in ((64, "dualgrid.12 lvl 2 stand-in (64^3)"), (128, "dualgrid.384 finest-level stand-in (128^3)"),
                 (64, "irregular stand-in of the dualgrid.12 lvl 2 size (64^3 points)")):
    per_kernel = collections.defaultdict(dict)
    for counters in (["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"]):
        irr = label.startswith("irregular")
        d = os.path.join(out_dir, f"n{n}{'i' if irr else ''}_{counters[0]}")
        cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
               "python3", os.path.join(ROOT, "tools", "prof_one.py")]
        e = dict(env, N=str(n), TP="0", L="0", ITERS="5", IRREGULAR="1" if irr else "0")
        r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=600)
        print(r.stdout[-300:], flush=True)
        f = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
        if not f:
            print("no counter file", r.stderr[-500:])
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f[0])):
            agg[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, dd in agg.items():
            if "gg_" in k:
                for c, v in dd.items():
                    per_kernel[k][c] = sum(v) / len(v)
                    rows_out.append((label, k, c, sum(v) / len(v), len(v)))
    entry = {}
    for k, c in per_kernel.items():
        kb_read = 2.0 * c.get("FETCH_SIZE", 0.0)   # gfx950 correction: x2 for wide coalesced reads
        kb_write = c.get("WRITE_SIZE", 0.0)
        entry[k] = {"hbm_read_bytes": kb_read * 1024, "hbm_write_bytes": kb_write * 1024,
                    "traffic_bytes": (kb_read + kb_write) * 1024, "raw": c}
    result[label] = entry
import hashlib
h = hashlib.sha256()
for f in ("cfd-proxy_amd/csrc/gg_device.h", "cfd-proxy_amd/csrc/gg_kernels.hip", "cfd-proxy_amd/host/tiling.c"):  # = bench.py kernel_source_tag()
    h.update(open(os.path.join(ROOT, f), "rb").read())
result["kernel_source_tag"] = h.hexdigest()[:16]
json.dump(result, open(os.path.join(out_dir, f"{tag}_traffic.json"), "w"), indent=1)
with open(os.path.join(out_dir, f"{tag}_pmc_counters.csv"), "w") as fh:
    w = csv.writer(fh)
    w.writerow(["workload", "kernel", "counter", "average_per_dispatch", "dispatches"])
    w.writerows(rows_out)
print(json.dumps({k: {kk: vv["traffic_bytes"] for kk, vv in v.items()} for k, v in result.items() if isinstance(v, dict)}, indent=1))
