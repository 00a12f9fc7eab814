"""Shader-side PMC counters of the three kernels (run on the GPU box): where a CU's cycles go.
Separate rocprofv3 --pmc passes of at most four SQ counters, each with --kernel-trace only; averages
per dispatch.  Writes gpurun_out/sq_<tag>/<tag>_sq_counters.csv (copy into profiles/ to keep).

    python tools/measure_sq.py r01 [64|128]
"""
import collections
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
sizes = sys.argv[2:] or ["64", "128"]  # "64i": the generator's irregular option at that size
out_dir = os.path.join(ROOT, "gpurun_out", f"sq_{tag}")
os.makedirs(out_dir, exist_ok=True)
PASSES = [
    ["SQ_BUSY_CU_CYCLES", "SQ_WAVES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"],
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_DATA_FIFO_FULL"],
    ["SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"],
    ["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"],
    ["SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_BUSY_CYCLES"],
    ["SQ_VMEM_TA_ADDR_FIFO_FULL", "SQ_VMEM_TA_CMD_FIFO_FULL", "SQ_LDS_CMD_FIFO_FULL", "SQ_VMEM_WR_TA_DATA_FIFO_FULL"],
    ["SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_INT32"],
]
rows = []
for n in sizes:
    for counters in PASSES:
        d = os.path.join(out_dir, f"n{n}_{counters[0]}")
        cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
               "python3", os.path.join(ROOT, "tools", "prof_one.py")]
        e = dict(os.environ, TMPDIR="/tmp", N=str(n).rstrip("i"), TP="0", L="0", PIPE="-1", ITERS="5", IRREGULAR="1" if str(n).endswith("i") else "0")
        r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=600)
        f = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
        if not f:
            print("no counter file for", counters, r.stderr[-300:], flush=True)
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(sorted(f, key=os.path.getmtime)[-1])):
            agg[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, dd in agg.items():
            if "gg_" in k:
                for c, v in dd.items():
                    rows.append((n, k, c, sum(v) / len(v), len(v)))
        print(n, counters, "ok", flush=True)
with open(os.path.join(out_dir, f"{tag}_sq_counters.csv"), "w") as fh:
    w = csv.writer(fh)
    w.writerow(["lattice", "kernel", "counter", "average_per_dispatch", "dispatches"])
    w.writerows(rows)
for r in rows:
    if "fused" in r[1]:
        print(r[0], r[2], "%.4g" % r[3])
