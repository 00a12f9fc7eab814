# rehearsal of the driver's multi-GPU bench commands on a ONE-GPU box (ranks share the device: timings mean nothing,
# code paths, checks and wall time do).  The 8-rank command itself cannot run here: a GPU box admits 6 GPU processes, so
# 6 ranks stand in for the 8-GPU line (CFDP_BENCH_AS_GPUS=8: dualgrid.384 + the dualgrid.192 ride-along + the 128^3 CPU
# baseline); "4as8" does the same with 4 ranks.  (Ranks that share a device wait for their partners with the one-workgroup
# wait kernel, not inside the fused pass: the hosts select that themselves -- waiting boundary tiles of several ranks
# fill the device's workgroup slots and starve the passes they wait for, profiles/README.md.)
# Usage: bash tools/rehearse.sh [6as8|4as8|2|4 ...]
#        bash tools/rehearse.sh two-devices     -- FIRST CONTACT with a box that has >= 2 GPUs: exactly the five tests a 1-GPU
#           box skips (tests/test_multirank.py: both transports + the three memory modes of the landing block, one rank per
#           DEVICE, 1000 scaled steps each, against the whole-mesh oracle) and `bench.py --gpus 2` with one rank per device;
#           ONE summary, gpurun_out/two_devices.json: which tests passed / were rejected with what evidence, the rung the
#           bench chose and its validation record, rccl_nranks, distinct_devices, the overlap block
mkdir -p gpurun_out
if [ "$1" = two-devices ]; then
  unset CFDP_SHARED_GPU
  NDEV=$(python -c 'import torch; print(torch.cuda.device_count())')
  if [ "$NDEV" -lt 2 ]; then
    # (CFDP_TWO_DEVICES_FORCE=1: go through the motions on a 1-GPU box -- the tests skip, the two ranks of the bench share the
    # device -- so that the summary itself has been run before the day it matters)
    if [ "$CFDP_TWO_DEVICES_FORCE" != 1 ]; then echo "two-devices: this box shows $NDEV device(s); nothing to do" | tee gpurun_out/two_devices.txt; exit 4; fi
    export CFDP_SHARED_GPU=1
  fi
  timeout -k 10 1500 python -m pytest tests/test_multirank.py -m gpu -q -rs -k "between_two_devices" --junitxml=gpurun_out/two_devices_tests.xml \
      > gpurun_out/two_devices_tests.log 2>&1
  TRC=$?
  timeout -k 10 900 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/two_devices_bench.json 2> gpurun_out/two_devices_bench.err
  BRC=$?
  python - "$TRC" "$BRC" <<'PY'
import json, sys, xml.etree.ElementTree as ET
out = {"tests_rc": int(sys.argv[1]), "bench_rc": int(sys.argv[2]), "tests": {}}
try:
    for tc in ET.parse("gpurun_out/two_devices_tests.xml").getroot().iter("testcase"):
        state, why = "passed", None
        for kind in ("failure", "error", "skipped"):
            e = tc.find(kind)
            if e is not None:
                state, why = kind, (e.get("message") or e.text or "")[:600]
        out["tests"][tc.get("name")] = {"state": state, "why": why, "seconds": float(tc.get("time", 0))}
except Exception as e:
    out["tests_error"] = repr(e)
try:
    line = [l for l in open("gpurun_out/two_devices_bench.json") if l.startswith("{")][-1]
    b = json.loads(line)
    c = b["config"]
    out["bench"] = {"value": b["value"], "ms_per_step": b["ms_per_step"], "shared_gpu": b["shared_gpu"], "transport": c["transport"],
                    "exchange_protocol": c["exchange_protocol"], "transport_probe_us_per_iteration": c["transport_probe_us_per_iteration"],
                    "transport_probe_validation": c["transport_probe_validation"], "rccl_nranks": c["rccl_nranks"],
                    "distinct_devices": c["distinct_devices"], "device_of_rank": c["device_of_rank"], "peer_access_of_rank0": c["peer_access_of_rank0"],
                    "exchange_check": b.get("exchange_check"), "overlap": b.get("overlap")}
except Exception as e:
    out["bench_error"] = repr(e)
json.dump(out, open("gpurun_out/two_devices.json", "w"), indent=1)
print(json.dumps(out)[:2000])
PY
  exit $(( TRC > BRC ? TRC : BRC ))
fi
export CFDP_SHARED_GPU=1
for what in ${@:-6as8 4as8 2 4}; do
  S=$(date +%s)
  if [ "$what" = 6as8 ]; then
    CFDP_BENCH_AS_GPUS=8 timeout -k 10 900 python bench.py --gpus 6 --steps 20 --warmup 5 > gpurun_out/r6_rehearsal_n6_as8.json 2> gpurun_out/r6_rehearsal_n6_as8.err
  elif [ "$what" = 4as8 ]; then
    CFDP_BENCH_AS_GPUS=8 timeout -k 10 900 python bench.py --gpus 4 --steps 20 --warmup 5 > gpurun_out/r6_rehearsal_n4_as8.json 2> gpurun_out/r6_rehearsal_n4_as8.err
  else
    timeout -k 10 600 python bench.py --gpus $what --steps 20 --warmup 5 > gpurun_out/r6_rehearsal_n$what.json 2> gpurun_out/r6_rehearsal_n$what.err
  fi
  echo "$what: rc=$? wall_s=$(( $(date +%s) - S ))" | tee -a gpurun_out/r6_rehearsal.wall
done
