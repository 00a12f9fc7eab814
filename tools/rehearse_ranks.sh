#!/bin/bash
# bench.py with N ranks sharing the one GPU of a gpurun box (CFDP_SHARED_GPU=1: gloo rendezvous, the
# xGMI write + notify exchange through HIP IPC between the processes).  The ranks time-slice the
# device, so the rates mean nothing; what this shows is the N>1 path end to end and its JSON line.
N=${1:-2}; STEPS=${2:-200}
R=${GRAFT_REPO_ROOT:-/root/repo}
PORT=$((20000 + RANDOM % 20000))
pids=()
for r in $(seq 0 $((N-1))); do
  RANK=$r LOCAL_RANK=0 WORLD_SIZE=$N MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT CFDP_SHARED_GPU=1 \
    timeout -k 10 500 python3 $R/bench.py --gpus $N --steps $STEPS --warmup ${3:-20} \
    > $R/gpurun_out/rehearse_n${N}_rank$r.json 2> $R/gpurun_out/rehearse_n${N}_rank$r.err &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
tail -n 1 $R/gpurun_out/rehearse_n${N}_rank0.json
exit $rc
