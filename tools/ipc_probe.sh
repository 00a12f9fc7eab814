#!/bin/bash
# two ranks of tools/ipc_probe.py on this GPU
export WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
RANK=1 LOCAL_RANK=1 timeout -k 5 200 python tools/ipc_probe.py > gpurun_out/ipc_probe_r1.log 2>&1 &
RANK=0 LOCAL_RANK=0 timeout -k 5 200 python tools/ipc_probe.py > gpurun_out/ipc_probe_r0.log 2>&1
rc=$?
wait
exit $rc
