#!/bin/bash
# two ranks of bench.py on this GPU with the ipc transport, stderr kept (development helper)
export WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29600 + RANDOM % 300)) CFDP_SHARED_GPU=1 CFDP_DEBUG_TRACE=1
n=$1
RANK=1 LOCAL_RANK=0 timeout -k 5 200 python bench.py --gpus 2 --steps 20 --warmup 3 --transport auto --no-files > gpurun_out/b2_${n}_r1.out 2> gpurun_out/b2_${n}_r1.err &
RANK=0 LOCAL_RANK=0 timeout -k 5 200 python bench.py --gpus 2 --steps 20 --warmup 3 --transport auto --no-files > gpurun_out/b2_${n}_r0.out 2> gpurun_out/b2_${n}_r0.err
wait
