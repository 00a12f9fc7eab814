# rehearsal of the driver's multi-GPU bench commands on a ONE-GPU box (ranks share the device: timings mean nothing,
# code paths, checks and wall time do).  The 8-rank command itself cannot run here: a GPU box admits 6 GPU processes, so
# 6 ranks stand in for the 8-GPU line (CFDP_BENCH_AS_GPUS=8: dualgrid.384 + the dualgrid.192 ride-along + the 128^3 CPU
# baseline); "4as8" does the same with 4 ranks.  (Ranks that share a device wait for their partners with the one-workgroup
# wait kernel, not inside the fused pass: the hosts select that themselves -- waiting boundary tiles of several ranks
# fill the device's workgroup slots and starve the passes they wait for, profiles/README.md.)
# Usage: bash tools/rehearse.sh [6as8|4as8|2|4 ...]
mkdir -p gpurun_out
export CFDP_SHARED_GPU=1
for what in ${@:-6as8 4as8 2 4}; do
  S=$(date +%s)
  if [ "$what" = 6as8 ]; then
    CFDP_BENCH_AS_GPUS=8 timeout -k 10 900 python bench.py --gpus 6 --steps 20 --warmup 5 > gpurun_out/r5_rehearsal_n6_as8.json 2> gpurun_out/r5_rehearsal_n6_as8.err
  elif [ "$what" = 4as8 ]; then
    CFDP_BENCH_AS_GPUS=8 timeout -k 10 900 python bench.py --gpus 4 --steps 20 --warmup 5 > gpurun_out/r5_rehearsal_n4_as8.json 2> gpurun_out/r5_rehearsal_n4_as8.err
  else
    timeout -k 10 600 python bench.py --gpus $what --steps 20 --warmup 5 > gpurun_out/r5_rehearsal_n$what.json 2> gpurun_out/r5_rehearsal_n$what.err
  fi
  echo "$what: rc=$? wall_s=$(( $(date +%s) - S ))" | tee -a gpurun_out/r5_rehearsal.wall
done
