# rocprofv3 kernel statistics of RANK 0 of a two-rank run (ranks share this GPU; rank 1 runs unprofiled beside it): what one
# iteration with the xGMI write + notify exchange launches.  CFDP_IPC_WAIT_INKERNEL=1: the schedule of ranks with a GPU each.
# Usage (on a GPU box): bash tools/profile_two_ranks.sh    -> gpurun_out/prof_n2/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export CFDP_SHARED_GPU=1 CFDP_IPC_WAIT_INKERNEL=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 WORLD_SIZE=2 LOCAL_RANK=0
ARGS="--gpus 2 --steps 2000 --warmup 100 --no-cpu --no-weak --transport ipc"
RANK=1 timeout -k 10 500 python3 bench.py $ARGS > gpurun_out/prof_n2_rank1.log 2>&1 &
R1=$!
RANK=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_n2 -o n2 -- python3 bench.py $ARGS > gpurun_out/prof_n2_rank0.log 2>&1
rc=$?
wait $R1
echo "rank0 rc=$rc rank1 rc=$?"
ls gpurun_out/prof_n2
