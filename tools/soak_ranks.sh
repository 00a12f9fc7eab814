# one-off soak of the xGMI write + notify path between ranks sharing one GPU: N ranks, every schedule's value checks plus
# SOAK steps in the scaled field.  usage: bash tools/soak_ranks.sh [N] [DIMS] [NDOMAINS] [SOAK]
N=${1:-3}; DIMS=${2:-16,12,10}; ND=${3:-12}; SOAK=${4:-5000}; PORT=${PORT:-29633}
mkdir -p gpurun_out
for r in $(seq 0 $((N-1))); do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=$N MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT OMP_NUM_THREADS=2 CFDP_IPC_WAIT_INKERNEL=${INKERNEL:-1} \
    timeout -k 10 800 python tests/_rank_worker.py --gpu --transport ipc --dims $DIMS --ndomains $ND --soak $SOAK $EXTRA > gpurun_out/soak_r$r.log 2>&1 &
done
wait
grep -h "RANK_OK\|Error\|assert\|CHECK" gpurun_out/soak_r*.log | cut -c1-700 | head -12
