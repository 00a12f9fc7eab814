# the fault-injection demonstration behind tests/test_multirank.py::test_scaled_field_check_sees_a_ghost_row_read_one_exchange_early:
# 2 ranks on one GPU, CFDP_IPC_FAULT=skip_wait (the boundary tiles of the fused pass do not wait for the previous exchange).
# Prints what the old checks (final states) and the scaled-field check say about that broken path.
PORT=${PORT:-29571}
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT OMP_NUM_THREADS=2 CFDP_EXPERIMENTS=1 CFDP_IPC_FAULT=skip_wait CFDP_IPC_WAIT_INKERNEL=1 CFDP_IPC_MODE=${1:-coarse} \
    python tests/_rank_worker.py --gpu --inject-early-read > gpurun_out/stale_demo_r$r.log 2>&1 &
done
wait
grep -h "STALE_READ_EVIDENCE\|RANK_OK\|Error\|assert" gpurun_out/stale_demo_r*.log
