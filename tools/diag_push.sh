N=${1:-3}; PORT=${PORT:-29641}; mkdir -p gpurun_out
for r in $(seq 0 $((N-1))); do
  RANK=$r LOCAL_RANK=0 WORLD_SIZE=$N MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT OMP_NUM_THREADS=2 CFDP_IPC_WAIT_INKERNEL=1 \
    timeout -k 10 300 python tools/diag_push.py > gpurun_out/diag_r$r.log 2>&1 &
done
wait
grep -v "^\[" gpurun_out/diag_r0.log | tail -40
