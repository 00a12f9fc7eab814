# development: repeat the multi-solver soak flow of tests/_rank_worker.py (4 ranks, 64^3, 48 domains) and keep the logs of runs that fail
N=${1:-10}; SOAK=${2:-2000}
mkdir -p gpurun_out/hunt
for i in $(seq 1 $N); do
  PORT=$((30000+i)) bash tools/soak_ranks.sh 4 64,64,64 48 $SOAK > gpurun_out/hunt/summary_$i.txt 2>&1
  if grep -q "Error\|assert\|setup failed" gpurun_out/hunt/summary_$i.txt gpurun_out/soak_r*.log; then
    echo "run $i: ANOMALY"; for r in 0 1 2 3; do grep -v "amdgpu.ids\|hostname\|Gloo" gpurun_out/soak_r$r.log > gpurun_out/hunt/run${i}_rank$r.log; done
    grep -h "setup failed\|VALIDATION\|AssertionError" gpurun_out/soak_r*.log | cut -c1-1500 | head -6
  else echo "run $i: ok"; fi
done
