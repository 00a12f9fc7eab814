#!/bin/bash
# round profile set (run on the GPU box): kernel stats of the bench workload and of the finest level + PMC traffic
set -o pipefail
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_$tag $R/gpurun_out/prof_${tag}_finest
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --steps 20000 --warmup 2000 --no-cpu --no-finest > $R/gpurun_out/prof_${tag}_bench.json 2> $R/gpurun_out/prof_${tag}_bench.err || exit 1
# the same with the headline workload alone (no loopback / irregular blocks): the fused kernel's row holds only launches on the 64^3 mesh
rm -rf $R/gpurun_out/prof_${tag}_headline
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_headline -- python3 $R/bench.py --steps 20000 --warmup 2000 --no-cpu --no-finest --no-loopback --no-irregular > $R/gpurun_out/prof_${tag}_headline_bench.json 2> $R/gpurun_out/prof_${tag}_headline_bench.err || exit 1
N=128 ITERS=200 TP=0 L=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_finest -- python3 $R/tools/prof_one.py > $R/gpurun_out/prof_${tag}_finest.log 2>&1 || exit 1
IRREGULAR=1 N=64 ITERS=2000 TP=0 L=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_irregular -- python3 $R/tools/prof_one.py > $R/gpurun_out/prof_${tag}_irregular.log 2>&1 || exit 1
cd $R && python3 tools/measure_traffic.py $tag > gpurun_out/traffic_$tag.log 2>&1
# one unambiguous copy of this run's summaries (gpurun merges, it never deletes older runs' files)
cp "$(ls -t gpurun_out/prof_$tag/*/*_kernel_stats.csv | sed -n 1p)" gpurun_out/prof_${tag}_kernel_stats.csv
cp "$(ls -t gpurun_out/prof_${tag}_finest/*/*_kernel_stats.csv | sed -n 1p)" gpurun_out/prof_${tag}_finest_kernel_stats.csv
cp "$(ls -t gpurun_out/prof_${tag}_irregular/*/*_kernel_stats.csv | sed -n 1p)" gpurun_out/prof_${tag}_irregular_kernel_stats.csv
cp "$(ls -t gpurun_out/prof_${tag}_headline/*/*_kernel_stats.csv | sed -n 1p)" gpurun_out/prof_${tag}_headline_kernel_stats.csv
