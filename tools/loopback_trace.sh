#!/bin/bash
# the two traces of tools/loopback_trace.py (run on the GPU box) and their analysis
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in free exch; do
  rm -rf $R/gpurun_out/lbtrace_$mode
  MODE=$mode rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/lbtrace_$mode -- python3 $R/tools/loopback_trace.py > $R/gpurun_out/lbtrace_$mode.log 2>&1 || exit 1
  f=$(ls -t $R/gpurun_out/lbtrace_$mode/*/*_kernel_trace.csv | sed -n 1p)
  python3 $R/tools/loopback_trace.py analyse "$f" >> $R/gpurun_out/lbtrace_$mode.log 2>&1 || exit 1
  rm -rf $R/gpurun_out/lbtrace_$mode
done
