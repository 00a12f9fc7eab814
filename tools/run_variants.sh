#!/bin/bash
# kernel-variant sweep on the GPU box (development helper): tools/run_variants.sh
set -o pipefail
export TPS=${TPS:-64} PIPES=${PIPES:-0} EXPLORE_OUT=gpurun_out/variants.log SIZES=${SIZES:-64,128}
run() { echo "== lib=${1:-default} ablate=${2:-0} lanes=$3" >> $EXPLORE_OUT
  CFDP_LIBDIR=${1:+$PWD/$1} CFDP_DEBUG_ABLATE=$2 LANES=$3 timeout -k 10 200 python tools/gpu_explore.py > /dev/null 2>> gpurun_out/variants.err || exit 1; }
run "" "" 4
run "" "" 4
