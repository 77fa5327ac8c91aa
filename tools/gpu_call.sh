#!/bin/bash
# scratch: call 34 -- TF kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/c34_prof -o tf -- python tools/microbench.py tf --steps 20 --warmup 3 > gpurun_out/c34_micro.json 2> gpurun_out/c34_micro.err
cat gpurun_out/c34_micro.json
find gpurun_out/c34_prof -name "*kernel_stats*" | head -2
f=$(find gpurun_out/c34_prof -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-220
