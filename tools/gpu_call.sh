#!/bin/bash
# scratch: call 31 -- ME result formatting parity + timing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
true
true
timeout 300 python tools/microbench.py meresults mesession --steps 20 --warmup 3 > gpurun_out/c31_micro.json 2> gpurun_out/c31_micro.err
cat gpurun_out/c31_micro.json; tail -3 gpurun_out/c31_micro.err
