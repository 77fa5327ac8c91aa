#!/bin/bash
# scratch: call 35 -- TF timing (no profiler) + parity
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python tools/microbench.py tf --steps 40 --warmup 5 > gpurun_out/c35_micro.json 2> gpurun_out/c35_micro.err
cat gpurun_out/c35_micro.json
timeout 600 python -m pytest tests/test_tf.py tests/test_rtcd_hook.py -q -m gpu -x 2>&1 | tail -2
