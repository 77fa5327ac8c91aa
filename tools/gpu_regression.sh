#!/bin/bash
# Full GPU regression of a round: parity tests (incl. the encoder identity runs), the default and --extra bench lines, rocprofv3 kernel statistics and the
# two PMC traffic passes.  Run on the GPU box from the repository root:  bash tools/gpu_regression.sh <tag> [quick]
# (outputs under gpurun_out/<tag>/; condense with tools/pmc_summary.py)
TAG=${1:-reg}
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
P="--steps 40 --warmup 5 --no-cpu --no-parity-check"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py $P > $O/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python bench.py $P > $O/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python bench.py $P > $O/write.log 2>&1
if [ "$2" != "quick" ]; then
  timeout 900 python bench.py --extra --steps 40 > $O/bench_extra.json 2> $O/bench_extra.err
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_extra -o s -- python bench.py --steps 10 --warmup 2 --no-cpu --no-parity-check --extra > $O/stats_extra.log 2>&1
fi
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -size +20M -exec sh -c 'head -c 20000000 "$1" > "$1.head" && rm "$1"' _ {} \;
du -sh $O; tail -3 $O/pytest_gpu.txt; tail -c 600 $O/bench_default.err; head -c 1500 $O/bench_default.json
