#!/bin/bash
# Full GPU regression of a round: parity tests (incl. the encoder identity runs), the default bench line (which collects its own PMC traffic in child passes),
# the frame-partition line (--mode strips: ME strips + CDEF / LR strips at N = 1) and the rocprofv3 kernel statistics of the default workload, condensed ON
# THE BOX into gpurun_out/<tag>/ (the raw traces are too large to travel).  Run from the repository root:  bash tools/gpu_regression.sh <tag> [quick]
TAG=${1:-reg}
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/$TAG; mkdir -p $O
if [ "$2" != "quick" ]; then
  timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1
fi
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null; grep -v BENCH_DETAIL $O/bench_default.err > $O/bench_default.err.short; mv $O/bench_default.err.short $O/bench_default.err
timeout 600 python bench.py --steps 20 --warmup 5 --mode strips --no-cpu --no-pmc --legs none > $O/bench_strips.json 2> $O/bench_strips.err
P="--steps 20 --warmup 5 --no-cpu --no-parity-check --no-pmc"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py $P > $O/stats.log 2>&1
python tools/pmc_summary.py ${TAG} $O/stats - - "python bench.py $P" > /dev/null 2>&1 && mv profiles/${TAG}_kernel_stats.txt $O/kernel_stats.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
du -sh $O; tail -3 $O/pytest_gpu.txt 2>/dev/null; tail -c 600 $O/bench_default.err; head -c 600 $O/bench_default.json; echo; head -c 1500 $O/bench_strips.json; tail -c 300 $O/bench_strips.err
