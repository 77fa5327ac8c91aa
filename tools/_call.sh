cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r04_call4; mkdir -p $O
# the default line as the driver runs it, then the strips line, then the kernel statistics of the default workload
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$? bytes=$(wc -c < $O/bench_default.json)"
grep -v BENCH_DETAIL $O/bench_default.err | tail -8
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
cat $O/bench_default.json
timeout 600 python bench.py --steps 20 --warmup 5 --mode strips --no-cpu --no-pmc --legs none > $O/bench_strips.json 2> $O/bench_strips.err; echo "strips rc=$? bytes=$(wc -c < $O/bench_strips.json)"; cat $O/bench_strips.json
P="--steps 20 --warmup 5 --no-cpu --no-parity-check --no-pmc"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py $P > $O/stats.log 2>&1
python tools/pmc_summary.py r04_call4 $O/stats - - "python bench.py $P" > /dev/null 2>&1 && mv profiles/r04_call4_kernel_stats.txt $O/kernel_stats.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; rm -rf $O/stats
head -40 $O/kernel_stats.txt | cut -c1-150
timeout 300 python -m pytest tests/test_partition.py tests/test_lr_search.py -q -m gpu > $O/pytest_part_lr.txt 2>&1; tail -2 $O/pytest_part_lr.txt
