cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r04_call1; mkdir -p $O
# 1. fast fail: one leg, no CPU, no counters
timeout 300 python bench.py --steps 5 --warmup 2 --legs sad --no-cpu --no-pmc > $O/bench_quick.json 2> $O/bench_quick.err; echo "quick rc=$? bytes=$(wc -c < $O/bench_quick.json)"; tail -c 400 $O/bench_quick.err | grep -v BENCH_DETAIL | tail -5
# 2. the default line as the driver runs it
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$? bytes=$(wc -c < $O/bench_default.json)"
grep -v BENCH_DETAIL $O/bench_default.err | tail -15
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
cat $O/bench_default.json
# 3. TPL tests incl. form 3
timeout 600 python -m pytest tests/test_tpl.py -q -m gpu -x > $O/pytest_tpl.txt 2>&1; tail -3 $O/pytest_tpl.txt
