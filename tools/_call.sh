cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r04_final3; mkdir -p $O
timeout 200 python -m pytest tests/test_cdef.py tests/test_cdef_pick.py tests/test_host_forms.py -q -m gpu > $O/pytest_cdef.txt 2>&1; tail -2 $O/pytest_cdef.txt
timeout 560 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null; grep -v BENCH_DETAIL $O/bench_default.err > $O/bench_default.err.short; mv $O/bench_default.err.short $O/bench_default.err
wc -c $O/bench_default.json; head -c 1200 $O/bench_default.json; echo; tail -c 400 $O/bench_default.err
