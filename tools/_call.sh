cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r06_call3; mkdir -p $O
# 1. the rewritten MFMA statistics kernel: parity (pytest + the reference's own fixture suites), then its leg
timeout 600 python -m pytest tests/test_misc.py tests/test_lr_search.py -q -m gpu -x > $O/pytest_stats.txt 2>&1; tail -3 $O/pytest_stats.txt
B=oracle/_ref/fixtures/SvtAv1HipFixtures
for i in 0 1 2 3 4 5 6 7; do ( GTEST_TOTAL_SHARDS=8 GTEST_SHARD_INDEX=$i timeout 600 $B --gtest_filter='HIP/av1_compute_stats_test*' > $O/stats_fix_$i.txt 2>&1 ) & done; wait
grep -h "PASSED\|FAILED TEST" $O/stats_fix_*.txt | sort | uniq -c
timeout 300 python bench.py --legs lrstats --no-cpu > $O/bench_lrstats.txt 2> $O/bench_lrstats_err.txt; python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/bench_detail.json')); k=d['kernels']['lr_compute_stats_4k10_win7']; print('lr_stats', k['ms'], k['roofline'].get('frac'), k['roofline'].get('valu_frac'))
except Exception as e: print('lrstats leg:', e)
PY
# 2. the whole bench line
( time timeout 1700 python bench.py ) > $O/bench_stdout.txt 2> $O/bench_stderr.txt; tail -c 2500 $O/bench_stdout.txt; cp gpurun_out/bench_detail.json $O/ 2>/dev/null
grep -v BENCH_DETAIL $O/bench_stderr.txt | tail -8 | cut -c1-500
# 3. the fixture pytest with the thinner interior-CDEF sets, 24 shards
( time timeout 1200 python -m pytest tests/test_ref_fixtures.py -q -m gpu -x ) > $O/pytest_fixtures.txt 2>&1; tail -5 $O/pytest_fixtures.txt
