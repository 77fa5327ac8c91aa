cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r06_call4; mkdir -p $O
B=oracle/_ref/fixtures/SvtAv1HipFixtures
# A/B: the per-call CDEF / deblocking wrappers with and without the zero-copy small-call mode (one process, no contention)
F='HIP/CDEFBlockTest.MatchTest/1:HIP/CDEFBlockTest.MatchTest/100:HIP/LbdLoopFilterTest.*:HIP/CDEFFindDirFewerRepeatsTest.*'
( time $B --gtest_filter="$F" ) > $O/ab_zero_copy.txt 2>&1; grep -E "^\[       OK|real" $O/ab_zero_copy.txt | tail -14
( time SVT_HIP_NO_ZERO_COPY=1 $B --gtest_filter="$F" ) > $O/ab_staged.txt 2>&1; grep -E "^\[       OK|real" $O/ab_staged.txt | tail -14
# the fixture pytest
( time timeout 1200 python -m pytest tests/test_ref_fixtures.py -q -m gpu -x ) > $O/pytest_fixtures.txt 2>&1; tail -5 $O/pytest_fixtures.txt
# the bench line
( time timeout 1700 python bench.py ) > $O/bench_stdout.txt 2> $O/bench_stderr.txt; tail -c 1200 $O/bench_stdout.txt; cp gpurun_out/bench_detail.json $O/ 2>/dev/null
grep -v BENCH_DETAIL $O/bench_stderr.txt | tail -6 | cut -c1-400
