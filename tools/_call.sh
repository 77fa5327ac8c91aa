cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r06_call20; mkdir -p $O
timeout 600 python -m pytest tests/test_misc.py tests/test_lr_search.py -q -m gpu -x > $O/pytest_stats.txt 2>&1; tail -2 $O/pytest_stats.txt
( time timeout 900 oracle/_ref/fixtures/SvtAv1HipFixtures --gtest_filter='HIP/av1_compute_stats_test*' ) > $O/fixtures_stats.txt 2>&1; grep -E "PASSED|FAILED|tests ran|real" $O/fixtures_stats.txt | head -8
for b in 8 4 2 0; do
SVT_HIP_STATS_BANDS=$b rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$b -o s -- python bench.py --legs lrstats --no-cpu --no-pmc --no-parity-check > $O/kt$b.txt 2>&1
python - <<PY
import csv,glob
for f in glob.glob('$O/kt$b/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'stats_' in r['Name']: print('bands=$b', r['Name'][:50], r['Calls'], r['AverageNs'])
PY
done
for b in 8 4; do SVT_HIP_STATS_BANDS=$b timeout 600 python tools/stats_census.py > $O/census_b$b.txt 2>&1; echo bands=$b; tail -7 $O/census_b$b.txt | grep -v amdgpu.ids; done
find $O -name "*kernel_trace.csv" -delete
