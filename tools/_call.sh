cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r06_call28; mkdir -p $O
timeout 900 python -m pytest tests/test_lr_search.py -q -m gpu -x > $O/pytest_lr.txt 2>&1; tail -2 $O/pytest_lr.txt
for i in 1 2 3; do timeout 600 python bench.py --legs lrsearch --no-cpu --no-pmc > $O/bench.txt 2> $O/bench_err.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_detail.json'))
print(' '.join('%s %.3f ms' % (k, v.get('ms')) for k, v in d['kernels'].items()))
PY
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o s -- python bench.py --legs lrsearch --no-cpu --no-pmc --no-parity-check > $O/kt.txt 2>&1
python - <<PY
import csv,glob
for f in glob.glob('$O/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'lr_' in r['Name'] or 'stats_' in r['Name']: print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
find $O -name "*kernel_trace.csv" -delete
