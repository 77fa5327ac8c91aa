cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r06_call24; mkdir -p $O
for p in 0 2 0 2; do
SVT_HIP_STATS_PRIO=$p rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$p -o s -- python bench.py --legs lrstats --no-cpu --no-pmc --no-parity-check > $O/kt$p.txt 2>&1
python - <<PY
import csv,glob
for f in glob.glob('$O/kt$p/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'stats_mfma' in r['Name']: print('prio=$p', r['Name'][:50], r['Calls'], r['AverageNs'])
PY
done
find $O -name "*kernel_trace.csv" -delete
