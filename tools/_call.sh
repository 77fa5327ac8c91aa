cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r06_call6; mkdir -p $O
for x in 1 0; do SVT_HIP_TF_XCD=$x timeout 600 python bench.py --legs tf,tfpic --no-cpu > $O/bench_tf_xcd$x.txt 2> $O/bench_tf_xcd${x}_err.txt; python - <<PY
import json
try:
    d=json.load(open('gpurun_out/bench_detail.json'))
    for n in ('tf_subpel_1080p8_6refs','tf_picture_stage_1080p8_4refs_resident'):
        r=d['kernels'][n]['roofline']; print('tf xcd=$x', n, 'us', r.get('kernel_us'), 'moved/alg', r.get('moved_over_algorithmic'), 'valu', r.get('valu_frac'))
except Exception as e: print('tf legs:', e)
PY
done
timeout 300 python -m pytest tests/test_tf_subpel.py tests/test_tf_picture.py tests/test_tf.py -q -m gpu > $O/pytest_tf.txt 2>&1; tail -2 $O/pytest_tf.txt
# the MFMA statistics leg under the kernel trace: where its 0.20 ms go
cd /tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_lrstats -o p -- python $GRAFT_REPO_ROOT/bench.py --legs lrstats --no-cpu --no-pmc > $GRAFT_REPO_ROOT/$O/prof_lrstats.txt 2>&1; cd $GRAFT_REPO_ROOT
python - <<'PY'
import glob,csv
for f in glob.glob('gpurun_out/r06_call6/prof_lrstats/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:8]: print(r.get('Name','')[:70], r.get('Calls'), r.get('TotalDurationNs'), r.get('AverageNs'), r.get('Percentage'))
PY
