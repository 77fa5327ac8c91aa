cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r06_call21; mkdir -p $O
timeout 900 python -m pytest tests/test_misc.py tests/test_lr_search.py -q -m gpu -x > $O/pytest_lr.txt 2>&1; tail -2 $O/pytest_lr.txt
for i in 1 2; do timeout 600 python bench.py --legs lrsearch,lrstats --no-cpu --no-pmc > $O/bench_lr.txt 2> $O/bench_lr_err.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_detail.json'))
for k,v in d['kernels'].items():
    print(k, 'ms', round(v.get('ms'),4))
PY
done
rocprofv3 --kernel-trace --output-format csv -d $O/lrs -o s -- python bench.py --legs lrsearch --no-cpu --no-pmc --no-parity-check > $O/lrs.txt 2>&1
python tools/lr_timeline.py $O/lrs $O/lrs_timeline.txt; grep "^call" $O/lrs_timeline.txt
find $O -name "*kernel_trace.csv" -delete
