cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r04_call16; mkdir -p $O
timeout 600 python -m pytest tests/test_lr_search.py tests/test_host_forms.py -q -m gpu > $O/pytest_lr.txt 2>&1; tail -2 $O/pytest_lr.txt
timeout 600 python bench.py --steps 20 --warmup 5 --legs lrsearch > $O/bench_lr.json 2> $O/bench_lr.err; echo "rc=$?"; grep -v BENCH_DETAIL $O/bench_lr.err | tail -3
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_detail.json'))
for n in ('lr_search_4k10_full','lr_search_4k10_fast'):
    k=d['kernels'][n]; print(n, round(k['ms'],3), 'ms', k['workspace_MB'], 'MB', k.get('parity_checked_units'), (k.get('cpu_baseline') or {}).get('value'))
PY
timeout 900 python -m pytest tests/test_encoder_identity.py -q -m gpu -k "lrseam or allseams or everyseam" > $O/pytest_lr_identity.txt 2>&1; tail -2 $O/pytest_lr_identity.txt
