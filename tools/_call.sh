cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r06_call33; mkdir -p $O
timeout 600 python -m pytest tests/test_cdef_pick.py tests/test_rtcd_hook.py -q -m gpu > $O/pytest_pick.txt 2>&1; tail -2 $O/pytest_pick.txt
for i in 1 2; do timeout 600 python bench.py --legs cdefchain --no-cpu --no-pmc > $O/bench.txt 2> $O/bench_err.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_detail.json'))
print(' '.join('%s %.3f ms' % (k, v.get('ms')) for k, v in d['kernels'].items()))
PY
done
timeout 1500 python -m pytest tests/test_encoder_identity.py -q -m gpu -k "cdef or everyseam or tiny" -x > $O/pytest_identity.txt 2>&1; tail -2 $O/pytest_identity.txt
