cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r06_call44; mkdir -p $O
timeout 900 python -m pytest tests/test_tf_picture.py tests/test_tf.py tests/test_lr_search.py -q -m gpu > $O/pytest.txt 2>&1; tail -1 $O/pytest.txt
for i in 1 2; do timeout 600 python bench.py --legs tfpic,lrsearch --no-cpu --no-pmc > $O/bench.txt 2> $O/bench_err.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_detail.json'))
print(' '.join('%s %.1f' % (n, (k.get('roofline') or {}).get('kernel_us') or k.get('ms',0)*1e3) for n,k in d['kernels'].items() if 'host' not in n))
PY
done
timeout 900 python -m pytest tests/test_encoder_identity.py -q -m gpu -k "tf or lr or everyseam" -x > $O/pytest_identity.txt 2>&1; tail -1 $O/pytest_identity.txt
