cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r04_call21; mkdir -p $O
timeout 200 python -m pytest tests/test_cdef.py tests/test_cdef_pick.py -q -m gpu > $O/pytest_cdef.txt 2>&1; tail -2 $O/pytest_cdef.txt
timeout 200 python bench.py --steps 20 --warmup 5 --legs cdef > $O/bench_cdef.json 2> $O/bench_cdef.err; cp gpurun_out/bench_detail.json $O/bench_cdef_detail.json
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r04_call21/bench_cdef.json').read().strip().split('\n')[-1])
for k,v in j['legs'].items(): print(k, v)
PY
