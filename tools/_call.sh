cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r04_call15; mkdir -p $O
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$? bytes=$(wc -c < $O/bench_default.json)"
grep -v BENCH_DETAIL $O/bench_default.err | tail -5
cp gpurun_out/bench_detail.json $O/bench_detail.json
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r04_call15/bench_default.json').read().strip().splitlines()[-1])
print(l['encoder_fps_1080p_preset8'])
d=json.load(open('gpurun_out/bench_detail.json')); print(d['encoder_fps_1080p_preset8'].get('fps_avx2_pairs'))
PY
