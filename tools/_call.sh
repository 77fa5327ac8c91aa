cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r03_call9; mkdir -p $O
timeout 600 python -m pytest tests/test_tf_picture.py -q -m gpu -x > $O/pytest.txt 2>&1; tail -1 $O/pytest.txt
timeout 300 python bench.py --no-cpu --legs tfpic,tpl --steps 20 --warmup 5 --no-pmc > $O/bench_tfpic.json 2> $O/bench_tfpic.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_call9/bench_tfpic.json").read().strip().split("\n")[-1])
for k,v in d["kernels"].items(): print(k, {x:v[x] for x in v if x in("ms","us","pictures_per_s","pred_64x64","pred_32x32","pred_16x16")})
PY
for h in avx2 c; do timeout 1500 python tools/enc_identity.py --case fps_4k8_p8_all,fps_4k10_p8_all,fps_1080p_p6_all,fps_1080p_p4_all --host $h --out /tmp/fps_$h > $O/fps_$h.log 2>&1; grep -a "identical=\|encoder fps\|MISMATCH\|returned" $O/fps_$h.log | cut -c1-160; done
