cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r03_call25; mkdir -p $O
timeout 100 python -m pytest tests/test_tpl.py -q -m gpu -x > $O/pytest_tpl.txt 2>&1; tail -3 $O/pytest_tpl.txt
timeout 150 python bench.py --legs tpl --no-pmc --steps 20 --warmup 5 > $O/bench_tpl.json 2> $O/bench_tpl.err; tail -c 300 $O/bench_tpl.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_call25/bench_tpl.json").read().strip().split("\n")[-1])
for k,v in d["kernels"].items(): print(k, {x:y for x,y in v.items() if x!="roofline"})
PY
timeout 120 python tools/enc_identity.py --case tplrecon_p8_8bit,tplrecon_everyseam_1080p_p8 --out /tmp/idt > $O/identity.log 2>&1; grep -av "^$\|^SVT_HIP\|^Svt" $O/identity.log | cut -c1-100 | tail -6; grep -ao "{'pictures_offloaded[^}]*recon[^}]*}" $O/identity.log | tail -2
