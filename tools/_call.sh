cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r03_call28; mkdir -p $O
timeout 60 python -m pytest tests/test_tpl.py -q -m gpu -x -k recon_stage > $O/pytest_tpl.txt 2>&1; echo "pytest rc $?"; tail -2 $O/pytest_tpl.txt
timeout 100 python bench.py --legs tpl --no-pmc --steps 20 --warmup 5 > $O/bench_tpl.json 2> $O/bench_tpl.err; echo "bench rc $?"; tail -c 300 $O/bench_tpl.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_call28/bench_tpl.json").read().strip().split("\n")[-1])
for k,v in d["kernels"].items(): print(k, {x:y for x,y in v.items() if x!="roofline"})
PY
