cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r06_call38; mkdir -p $O
timeout 1500 python -m pytest tests/test_tpl.py tests/test_tpl_full.py -q -m gpu > $O/pytest_tpl.txt 2>&1; tail -2 $O/pytest_tpl.txt
for i in 1 2; do timeout 900 python bench.py --legs tpl,tpl1 --no-cpu --no-pmc > $O/bench.txt 2> $O/bench_err.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_detail.json'))
for n in ('tpl_recon_stage_1080p8','tpl_l1_recon_1080p8','tpl_stage_host_resident_1080p8','tpl_stage_host_1080p8'):
    k=d['kernels'].get(n,{})
    print(n, {x: round(v,1) for x,v in k.items() if isinstance(v,(int,float)) and (x.endswith('_us') or x in ('us','ms'))}, k.get('parity_checked_values'), k.get('parity'))
PY
done
SVT_HIP_TPL_RECON_FORM=5 timeout 900 python bench.py --legs tpl1 --no-cpu --no-pmc > $O/bench5.txt 2> $O/bench5_err.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_detail.json'))
k=d['kernels'].get('tpl_l1_recon_1080p8',{}); print('form 5 (fences) tpl_l1_recon', {x: round(v,1) for x,v in k.items() if isinstance(v,(int,float)) and (x.endswith('_us') or x in ('us','ms'))})
PY
timeout 1500 python -m pytest tests/test_encoder_identity.py -q -m gpu -k "tpl or everyseam" -x > $O/pytest_identity.txt 2>&1; tail -2 $O/pytest_identity.txt
