cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r06_call41; mkdir -p $O
for i in 1 2 3; do timeout 1500 python -m pytest tests/test_tpl.py tests/test_tpl_full.py -q -m gpu > $O/pytest_tpl_$i.txt 2>&1; tail -1 $O/pytest_tpl_$i.txt; done
for i in 1 2; do timeout 900 python bench.py --legs tpl,tpl1 --no-cpu --no-pmc > $O/bench.txt 2> $O/bench_err.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_detail.json'))
for n in ('tpl_recon_stage_1080p8','tpl_l1_recon_1080p8','tpl_stage_host_resident_1080p8'):
    k=d['kernels'].get(n,{})
    print(n, {x: round(v,1) for x,v in k.items() if isinstance(v,(int,float)) and (x.endswith('_us') or x in ('us','ms'))})
PY
done
timeout 1500 python -m pytest tests/test_encoder_identity.py -q -m gpu -k "tpl or everyseam" -x > $O/pytest_identity.txt 2>&1; tail -1 $O/pytest_identity.txt
