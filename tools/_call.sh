cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r04_call8; mkdir -p $O
timeout 600 python -m pytest tests/test_tpl.py -q -m gpu > $O/pytest_tpl.txt 2>&1; tail -2 $O/pytest_tpl.txt
timeout 600 python bench.py --steps 20 --warmup 5 --legs tpl > $O/bench_tpl.json 2> $O/bench_tpl.err; echo "tpl rc=$?"; grep -v BENCH_DETAIL $O/bench_tpl.err | tail -4
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_detail.json'))
for n in ('tpl_stage_host_1080p8','tpl_stage_host_resident_1080p8','tpl_recon_stage_1080p8','tpl_src_stage_1080p8'):
    k=d['kernels'][n]; print(n, {x:(round(k[x],3) if isinstance(k[x],float) else k[x]) for x in k if x in ('ms','us','uploaded_MB','pictures_per_s')}, k['roofline'].get('kernels_per_call'))
PY
E="python tools/enc_identity.py --host avx2 --out /tmp/idt"
echo "== resident off / on x3"; for i in 1 2 3; do SVT_HIP_TPL_RESIDENT=0 timeout 200 $E --case fps_1080p_p8_all_tplrecon 2>&1 | grep -a "encoder fps"; timeout 200 $E --case fps_1080p_p8_all_tplrecon 2>&1 | grep -a "encoder fps"; done
echo "== cpu stats, resident on"; timeout 300 $E --case fps_1080p_p8_all_tplrecon --cpu-stats > $O/enc.log 2>&1; grep -a "encoder fps\|stage CPU" $O/enc.log | cut -c1-420; grep -ao "planes_found_resident': [0-9]*, 'planes_uploaded': [0-9]*" $O/enc.log
echo "== 4K 8-bit"; timeout 600 $E --case fps_4k8_p8_all_tplrecon 2>&1 | grep -a "encoder fps"
echo "== identity cases with the TPL seams"; timeout 900 python -m pytest tests/test_encoder_identity.py -q -m gpu -k "tpl" > $O/pytest_tpl_identity.txt 2>&1; tail -2 $O/pytest_tpl_identity.txt
