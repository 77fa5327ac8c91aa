cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r04_call3; mkdir -p $O
timeout 600 python -m pytest tests/test_tpl.py -q -m gpu > $O/pytest_tpl.txt 2>&1; tail -3 $O/pytest_tpl.txt
timeout 600 python bench.py --steps 20 --warmup 5 --legs tpl > $O/bench_tpl.json 2> $O/bench_tpl.err; echo "tpl rc=$?"; grep -v BENCH_DETAIL $O/bench_tpl.err | tail -5
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_detail.json'))
k=d['kernels']['tpl_recon_stage_1080p8']; print({x:k[x] for x in k if x.endswith('_us') or x in ('us','intra_blocks','blocks_16x16')}); print(k['roofline'])
PY
E="python tools/enc_identity.py --host avx2 --out /tmp/idt"
echo "== tplrecon, cpu stats (default sync)"; timeout 300 $E --case fps_1080p_p8_all_tplrecon --cpu-stats > $O/enc_tplrecon.log 2>&1; grep -a "identical=\|encoder fps\|stage CPU" $O/enc_tplrecon.log | cut -c1-420
echo "== fps repeated x4 (default, block)"; for i in 1 2 3 4; do timeout 200 $E --case fps_1080p_p8_all_tplrecon 2>&1 | grep -a "encoder fps"; SVT_HIP_SYNC=block timeout 200 $E --case fps_1080p_p8_all_tplrecon 2>&1 | grep -a "encoder fps"; done
echo "== instances 4 (default, block)"
timeout 300 $E --case fps_1080p_p8_all_tplrecon --instances 4; SVT_HIP_SYNC=block timeout 300 $E --case fps_1080p_p8_all_tplrecon --instances 4
echo "== 300 frames x2"; for i in 1 2; do timeout 300 $E --case fps_1080p_p8_all_tplrecon_300 2>&1 | grep -a "encoder fps"; done
echo "== avx512 host"; timeout 300 python tools/enc_identity.py --host avx512 --out /tmp/idt --case fps_1080p_p8_all_tplrecon --cpu-stats 2>&1 | grep -a "identical=\|encoder fps\|stage CPU" | cut -c1-420
echo "== full regression"
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
