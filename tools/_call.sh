cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r04_call2; mkdir -p $O
timeout 600 python -m pytest tests/test_tpl.py -q -m gpu > $O/pytest_tpl.txt 2>&1; tail -3 $O/pytest_tpl.txt
timeout 600 python bench.py --steps 20 --warmup 5 --legs tpl > $O/bench_tpl.json 2> $O/bench_tpl.err; echo "tpl rc=$?"; grep -v BENCH_DETAIL $O/bench_tpl.err | tail -5
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_detail.json'))
k=d['kernels']['tpl_recon_stage_1080p8']; print({x:k[x] for x in k if x.endswith('_us') or x in ('us','intra_blocks','blocks_16x16')}); print(k['roofline'])
PY
E="python tools/enc_identity.py --host avx2 --out /tmp/idt"
echo "== default sync, cpu stats"; timeout 300 $E --case fps_1080p_p8_all --cpu-stats > $O/enc_default.log 2>&1; grep -a "identical=\|encoder fps\|stage CPU" $O/enc_default.log | cut -c1-400
echo "== blocking sync, cpu stats"; SVT_HIP_SYNC=block timeout 300 $E --case fps_1080p_p8_all --cpu-stats > $O/enc_block.log 2>&1; grep -a "identical=\|encoder fps\|stage CPU" $O/enc_block.log | cut -c1-400
echo "== tplrecon (form 4), default / blocking"; timeout 300 $E --case fps_1080p_p8_all_tplrecon --cpu-stats > $O/enc_tplrecon.log 2>&1; grep -a "identical=\|encoder fps\|stage CPU" $O/enc_tplrecon.log | cut -c1-400
SVT_HIP_SYNC=block timeout 300 $E --case fps_1080p_p8_all_tplrecon > $O/enc_tplrecon_block.log 2>&1; grep -a "identical=\|encoder fps" $O/enc_tplrecon_block.log | cut -c1-300
echo "== fps repeated x3 (default, block)"; for i in 1 2 3; do timeout 200 $E --case fps_1080p_p8_all 2>&1 | grep -a "encoder fps"; SVT_HIP_SYNC=block timeout 200 $E --case fps_1080p_p8_all 2>&1 | grep -a "encoder fps"; done
echo "== instances 4 / 8 (default, block)"
timeout 300 $E --case fps_1080p_p8_all --instances 4; SVT_HIP_SYNC=block timeout 300 $E --case fps_1080p_p8_all --instances 4
timeout 300 $E --case fps_1080p_p8_all --instances 8; SVT_HIP_SYNC=block timeout 300 $E --case fps_1080p_p8_all --instances 8
echo "== 300 frames"; timeout 300 $E --case fps_1080p_p8_all_300 2>&1 | grep -a "encoder fps"; SVT_HIP_SYNC=block timeout 300 $E --case fps_1080p_p8_all_300 2>&1 | grep -a "encoder fps"
nproc; cat /sys/fs/cgroup/cpu.max
