cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r03_call7; mkdir -p $O
timeout 600 python -m pytest tests/test_tf_picture.py tests/test_tf_subpel.py -q -m gpu -x > $O/pytest_tf.txt 2>&1; tail -2 $O/pytest_tf.txt
timeout 1200 python tools/enc_identity.py --case tfdriver_p8_8bit,tfdriver_p4_8bit,tfdriver_p6_8bit_lp4,tfdriver_p2_8bit,tfdriver_p10_8bit,tfdriver_1080p_p8,tfdriver_p8_10bit,everyseam_p4_8bit_lp2 --out /tmp/idt > $O/identity_tfdriver.log 2>&1; grep -av "^    \|^$" $O/identity_tfdriver.log | cut -c1-60 | tail -10; grep -ao "'pictures_filtered': [0-9]*, 'pictures_declined': [0-9]*, 'reference_frames': [0-9]*, 'pred_64x64': [0-9]*, 'pred_32x32': [0-9]*, 'pred_16x16': [0-9]*, 'pred_8x8': [0-9]*, 'early_exit_blocks': [0-9]*, 'last_decline': '[^']*'" $O/identity_tfdriver.log
for h in avx2 c; do timeout 900 python tools/enc_identity.py --case fps_1080p_p8_all,fps_1080p_p8_all_300,fps_1080p_p8_metf_300 --host $h --out /tmp/fps_$h > $O/fps_$h.log 2>&1; grep -a "identical=\|encoder fps\|MISMATCH\|returned" $O/fps_$h.log | cut -c1-150; done
for ar in 8x3 8x4; do timeout 300 python bench.py --only-me --no-cpu --steps 50 --warmup 5 --area $ar > $O/bench_me_$ar.json 2> $O/bench_me_$ar.err; python - <<PY
import json
d=json.load(open("$O/bench_me_$ar.json")); r=d["roofline"]
print("$ar", {k:r.get(k) for k in ("frac","kernel_us","valu_frac","traffic","algorithmic_bytes_per_launch")}, (r.get("traffic_detail") or {}).get("read"), (r.get("traffic_detail") or {}).get("write"))
PY
done
