cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r03_call19; mkdir -p $O
timeout 900 python tools/enc_identity.py --case tfdriver_p8_8bit,tfdriver_p4_8bit,tfdriver_p6_8bit_lp4,tfdriver_p10_8bit,tfdriver_p4_10bit,tfdriver_1080p_p8,everyseam_p4_8bit_lp2,fps_1080p_p6_all --out /tmp/idt > $O/identity.log 2>&1; grep -av "^    \|^$\|^SVT_HIP" $O/identity.log | cut -c1-70 | tail -10
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_call19/bench_default.json").read().strip().split("\n")[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["valu_frac"])
for k in ("tf_picture_stage_1080p8_4refs_resident","tf_picture_stage_1080p8_4refs_host","tpl_src_stage_1080p8"): print(k, {x:v for x,v in d["kernels"][k].items() if x in ("us","ms","pictures_per_s","parity_checked_values")})
e=d["encoder_fps_1080p_preset8"]; print({k:e[k] for k in e if k.startswith("fps") or k.startswith("host_ms") or k in("bitstream_identical","steady_state_300_frames")})
PY
