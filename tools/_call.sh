cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r03_call23; mkdir -p $O
timeout 420 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_call23/bench_default.json").read().strip().split("\n")[-1])
print({k:d[k] for k in ("metric","value","unit","ms_per_step")}, d["roofline"], d["cpu_baseline"])
for k in ("tf_picture_stage_1080p8_4refs_resident","tf_picture_stage_1080p8_4refs_host","tpl_src_stage_1080p8"): print(k, {x:v for x,v in d["kernels"][k].items() if x in ("us","ms","pictures_per_s","parity_checked_values")})
e=d["encoder_fps_1080p_preset8"]; print({k:e[k] for k in e if k.startswith("fps") or k.startswith("host_ms") or k in("bitstream_identical","steady_state_300_frames")})
PY
