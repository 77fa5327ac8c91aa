cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_call1; mkdir -p $O
timeout 600 python -m pytest tests/test_txfm.py tests/test_hme.py tests/test_host_forms.py tests/test_quant.py -q -m gpu -x > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 1500 $O/bench_default.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py --steps 20 --warmup 5 --no-cpu --no-parity-check --no-pmc > $O/stats.log 2>&1
find $O -name "*kernel_trace.csv" -size +8M -delete
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_call1/bench_default.json"))
print({k:d[k] for k in ("value","ms_per_step","steps")}, d["config"].get("launches_per_step"), d["config"].get("timed_region_s"))
print(json.dumps(d["roofline"])[:3000])
print(json.dumps(d["cpu_baseline"])[:1500])
print(json.dumps(d.get("encoder_fps_1080p_preset8"))[:800])
PY
