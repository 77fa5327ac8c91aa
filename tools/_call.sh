cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r03_call16; mkdir -p $O
timeout 600 python -m pytest tests/test_tf_picture.py tests/test_tf.py tests/test_abi.py tests/test_rtcd_hook.py -q -m gpu -x > $O/pytest.txt 2>&1; tail -1 $O/pytest.txt
timeout 300 python bench.py --no-cpu --legs tfpic --steps 20 --warmup 5 --no-pmc > $O/bench_tfpic.json 2> $O/bench_tfpic.err; tail -c 400 $O/bench_tfpic.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_call16/bench_tfpic.json").read().strip().split("\n")[-1])
for k,v in d["kernels"].items(): print(k, {x:v[x] for x in v if x in("ms","us","pictures_per_s","equals_host_form")})
PY
true
python - <<'PY'
import csv,glob
for f in glob.glob("/tmp/prof_tf/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    with open("gpurun_out/r03_call16/tf_picture_kernel_stats.txt","w") as o:
        o.write("# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu --legs tfpic --steps 10 (MI355X): kernels of the temporal-filter picture stage, 1080p 8-bit, 4 frames\n")
        for r in rows[:16]:
            o.write("%-70s calls=%6s avg_us=%10.2f pct=%6s\n"%(r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
    print(open("gpurun_out/r03_call16/tf_picture_kernel_stats.txt").read())
PY
