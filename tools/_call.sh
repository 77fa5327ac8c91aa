cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r03_call6; mkdir -p $O
timeout 300 python bench.py --no-cpu --legs lrsearch --steps 20 --warmup 5 --no-pmc > $O/bench_lrsearch.json 2> $O/bench_lrsearch.err; tail -c 300 $O/bench_lrsearch.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_call6/bench_lrsearch.json"))
for k,v in (d.get("kernels") or {}).items(): print(k, json.dumps(v)[:700])
PY
for i in 1 2 3; do timeout 600 python tools/enc_identity.py --case fps_1080p_p8_all --host avx2 --out /tmp/fps$i > $O/fps_all_$i.log 2>&1; grep -a "refused\|returned\|identical\|encoder fps\|MISMATCH\|Abort\|abort" $O/fps_all_$i.log | cut -c1-400; done
timeout 300 python bench.py --only-me --no-cpu --steps 50 --warmup 5 --area 8x3 > $O/bench_me_8x3.json 2> $O/bench_me_8x3.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_call6/bench_me_8x3.json")); r=d["roofline"]
print({k:r.get(k) for k in ("frac","kernel_us","valu_frac","traffic","algorithmic_bytes_per_launch","traffic_source")}, r.get("traffic_detail"))
PY
cat > /tmp/pmc_me.txt <<'P'
pmc: SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU
pmc: SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
pmc: FETCH_SIZE WRITE_SIZE
P
timeout 300 rocprofv3 --kernel-trace -i /tmp/pmc_me.txt -d /tmp/pmc_me8x3 --output-format csv -- python bench.py --only-me --no-cpu --no-parity-check --no-pmc --steps 5 --warmup 2 --area 8x3 > $O/pmc_me8x3.log 2>&1
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob("/tmp/pmc_me8x3/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"].split("(")[0]
        if "me_fullpel" not in k: continue
        acc[k][row["Counter_Name"]]+=float(row["Counter_Value"]); n[(k,row["Counter_Name"])]+=1
with open("gpurun_out/r03_call6/me8x3_counters.txt","w") as o:
    o.write("# bench.py --only-me --area 8x3 (65 280 SB-refs per launch), rocprofv3 --pmc, sums over the launches sampled\n")
    for k,v in acc.items():
        o.write(k+"\n")
        for c,x in sorted(v.items()): o.write("   %-24s %16.0f  (%d launches)\n"%(c,x,n[(k,c)]))
print(open("gpurun_out/r03_call6/me8x3_counters.txt").read())
PY
tail -5 $O/pmc_me8x3.log | cut -c1-300
