cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_call5; mkdir -p $O
timeout 600 python -m pytest tests/test_misc.py tests/test_lr_search.py tests/test_cdef.py -q -m gpu -x > $O/pytest_lr_cdef.txt 2>&1; tail -2 $O/pytest_lr_cdef.txt
timeout 300 python bench.py --steps 20 --warmup 5 --mode strips --no-cpu --no-pmc --legs none > $O/bench_strips.json 2> $O/bench_strips.err; tail -c 600 $O/bench_strips.err
for ar in 8x3 8x4 16x6; do timeout 300 python bench.py --only-me --no-cpu --steps 50 --warmup 5 --area $ar > $O/bench_me_$ar.json 2> $O/bench_me_$ar.err; done
timeout 300 python bench.py --no-cpu --legs lrsearch,cdef --steps 20 --warmup 5 --no-pmc > $O/bench_lrsearch.json 2> $O/bench_lrsearch.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_call5/bench_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f,"ERR",open(f.replace(".json",".err")).read()[-600:]); continue
    r=d.get("roofline",{})
    print(f.split("/")[-1], "value=%.1f ms=%.4f"%(d.get("value",0),d.get("ms_per_step",0)), {k:r.get(k) for k in ("frac","kernel_us","valu_frac","traffic","algorithmic_bytes_per_launch")})
    if "frame_partition" in d: print("   frame_partition", json.dumps(d["frame_partition"])[:900])
    for k,v in (d.get("kernels") or {}).items():
        rr=v.get("roofline",{})
        print("   ",k,"us=%.1f frac=%.3f"%(rr.get("kernel_us",0),rr.get("frac",0)), str(v.get("value"))[:12], v.get("unit"))
PY
cat > /tmp/pmc_me.txt <<'P'
pmc: SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU
pmc: SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
P
timeout 300 rocprofv3 --kernel-trace -i /tmp/pmc_me.txt -d $O/pmc_me8x3 --output-format csv -- python bench.py --only-me --no-cpu --no-parity-check --no-pmc --steps 5 --warmup 2 --area 8x3 > $O/pmc_me8x3.log 2>&1
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob("gpurun_out/r03_call5/pmc_me8x3/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"].split("(")[0]
        if "me_fullpel" not in k: continue
        acc[k][row["Counter_Name"]]+=float(row["Counter_Value"]); n[(k,row["Counter_Name"])]+=1
with open("gpurun_out/r03_call5/me8x3_counters.txt","w") as o:
    for k,v in acc.items():
        o.write(k+"\n")
        for c,x in sorted(v.items()): o.write("   %-24s %16.0f  (%d samples)\n"%(c,x,n[(k,c)]))
print(open("gpurun_out/r03_call5/me8x3_counters.txt").read())
PY
rm -rf $O/pmc_me8x3
timeout 1500 python tools/enc_identity.py --case fps_1080p_p8_all,fps_1080p_p8_all_300,fps_1080p_p8_metf_300,fps_1080p_p8_all_lp4,fps_1080p_p8_all_lp8,fps_1080p_p8_all_lp16 --host avx2 --out $O/fps > $O/fps.log 2>&1; cut -c1-900 $O/fps.log | tail -14
rm -f $O/fps/*.ivf
