cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_call3; mkdir -p $O
timeout 300 python -m pytest tests/test_sad.py tests/test_cdef.py -q -m gpu -x > $O/pytest_sad_cdef.txt 2>&1; tail -2 $O/pytest_sad_cdef.txt
for form in 0 1; do SVT_HIP_SAD_FORM=$form timeout 300 python bench.py --no-cpu --legs sad --steps 20 --warmup 5 > $O/ab_sad_form$form.json 2> $O/ab_sad_form$form.err; done
for m in 3 4 2; do SVT_HIP_CDEF_MINB=$m timeout 300 python bench.py --no-cpu --legs cdef --steps 20 --warmup 5 > $O/ab_cdef_minb$m.json 2> $O/ab_cdef_minb$m.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_call3/ab_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f,"ERR",open(f.replace(".json",".err")).read()[-400:]); continue
    for k,v in d["kernels"].items():
        r=v.get("roofline",{})
        print(f.split("/")[-1], k, "us=%.1f frac=%.3f traffic=%s alg=%s src=%s"%(r.get("kernel_us",0), r.get("frac",0), r.get("traffic"), r.get("algorithmic_bytes_per_launch"), (r.get("traffic_detail") or {}).get("source","")[:12]))
PY
timeout 900 python tools/enc_identity.py --case dlfseam_sb_p8_8bit,dlfseam_sb_p8_8bit_lp4,dlfseam_sb_p10_10bit,dlfseam_sb_1080p_p8,tplseam_me_p8_8bit --out $O/identity > $O/identity.log 2>&1; grep -v "^    \|^$" $O/identity.log | cut -c1-330 | tail -7
timeout 900 python tools/enc_identity.py --case fps_1080p_p8_all,fps_1080p_p8_all_300,fps_1080p_p8_me --host avx2 --out $O/fps > $O/fps.log 2>&1; cut -c1-1400 $O/fps.log | tail -8
rm -f $O/identity/*.ivf $O/fps/*.ivf
