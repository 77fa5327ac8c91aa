cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r04_call10; mkdir -p $O
for s in 1 3; do SVT_BENCH_SLOTS=$s timeout 300 python bench.py --steps 20 --warmup 5 --legs session --no-cpu --no-pmc > $O/sess$s.json 2> $O/sess$s.err; python - <<PY
import json
d=json.load(open('gpurun_out/bench_detail.json'))
for n in ('me_session_stage_1080p_host','me_session_stage_1080p_host_preset8'):
    k=d['kernels'][n]; print("slots $s", n, round(k['us_per_picture'],1), 'us/picture; host submit', round(k['host_submit_us_per_picture'],1))
PY
done
SVT_BENCH_SLOTS=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr -o t -- python bench.py --steps 20 --warmup 5 --legs session --no-cpu --no-pmc > $O/tr.log 2>&1
python - <<'PY'
import csv, glob
k=glob.glob('gpurun_out/r04_call10/tr/**/*kernel_trace.csv', recursive=True); m=glob.glob('gpurun_out/r04_call10/tr/**/*memory_copy_trace.csv', recursive=True)
ev=[]
for r in csv.DictReader(open(k[0])):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ','').replace('(anonymous namespace)::','')[:40]))
for r in csv.DictReader(open(m[0])):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY '+r.get('Direction', r.get('Name',''))[:24]))
ev.sort()
# the last picture-sized upload marks a stage: print the 45 events after a late 2.8 MB H2D copy
idx=[i for i,e in enumerate(ev) if e[2].startswith('COPY') and (e[1]-e[0])>40000]
i0=idx[len(idx)*3//4] if idx else 0
t0=ev[i0][0]
for s,e,n in ev[i0:i0+48]:
    print("%8.1f us  +%6.1f  %s" % ((s-t0)/1e3, (e-s)/1e3, n))
PY
rm -rf $O/tr
