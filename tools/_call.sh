cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r06_call8; mkdir -p $O
timeout 600 python -m pytest tests/test_misc.py tests/test_lr_search.py -q -m gpu -x > $O/pytest_stats.txt 2>&1; tail -3 $O/pytest_stats.txt
( time timeout 900 oracle/_ref/fixtures/SvtAv1HipFixtures --gtest_filter='HIP/av1_compute_stats_test*' ) > $O/fixtures_stats.txt 2>&1; grep -E "PASSED|FAILED|tests ran|real" $O/fixtures_stats.txt | head -8
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o s -- python bench.py --legs lrstats --no-cpu --no-pmc --no-parity-check > $O/kt.txt 2>&1
python - <<PY
import csv,glob
for f in glob.glob('$O/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'stats_' in r['Name']: print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
grep -o '"lr_compute_stats_4k10_win7": {[^}]*' gpurun_out/bench_detail.json | head -c 300; echo
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc1 -o s -- python $GRAFT_REPO_ROOT/bench.py --legs lrstats --no-cpu --no-pmc --no-parity-check > $GRAFT_REPO_ROOT/$O/pmc1.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc2 -o s -- python $GRAFT_REPO_ROOT/bench.py --legs lrstats --no-cpu --no-pmc --no-parity-check > $GRAFT_REPO_ROOT/$O/pmc2.txt 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,collections
for p in ('pmc1','pmc2'):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('gpurun_out/r06_call8/%s/**/*counter_collection.csv'%p, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'stats_' in r['Kernel_Name']:
                acc[r['Kernel_Name'][:50]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        for c,x in v.items(): print(p,k,c,sum(x)/len(x), len(x))
PY
# LR search: the timeline of one call per configuration (which kernel on which stream when)
rocprofv3 --kernel-trace --output-format csv -d $O/lrs -o s -- python bench.py --legs lrsearch --no-cpu --no-pmc --no-parity-check > $O/lrs.txt 2>&1
python - <<'PY'
import csv,glob
rows=[]
for f in glob.glob('gpurun_out/r06_call8/lrs/**/*kernel_trace.csv', recursive=True):
    rows+=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=[r for r in rows if not r['Kernel_Name'].startswith('void at::') and 'rate_kernel' not in r['Kernel_Name'] and 'probe' not in r['Kernel_Name']]
# split into calls at gaps > 300 us
calls=[];cur=[]
last=None
for r in rows:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if last is not None and s-last>300000: calls.append(cur);cur=[]
    cur.append(r); last=max(last or 0,e)
calls.append(cur)
print(len(calls),'bursts')
out=open('gpurun_out/r06_call8/lrs_timeline.txt','w')
for ci,c in enumerate(calls):
    if len(c)<8: continue
    t0=int(c[0]['Start_Timestamp']); t1=max(int(r['End_Timestamp']) for r in c)
    out.write('burst %d: %d kernels, %.1f us\n'%(ci,len(c),(t1-t0)/1e3))
    for r in c:
        out.write('  %9.1f %9.1f q%-3s %s grid=%s\n'%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r.get('Queue_Id','?'),r['Kernel_Name'].replace('(anonymous namespace)::','')[:60],r['Grid_Size_X']))
out.close()
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +2M -delete
grep -o '"lr_search_4k10_[a-z]*": {"[^,]*,[^,]*' gpurun_out/bench_detail.json
