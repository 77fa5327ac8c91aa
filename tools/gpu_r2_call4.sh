#!/bin/bash
# round 2, GPU call 4: one-wave-per-item ME kernel -- parity, bench, instruction counters
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r02c4; mkdir -p $O
timeout 900 python -m pytest tests/test_sad.py tests/test_hme.py tests/test_rtcd_hook.py -q -m gpu -x > $O/pytest_me.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_me.txt
timeout 300 python bench.py --only-me --no-cpu --steps 200 > $O/bench_me.json 2> $O/bench_me.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02c4/bench_me.json')); r=d['roofline']
print("value",d['value'],"ms",d['ms_per_step'],"valu_frac",r['valu_frac'],"hbm frac",r['frac'], "fp", d['frame_partition']['ms_per_step'])
PY
for a in 8x4 8x3 16x6 24x12; do timeout 120 python bench.py --only-me --no-cpu --steps 100 --area $a 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$a', round(d['value']), 'Mblk/s', round(d['ms_per_step'],4),'ms valu_frac', round(r['valu_frac'],3))"; done
P="--steps 30 --warmup 5 --no-cpu --no-parity-check --only-me"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py $P > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/insts -o i -- python bench.py $P > $O/insts.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --output-format csv -d $O/cyc -o c -- python bench.py $P > $O/cyc.log 2>&1
python tools/pmc_dump.py $O/insts $O/cyc 2>/dev/null | head -40
