#!/bin/bash
# round 2, GPU call 7: LR frame kernel with row-uniform staging
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r02c7; mkdir -p $O
timeout 600 python -m pytest tests/test_restoration.py -q -m gpu -x > $O/pytest_lr.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_lr.txt
timeout 300 python tools/microbench.py lr --steps 40 > $O/lr.json 2>$O/lr.err; python -c "
import json; d=json.load(open('$O/lr.json'))
for k,v in d.items(): print(k, round(v['ms']*1000,1),'us frac',round(v['roofline']['frac'],3))"
for t in wiener sgrproj; do
  SVT_LR_ONLY=$t timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/insts_$t -o i -- python tools/microbench.py lr --steps 10 > $O/i_$t.log 2>&1
  SVT_LR_ONLY=$t timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --output-format csv -d $O/cyc_$t -o c -- python tools/microbench.py lr --steps 10 > $O/c_$t.log 2>&1
  echo "== $t"; python tools/pmc_dump.py $O/insts_$t $O/cyc_$t
done
