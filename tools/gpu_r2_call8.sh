#!/bin/bash
# round 2, GPU call 8: LR (chunk 8 on the fast path) and the CDEF search with 1 / 2 / 4 level groups per workgroup
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r02c8; mkdir -p $O
timeout 900 python -m pytest tests/test_restoration.py tests/test_cdef.py tests/test_cdef_pick.py -q -m gpu -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
SVT_HIP_CDEF_GPW=4 timeout 600 python -m pytest tests/test_cdef.py -q -m gpu -x -k frame > $O/pytest_gpw4.txt 2>&1; echo "pytest gpw4 rc=$?"; tail -2 $O/pytest_gpw4.txt
timeout 300 python tools/microbench.py lr --steps 40 > $O/lr.json 2>$O/lr.err; python -c "
import json; d=json.load(open('$O/lr.json'))
for k,v in d.items(): print(k, round(v['ms']*1000,1),'us frac',round(v['roofline']['frac'],3))"
for g in 1 2 4; do SVT_HIP_CDEF_GPW=$g timeout 300 python tools/microbench.py cdef --steps 40 --no-parity-check 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('gpw $g', {k: round(1e6/v['frames_per_s'],1) for k,v in d.items()})"; done
for t in wiener; do
  SVT_LR_ONLY=$t timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/insts_$t -o i -- python tools/microbench.py lr --steps 10 > $O/i_$t.log 2>&1
  echo "== $t"; python tools/pmc_dump.py $O/insts_$t
done
for g in 1 2; do
SVT_HIP_CDEF_GPW=$g timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_cdef$g -o f -- python tools/microbench.py cdef --steps 10 --no-parity-check > $O/f_cdef$g.log 2>&1
SVT_HIP_CDEF_GPW=$g timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/insts_cdef$g -o i -- python tools/microbench.py cdef --steps 10 --no-parity-check > $O/i_cdef$g.log 2>&1
echo "== cdef gpw $g"; python tools/pmc_dump.py $O/fetch_cdef$g $O/insts_cdef$g
done
