#!/bin/bash
# round 2, GPU call 9+: TF sub-pel refinement -- parity on the GPU, timing, instruction counters (output dir = $1, default r02c9)
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-r02c9}; mkdir -p $O
timeout 900 python -m pytest tests/test_tf_subpel.py -q -m gpu -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
timeout 300 python tools/microbench.py tfsubpel --steps 5 --warmup 1 > $O/tfsubpel.json 2>$O/tfsubpel.err; cat $O/tfsubpel.json; tail -3 $O/tfsubpel.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/insts -o i -- python tools/microbench.py tfsubpel --steps 3 --warmup 1 > $O/i.log 2>&1
python tools/pmc_dump.py $O/insts
echo finished
