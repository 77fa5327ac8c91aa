#!/bin/bash
# round 2: loop-restoration search stage -- parity on the GPU, timing, kernel statistics
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-r02c14}; mkdir -p $O
timeout 1200 python -m pytest tests/test_lr_search.py tests/test_restoration.py -q -m gpu -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
timeout 600 python tools/microbench.py lrsearch --steps 3 --warmup 1 > $O/lrsearch.json 2>$O/lrsearch.err; cat $O/lrsearch.json; tail -3 $O/lrsearch.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python tools/microbench.py lrsearch --steps 2 --warmup 1 > $O/stats.log 2>&1
python tools/pmc_summary.py ${1:-r02c14}_tmp $O/stats - - "python tools/microbench.py lrsearch --steps 2 --warmup 1" > /dev/null 2>&1; head -24 profiles/${1:-r02c14}_tmp_kernel_stats.txt | cut -c1-140; mv profiles/${1:-r02c14}_tmp_kernel_stats.txt $O/kernel_stats.txt
echo finished
