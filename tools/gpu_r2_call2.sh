#!/bin/bash
# round 2, GPU call 2: ME seam in the encoder (identity + fps), exhaustive full-frame ME parity, the whole parity suite
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r02c2; mkdir -p $O
nproc > $O/nproc.txt; lscpu | grep -i "model name\|^CPU(s)" >> $O/nproc.txt
timeout 1500 python tools/enc_identity.py --case seam_p8_8bit,seam_p8_10bit_lp4,seam_p4_8bit_lp2,seam_p6_8bit_hook,seam_p10_8bit,seam_p2_8bit,fps_1080p_p8 --out gpurun_out/identity_seam --timeout 900 > $O/identity_seam.log 2>&1; echo "identity rc=$?"
grep -v SvtMalloc $O/identity_seam.log | tail -15
timeout 1200 python -m pytest tests -q -m gpu -x --deselect tests/test_encoder_identity.py > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest_gpu.txt
