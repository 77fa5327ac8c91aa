#!/bin/bash
# Encoder-level runs on the GPU box (gpurun -- bash tools/gpu_encoder_seams.sh <tag>): bitstream identity of the reference encoder with every stage seam
# (ME, deblocking, CDEF search + apply, LR search + filter; alone, together, 1080p, 4K 10-bit at --lp 1), then encoder fps C-only vs seams at 1080p
# presets 8 / 6 / 4 and SURVEY's config 5 (4K 10-bit preset 8).  Outputs under gpurun_out/<tag>/; copy the logs you quote into profiles/.
TAG=${1:-seams}
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/$TAG; mkdir -p $O
CASES=$(python -c "import sys; sys.path.insert(0, 'tools'); import enc_identity as e; print(','.join(e.SEAM_CASES))")
timeout 3000 python tools/enc_identity.py --case $CASES --out $O/identity --timeout 900 > $O/identity.log 2>&1; echo "identity rc=$?"; grep -v "^Svt" $O/identity.log | grep "identical=\|IDENTICAL\|MISMATCH" | cut -c1-70
timeout 3000 python tools/enc_identity.py --case fps_1080p_p8_me,fps_1080p_p8_all,fps_1080p_p6_all,fps_1080p_p4_all,fps_4k10_p8_all --out $O/fps --timeout 1200 > $O/fps.log 2>&1; echo "fps rc=$? (fps_4k10_p8_all: the multi-threaded C-only reference is not reproducible at 10-bit preset 8 -- see reference_deterministic)"
grep -v "^Svt" $O/fps.log | grep "encoder fps\|identical=" | cut -c1-90
echo finished
