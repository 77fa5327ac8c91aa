#!/bin/bash
# round 2: CDEF apply seam + all three seams together inside the reference encoder (bitstream identity), then encoder fps at 1080p
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-r02c21}; mkdir -p $O
timeout 2400 python tools/enc_identity.py --case cdefseam_p8_8bit,cdefseam_p4_10bit,cdefseam_p6_8bit_lp4,allseams_p5_8bit_lp2,allseams_1080p_p6 --out $O/identity --timeout 900 > $O/identity.log 2>&1; echo "rc=$?"; grep -v "^Svt" $O/identity.log | tail -14
timeout 2400 python tools/enc_identity.py --case fps_1080p_p8_me,fps_1080p_p8_all,fps_1080p_p6_all,fps_1080p_p4_all --out $O/fps --timeout 900 > $O/fps.log 2>&1; echo "rc=$?"; grep -v "^Svt" $O/fps.log | tail -12
echo finished
