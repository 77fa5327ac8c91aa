#!/bin/bash
# round 2: deblocking seam + every stage seam together inside the reference encoder (bitstream identity), then encoder fps at 1080p with every seam
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-r02c24}; mkdir -p $O
timeout 2400 python tools/enc_identity.py --case dlfseam_p5_8bit,dlfseam_p2_10bit,dlfseam_p6_8bit_lp4,everyseam_p4_8bit_lp2,allseams_1080p_p6 --out $O/identity --timeout 900 > $O/identity.log 2>&1; echo "rc=$?"; grep -v "^Svt" $O/identity.log | tail -14
timeout 2400 python tools/enc_identity.py --case fps_1080p_p8_all,fps_1080p_p6_all,fps_1080p_p4_all --out $O/fps --timeout 900 > $O/fps.log 2>&1; echo "rc=$?"; grep -v "^Svt" $O/fps.log | tail -10
echo finished
