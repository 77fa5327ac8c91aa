#!/bin/bash
# round 2: loop-restoration search seam inside the reference encoder -- bitstream identity on the GPU
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-r02c17}; mkdir -p $O
timeout 1500 python tools/enc_identity.py --case lrseam_p4_8bit,lrseam_p2_10bit,lrseam_p6_8bit_lp4,lrseam_me_p5_8bit_lp2,lrseam_1080p_p6,lrseam_p4_8bit_crf55,lrseam_p3_10bit_crf50 --out $O/identity --timeout 600 > $O/identity.log 2>&1; echo "rc=$?"; grep -v "^Svt" $O/identity.log | tail -12
echo finished
