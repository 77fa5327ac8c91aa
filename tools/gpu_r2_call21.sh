#!/bin/bash
# round 2: CDEF apply seam + all three seams together inside the reference encoder (bitstream identity), then encoder fps at 1080p
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-r02c21}; mkdir -p $O
true
timeout 2400 python tools/enc_identity.py --case fps_1080p_p8_me,fps_1080p_p8_all,fps_1080p_p6_all,fps_1080p_p4_all --out $O/fps --timeout 900 > $O/fps.log 2>&1; echo "rc=$?"; grep -v "^Svt" $O/fps.log | tail -12
echo finished
