#!/bin/bash
# round 2: where the ME seam spends its time (stage calls under the lock, plane hashing) at 1080p preset 8
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-r02c29}; mkdir -p $O
timeout 1200 python tools/enc_identity.py --case fps_1080p_p8_me,fps_1080p_p8_all --out $O/fps --timeout 600 > $O/fps.log 2>&1; echo "rc=$?"; grep -v "^Svt" $O/fps.log | tail -8 | cut -c1-700
echo finished
