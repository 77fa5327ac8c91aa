#!/bin/bash
# round 2: encoder fps at 1080p with the ME seam and the LR seam, presets 8 / 6 / 4 (C-only reference vs the same encoder with the stages on the MI355X)
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-r02c20}; mkdir -p $O
nproc; timeout 2400 python tools/enc_identity.py --case fps_1080p_p8,fps_1080p_p8_both,fps_1080p_p6_both,fps_1080p_p4_both --out $O/identity --timeout 900 > $O/identity.log 2>&1; echo "rc=$?"; grep -v "^Svt" $O/identity.log | tail -12
echo finished
