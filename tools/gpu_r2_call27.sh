#!/bin/bash
# round 2: SURVEY config 5 -- 4K 10-bit preset 8, 60 frames: C-only reference encoder vs the same encoder with every stage seam on the MI355X
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-r02c27}; mkdir -p $O
timeout 2400 python tools/enc_identity.py --case everyseam_4k10_p8_lp1 --out $O/fps --timeout 1200 > $O/fps.log 2>&1; echo "rc=$?"; grep -v "^Svt" $O/fps.log | tail -8
python -c "
import json; d=json.load(open('$O/fps/identity.json'))['cases'][0]
print({k: d.get(k) for k in ('identical','reference_deterministic','fps_c','fps_hip','seconds_c','seconds_hip')})"
echo finished
