#!/bin/bash
# round 2, GPU call 1: encoder bitstream identity with the hook on the MI355X + the whole parity suite
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r02c1; mkdir -p $O
timeout 1500 python tools/enc_identity.py --case all --out gpurun_out/identity --timeout 600 > $O/identity.log 2>&1; echo "identity rc=$?"
grep -v SvtMalloc $O/identity.log | tail -15
timeout 900 python -m pytest tests -q -m gpu -x --deselect tests/test_encoder_identity.py > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest_gpu.txt
