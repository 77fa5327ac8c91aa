#!/bin/bash
# round 2, GPU call 6: fused transform round trip (parity + config-3 chain vs fused), LR after the store rework
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r02c6; mkdir -p $O
timeout 900 python -m pytest tests/test_txfm.py tests/test_quant.py tests/test_restoration.py -q -m gpu -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
timeout 900 python tools/microbench.py txfm --steps 12 > $O/txfm.json 2>$O/txfm.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02c6/txfm.json'))['txfm_quant_roundtrip']
for k,v in d.items():
    print("%-10s chain %7.1f Mblk/s (%.3f of HBM)  fused %7.1f Mblk/s (%.3f of HBM at 10 B/px)  x%.2f" % (k, v['chain_Mblocks_s'], v['chain_hbm_frac'], v['fused_Mblocks_s'], v['fused_hbm_frac'], v['fused_speedup_vs_chain']))
PY
