cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r02tf; mkdir -p $O
timeout 2400 python tools/enc_identity.py --case tfseam_p8_8bit,tfseam_p4_10bit,tfseam_p6_8bit_lp4 --out $O/identity --timeout 900 > $O/identity.log 2>&1; echo "rc=$?"; grep -v "^Svt" $O/identity.log | tail -9 | cut -c1-520
echo finished
