cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r02sp; mkdir -p $O
timeout 2400 python tools/enc_identity.py --case tfsubpel_p8_8bit,tfsubpel_p4_8bit,tfsubpel_p6_8bit_lp4,tfsubpel_p2_10bit,everyseam_p4_8bit_lp2,allseams_1080p_p6 --out $O/identity --timeout 900 > $O/identity.log 2>&1; echo "rc=$?"; grep -v "^Svt" $O/identity.log | grep "identical=\|IDENT\|MISM" | cut -c1-60
grep -o "pairs_batched': [0-9]*, 'blocks_computed': [0-9]*, 'searches_served': [0-9]*, 'searches_by_reference': [0-9]*" $O/identity.log
timeout 2400 python tools/enc_identity.py --case fps_1080p_p8_all,fps_1080p_p6_all,fps_1080p_p4_all,fps_4k10_p8_all --out $O/fps --timeout 1200 > $O/fps.log 2>&1; echo "rc=$?"; grep -v "^Svt" $O/fps.log | grep "encoder fps\|identical=" | cut -c1-100
grep -o "pairs_batched': [0-9]*, 'blocks_computed': [0-9]*, 'searches_served': [0-9]*, 'searches_by_reference': [0-9]*" $O/fps.log
echo finished
