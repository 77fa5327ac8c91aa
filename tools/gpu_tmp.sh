cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r02tf2; mkdir -p $O
timeout 2400 python tools/enc_identity.py --case everyseam_p4_8bit_lp2,allseams_p5_8bit_lp2,allseams_1080p_p6,everyseam_4k10_p8_lp1 --out $O/identity --timeout 900 > $O/identity.log 2>&1; echo "rc=$?"; grep -v "^Svt" $O/identity.log | grep "identical=\|IDENT\|MISM" | cut -c1-60
timeout 2400 python tools/enc_identity.py --case fps_1080p_p8_me,fps_1080p_p8_all,fps_1080p_p6_all,fps_1080p_p4_all,fps_4k10_p8_all --out $O/fps --timeout 1200 > $O/fps.log 2>&1; echo "rc=$?"; grep -v "^Svt" $O/fps.log | grep "encoder fps\|identical=" | cut -c1-100
grep -o "tf_pairs_offloaded': [0-9]*, 'tf_sb_results': [0-9]*, 'tf_pairs_declined': [0-9]*" $O/fps.log | head -5
echo finished
