cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_call2; mkdir -p $O
timeout 300 python -m pytest tests/test_tpl.py -q -m gpu -x > $O/pytest_tpl.txt 2>&1; tail -2 $O/pytest_tpl.txt
timeout 900 python tools/enc_identity.py --case tplseam_p8_8bit,tplseam_p6_8bit_lp4,tplseam_p4_8bit,tplseam_p10_8bit,tplseam_p8_10bit,tplseam_1080p_p8,tplseam_me_p8_8bit,everyseam_p4_8bit_lp2 --out $O/identity > $O/identity.log 2>&1; grep -v "^    \|^$" $O/identity.log | cut -c1-400 | tail -12
timeout 900 python tools/enc_identity.py --case fps_1080p_p8_all,fps_1080p_p6_all,fps_1080p_p8_me --host avx2 --out $O/fps > $O/fps.log 2>&1; cut -c1-900 $O/fps.log | tail -14
rm -f $O/identity/*.ivf $O/fps/*.ivf
nproc; grep -m1 "model name" /proc/cpuinfo
