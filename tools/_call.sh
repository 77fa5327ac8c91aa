cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r03_call20; mkdir -p $O
timeout 210 python -m pytest tests/test_hme.py tests/test_tf_picture.py -q -m gpu -x > $O/pytest_hme_tfpic.txt 2>&1; tail -3 $O/pytest_hme_tfpic.txt
timeout 170 python tools/enc_identity.py --case lowdelay_720p_p8_8bit,lowdelay_720p_p10_10bit,lowdelay_1080p_p9_lp4 --out /tmp/idt > $O/identity.log 2>&1; grep -av "^$\|^SVT_HIP\|^Svt" $O/identity.log | cut -c1-1500 | tail -12
