cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r03_call11; mkdir -p $O
timeout 600 python -m pytest tests/test_tf_picture.py tests/test_tf.py -q -m gpu -x > $O/pytest.txt 2>&1; tail -1 $O/pytest.txt
for h in c avx2; do timeout 900 python tools/enc_identity.py --case fps_1080p_p6_all,fps_1080p_p8_all --host $h --out /tmp/fps_$h > $O/fps_$h.log 2>&1; grep -a "identical=\|encoder fps\|MISMATCH\|returned" $O/fps_$h.log | cut -c1-130; grep -ao "'ms_in_stage_calls': [0-9]*\|'ms_in_me_pairs': [0-9]*\|'pictures_filtered': [0-9]*, 'pictures_declined': [0-9]*, 'reference_frames': [0-9]*\|'last_decline': '[^']*'" $O/fps_$h.log | tr '\n' ' '; echo; done
