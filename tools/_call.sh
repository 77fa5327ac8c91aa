cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r03_call15; mkdir -p $O
timeout 600 python -m pytest tests/test_tpl.py tests/test_tf_picture.py tests/test_host_forms.py -q -m gpu -x > $O/pytest.txt 2>&1; tail -1 $O/pytest.txt
for i in 1 2 3; do timeout 900 python tools/enc_identity.py --case fps_1080p_p8_all_300,fps_1080p_p8_all --host avx2 --out /tmp/fps_avx2_$i > $O/fps_avx2_$i.log 2>&1; grep -a "identical=\|encoder fps\|MISMATCH\|returned" $O/fps_avx2_$i.log | cut -c1-100; grep -ao "'ms_in_stage_calls': [0-9]*\|'ms_in_me_pairs': [0-9]*" $O/fps_avx2_$i.log | tr '\n' ' '; echo; done
