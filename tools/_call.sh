cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r03_call14; mkdir -p $O
for i in 1 2 3 4; do timeout 900 python tools/enc_identity.py --case fps_1080p_p8_all_300,fps_1080p_p8_all --host avx2 --out /tmp/fps_avx2_$i > $O/fps_avx2_$i.log 2>&1; grep -a "identical=\|encoder fps\|MISMATCH\|returned" $O/fps_avx2_$i.log | cut -c1-100; grep -ao "'picture_buffers_page_locked': [0-9]*\|'ms_in_stage_calls': [0-9]*, 'ms_hashing_planes': [0-9]*, 'ms_first_stage_call': [0-9]*" $O/fps_avx2_$i.log | tr '\n' ' '; echo; done
timeout 600 python tools/enc_identity.py --case fps_1080p_p8_all_300,fps_1080p_p8_all --host c --out /tmp/fps_c > $O/fps_c.log 2>&1; grep -a "identical=\|encoder fps" $O/fps_c.log | cut -c1-100
