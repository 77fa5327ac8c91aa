cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r03_call10; mkdir -p $O
for i in 1 2; do timeout 900 python tools/enc_identity.py --case fps_1080p_p8_all,fps_1080p_p8_all_300 --host avx2 --out /tmp/fps_avx2_$i > $O/fps_avx2_$i.log 2>&1; grep -a "identical=\|encoder fps\|MISMATCH\|returned" $O/fps_avx2_$i.log | cut -c1-150; grep -ao "'ms_in_stage_calls': [0-9]*, 'ms_hashing_planes': [0-9]*, 'ms_first_stage_call': [0-9]*, 'ms_holding_device_lock': [0-9]*" $O/fps_avx2_$i.log; done
timeout 900 python tools/enc_identity.py --case fps_1080p_p8_all,fps_1080p_p6_all --host c --out /tmp/fps_c > $O/fps_c.log 2>&1; grep -a "identical=\|encoder fps\|MISMATCH\|returned" $O/fps_c.log | cut -c1-150
