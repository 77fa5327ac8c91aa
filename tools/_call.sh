cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r03_call13; mkdir -p $O
for i in 1 2 3 4; do timeout 900 python tools/enc_identity.py --case fps_1080p_p8_all_300,fps_1080p_p8_all --host avx2 --out /tmp/fps_avx2_$i > $O/fps_avx2_$i.log 2>&1; grep -a "identical=\|encoder fps\|MISMATCH\|returned" $O/fps_avx2_$i.log | cut -c1-100; done
timeout 900 python tools/enc_identity.py --case fps_4k8_p8_all,fps_1080p_p6_all,fps_1080p_p4_all --host avx2 --out /tmp/fps_avx2_x > $O/fps_avx2_x.log 2>&1; grep -a "identical=\|encoder fps\|MISMATCH\|returned" $O/fps_avx2_x.log | cut -c1-100
