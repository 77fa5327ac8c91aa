cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r03_call29; mkdir -p $O
timeout 120 python tools/enc_identity.py --host avx2 --case fps_1080p_p8_all,fps_1080p_p8_all_tplrecon,fps_1080p_p8_all,fps_1080p_p8_all_tplrecon --out /tmp/idt > $O/identity.log 2>&1; grep -a "identical=\|encoder fps" $O/identity.log | cut -c1-160; grep -ao "recon_pictures[^}]*" $O/identity.log | tail -2
