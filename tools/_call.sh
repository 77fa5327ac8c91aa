cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r03_call8; mkdir -p $O
timeout 900 python -m pytest tests/test_tf_picture.py tests/test_hme.py tests/test_lr_search.py tests/test_host_forms.py -q -m gpu -x > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
timeout 1500 python tools/enc_identity.py --case tfdriver_p10_8bit,tfdriver_p8_10bit,tfdriver_p4_10bit,tfdriver_p8_8bit,everyseam_4k10_p8_lp1,lrseam_p4_8bit,lrseam_1080p_p6 --out /tmp/idt > $O/identity.log 2>&1; grep -av "^    \|^$\|^SVT_HIP" $O/identity.log | cut -c1-70 | tail -9; grep -ao "'pictures_filtered': [0-9]*, 'pictures_declined': [0-9]*, 'reference_frames': [0-9]*, 'pred_64x64': [0-9]*, 'pred_32x32': [0-9]*, 'pred_16x16': [0-9]*, 'pred_8x8': [0-9]*, 'early_exit_blocks': [0-9]*, 'last_decline': '[^']*'" $O/identity.log
timeout 900 python tools/enc_identity.py --case fps_1080p_p8_all,fps_1080p_p8_all_300,fps_1080p_p8_metf_300 --host avx2 --out /tmp/fps_avx2 > $O/fps_avx2.log 2>&1; grep -a "identical=\|encoder fps\|MISMATCH\|returned" $O/fps_avx2.log | cut -c1-150
