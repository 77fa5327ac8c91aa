cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r04_call9; mkdir -p $O
timeout 600 python -m pytest tests/test_tpl.py -q -m gpu -k resident > $O/pytest_tpl.txt 2>&1; tail -2 $O/pytest_tpl.txt
E="python tools/enc_identity.py --host avx2 --out /tmp/idt"
echo "== resident off / on x5 (60 frames)"; for i in 1 2 3 4 5; do SVT_HIP_TPL_RESIDENT=0 timeout 200 $E --case fps_1080p_p8_all_tplrecon 2>&1 | grep -a "encoder fps" | cut -c1-200; timeout 200 $E --case fps_1080p_p8_all_tplrecon 2>&1 | grep -a "encoder fps" | cut -c1-200; done
echo "== resident off / on x2 (300 frames)"; for i in 1 2; do SVT_HIP_TPL_RESIDENT=0 timeout 300 $E --case fps_1080p_p8_all_tplrecon_300 2>&1 | grep -a "encoder fps" | cut -c1-200; timeout 300 $E --case fps_1080p_p8_all_tplrecon_300 2>&1 | grep -a "encoder fps" | cut -c1-200; done
