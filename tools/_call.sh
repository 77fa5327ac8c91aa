cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r04_call12; mkdir -p $O
E="python tools/enc_identity.py --host avx2 --out /tmp/idt"
for i in 1 2 3; do timeout 300 $E --case fps_1080p_p8_all_tplrecon > $O/enc$i.log 2>&1; grep -a "encoder fps" $O/enc$i.log | cut -c1-200; grep -ao "ms_in_stage_calls': [0-9]*, 'ms_first_stage_call': [0-9]*, 'ms_holding_device_lock': [0-9]*" $O/enc$i.log | head -1; done
timeout 300 $E --case fps_1080p_p10_all_tplrecon 2>&1 | grep -a "encoder fps" | cut -c1-200
rocm-smi --showmeminfo vram 2>/dev/null | head -5
