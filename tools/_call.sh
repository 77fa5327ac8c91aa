cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r03_call22; mkdir -p $O

timeout 200 python tools/enc_identity.py --case screen_lowdelay_720p_p9 --out /tmp/idt > $O/identity.log 2>&1; grep -av "^$\|^SVT_HIP\|^Svt" $O/identity.log | cut -c1-1200 | tail -10
