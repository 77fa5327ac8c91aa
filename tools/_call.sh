cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r04_call14; mkdir -p $O
timeout 1200 python tools/enc_identity.py --case sweep --out /tmp/idt_sweep > $O/sweep.log 2>&1; grep -a "identical=\|ALL IDENT\|MISMATCH" $O/sweep.log | cut -c1-64
echo "== the same with the AVX-512 host build"; timeout 600 python tools/enc_identity.py --host avx512 --case tplrecon_p8_8bit,everyseam_p4_8bit_lp2,tfdriver_p8_10bit --out /tmp/idt512 2>&1 | grep -a "identical=\|ALL IDENT\|MISMATCH\|encoder fps" | cut -c1-120
