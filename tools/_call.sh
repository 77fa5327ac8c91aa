cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r04_call5; mkdir -p $O
E="python tools/enc_identity.py --host avx2 --out /tmp/idt --cpu-stats"
for c in fps_1080p_p6_all_tplrecon fps_1080p_p4_all_tplrecon fps_1080p_p10_all_tplrecon fps_4k8_p8_all_tplrecon; do
  echo "== $c"; timeout 600 $E --case $c > $O/$c.log 2>&1; grep -a "identical=\|encoder fps\|stage CPU" $O/$c.log | cut -c1-100,300-520 | cut -c1-400
done
echo "== fps_4k10_p8_all_tplrecon (identity provable at --lp 1 only: fps and CPU reported)"; timeout 900 $E --case fps_4k10_p8_all_tplrecon > $O/fps_4k10.log 2>&1; grep -a "identical=\|encoder fps\|stage CPU" $O/fps_4k10.log | cut -c1-400
