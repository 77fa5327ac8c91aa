cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r06_call25; mkdir -p $O
run() { for i in 1 2; do timeout 600 python bench.py --legs lrsearch --no-cpu --no-pmc > $O/bench_$1.txt 2> $O/bench_$1_err.txt
python - <<PY
import json
d=json.load(open('gpurun_out/bench_detail.json'))
print('$1', ' '.join('%s %.3f ms' % (k, v.get('ms')) for k, v in d['kernels'].items()))
PY
done; }
run waves5_default
for w in 4 6; do
  sed -i "s/SVT_HIP_WAVES_PER_EU(5, 5) void lr_sgr_proj_kernel/SVT_HIP_WAVES_PER_EU($w, $w) void lr_sgr_proj_kernel/" svt-av1-psy_amd/csrc/lr_search.hip
  make -s -C svt-av1-psy_amd/csrc -j32 > $O/make_$w.txt 2>&1
  run waves$w
  sed -i "s/SVT_HIP_WAVES_PER_EU($w, $w) void lr_sgr_proj_kernel/SVT_HIP_WAVES_PER_EU(5, 5) void lr_sgr_proj_kernel/" svt-av1-psy_amd/csrc/lr_search.hip
done
