cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r06_call29; mkdir -p $O
run() { for i in 1 2; do timeout 600 python bench.py --legs tf,tfpic --no-cpu --no-pmc > $O/bench_$1.txt 2> $O/bench_$1_err.txt
python - <<PY
import json
d=json.load(open('gpurun_out/bench_detail.json'))
print('$1', ' '.join('%s %.1f us' % (k, (v.get('roofline') or {}).get('kernel_us') or v.get('ms',0)*1e3) for k, v in d['kernels'].items() if 'tf_' in k))
PY
done; }
timeout 600 python -m pytest tests/test_tf_subpel.py tests/test_tf_picture.py -q -m gpu > $O/pytest_tf.txt 2>&1; tail -2 $O/pytest_tf.txt
run slices
for w in 5 6; do
  sed -i "s/^__global__ __launch_bounds__(256) void tf_subpel_kernel(/__global__ __launch_bounds__(256) SVT_HIP_WAVES_PER_EU($w, $w) void tf_subpel_kernel(/" svt-av1-psy_amd/csrc/tf_subpel.hip
  make -s -C svt-av1-psy_amd/csrc -j32 > $O/make_$w.txt 2>&1; grep -c "tf_subpel" $O/make_$w.txt
  run waves$w
  sed -i "s/^__global__ __launch_bounds__(256) SVT_HIP_WAVES_PER_EU($w, $w) void tf_subpel_kernel(/__global__ __launch_bounds__(256) void tf_subpel_kernel(/" svt-av1-psy_amd/csrc/tf_subpel.hip
done
