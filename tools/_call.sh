cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r04_call7; mkdir -p $O
E="python tools/enc_identity.py --host avx2 --out /tmp/idt"
echo "== preset 10, 300 frames"; timeout 600 $E --case fps_1080p_p10_all_tplrecon_300 > $O/p10_300.log 2>&1; grep -a "encoder fps" $O/p10_300.log; grep -ao "identical=[A-Za-z]* (bitstream [A-Za-z]*)" $O/p10_300.log
echo "== default bench"
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$? bytes=$(wc -c < $O/bench_default.json)"
grep -v BENCH_DETAIL $O/bench_default.err | tail -5
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
cat $O/bench_default.json
echo "== full regression"
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
