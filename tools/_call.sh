cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r04_call19; mkdir -p $O
timeout 300 python -m pytest tests/test_tpl_full.py -q -m gpu > $O/pytest_tpl_full.txt 2>&1; tail -3 $O/pytest_tpl_full.txt
timeout 200 python bench.py --steps 20 --warmup 5 --legs tpl1 > $O/bench_tpl1.json 2> $O/bench_tpl1.err; tail -c 1500 $O/bench_tpl1.json; cp gpurun_out/bench_detail.json $O/bench_tpl1_detail.json 2>/dev/null
timeout 400 python tools/enc_identity.py --case tplrecon_p2_8bit,tplrecon_everyseam_p1_8bit,tplrecon_p8_8bit --out gpurun_out/identity > $O/identity.txt 2>&1; tail -8 $O/identity.txt
timeout 300 python -m pytest tests/test_tpl.py tests/test_tf_subpel.py -q -m gpu -x > $O/pytest_tpl_tfsubpel.txt 2>&1; tail -3 $O/pytest_tpl_tfsubpel.txt
