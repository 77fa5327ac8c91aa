cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r13_pytest.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r13_pytest.log
timeout 900 python bench.py --steps 20 --warmup 3 --extra > gpurun_out/r13_bench.json 2> gpurun_out/r13_bench.err; echo "bench rc=$?"
tail -c 800 gpurun_out/r13_bench.err
