cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/h6
timeout 600 python -m pytest tests/test_hme.py tests/test_sad.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/h6/pytest.txt
python tools/microbench.py hmechain mestage mesessionstage hme mesession --steps 20 2>&1 | grep "^{" > gpurun_out/h6/bench.txt
python bench.py 2>&1 | tail -1 > gpurun_out/h6/bench_default.txt
