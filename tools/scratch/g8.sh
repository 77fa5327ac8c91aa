cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/h8
timeout 600 python -m pytest tests/test_hme.py tests/test_sad.py tests/test_me_results.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/h8/pytest.txt
python tools/microbench.py hmechain mesessionstage --steps 20 2>&1 | grep "^{" > gpurun_out/h8/bench.txt
