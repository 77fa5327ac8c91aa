cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/h3
timeout 400 python -m pytest tests/test_hme.py tests/test_me_results.py tests/test_me_session.py tests/test_threads.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/h3/pytest.txt
python tools/microbench.py mesession mesessionstage --steps 20 > gpurun_out/h3/bench.txt 2>&1
python tools/microbench.py mesession mesessionstage --steps 20 >> gpurun_out/h3/bench.txt 2>&1
