cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/h4
timeout 600 python -m pytest tests/test_hme.py tests/test_me_results.py tests/test_sad.py tests/test_threads.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/h4/pytest.txt
for i in 1 2 3; do timeout 300 python -m pytest tests/test_hme.py tests/test_sad.py -q -m gpu -x -k "in_flight or ring_reuse or session" 2>&1 | tail -2 >> gpurun_out/h4/pytest.txt; done
