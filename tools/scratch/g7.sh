cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/h7
timeout 600 python -m pytest tests/test_hme.py tests/test_sad.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/h7/pytest.txt
python tools/microbench.py mesessionstage --steps 20 2>&1 | grep "^{" > gpurun_out/h7/bench.txt
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/h7 -o p8 -- python tools/microbench.py mesessionstage --steps 8 > gpurun_out/h7/log.txt 2>&1
