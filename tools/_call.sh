cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r03_call24; mkdir -p $O
timeout 200 python -m pytest tests -q -m gpu -x --ignore=tests/test_encoder_identity.py > $O/pytest_gpu_kernels.txt 2>&1; tail -4 $O/pytest_gpu_kernels.txt
