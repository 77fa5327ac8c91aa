cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r04_final4; mkdir -p $O
timeout 420 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
