cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r06_call22; mkdir -p $O
timeout 900 python -m pytest tests/test_misc.py -q -m gpu > $O/pytest_misc.txt 2>&1; tail -3 $O/pytest_misc.txt
