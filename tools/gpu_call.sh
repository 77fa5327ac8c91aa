cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sad.py -m gpu -x -q -k "session or full_frame" 2>&1 | tail -2
python tools/microbench.py mesession 2>&1 | tail -1 | cut -c1-500
