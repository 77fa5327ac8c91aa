cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_txfm.py -m gpu -x -q 2>&1 | tail -2
python tools/microbench.py txfm --steps 5 > gpurun_out/r19_micro.json 2> gpurun_out/r19_micro.err; echo "micro rc=$?"; tail -c 300 gpurun_out/r19_micro.err
