cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_restoration.py -m gpu -x -q 2>&1 | tail -3
python tools/microbench.py lr > gpurun_out/r18_micro.json 2> gpurun_out/r18_micro.err; echo "micro rc=$?"; tail -c 300 gpurun_out/r18_micro.err; cat gpurun_out/r18_micro.json | cut -c1-1500
