cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r20_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r20_pytest.log
python tools/microbench.py deblock lr picprep > gpurun_out/r20_micro.json 2> gpurun_out/r20_micro.err; echo "micro rc=$?"; tail -c 300 gpurun_out/r20_micro.err
