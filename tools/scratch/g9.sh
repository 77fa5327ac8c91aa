cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/h9
python tools/microbench.py mesession --steps 20 2>&1 | grep "^{" > gpurun_out/h9/bench.txt
python tools/microbench.py mesession --steps 20 2>&1 | grep "^{" >> gpurun_out/h9/bench.txt
