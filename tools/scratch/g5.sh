cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/h5
for s in 2 3 4; do echo "slots $s" >> gpurun_out/h5/bench.txt; SVT_BENCH_SLOTS=$s python tools/microbench.py mesessionstage --steps 20 2>&1 | grep "^{" >> gpurun_out/h5/bench.txt; done
