cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/microbench.py mesessionstage --steps 12 --warmup 3 > gpurun_out/c60.out 2> gpurun_out/c60.err; echo "rc=$?"
tail -c 1500 gpurun_out/c60.out; tail -c 1500 gpurun_out/c60.err
