cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_cdef_pick.py -m gpu -x -q 2>&1 | tail -2
python tools/microbench.py cdefchain --steps 5 2>&1 | tail -1 | cut -c1-300
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r38 -o run -- python tools/microbench.py cdefchain --steps 3 --warmup 1 > gpurun_out/r38_prof.log 2>&1; echo "prof rc=$?"
