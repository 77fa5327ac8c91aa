cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sad.py -m gpu -x -q 2>&1 | tail -2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r33 -o run -- python tools/microbench.py hme --steps 5 --warmup 1 > gpurun_out/r33_prof.log 2>&1; echo "prof rc=$?"
python tools/microbench.py hme 2>/dev/null | cut -c1-600
