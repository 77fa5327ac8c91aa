cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/microbench.py picprep 2>&1 | tail -2 | cut -c1-700
