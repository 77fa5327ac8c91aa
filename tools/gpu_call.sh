cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/microbench.py deblock --steps 20 --warmup 3 2>/dev/null
