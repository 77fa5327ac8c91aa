cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/microbench.py mestage --steps 20 --warmup 3 2>&1 | tail -2
