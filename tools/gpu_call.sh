cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hme.py tests/test_me_results.py -m gpu -x -q 2>&1 | tail -3
python tools/microbench.py mestage mesessionstage --steps 20 --warmup 3 2>&1 | tail -1
