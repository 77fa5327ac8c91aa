cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
bash tools/gpu_regression.sh r04_final2
