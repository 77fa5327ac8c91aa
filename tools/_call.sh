cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r06_call18; mkdir -p $O
timeout 300 python bench.py --probe > $O/probe.txt 2>&1; grep -E "cycles per wave64" $O/probe.txt
