cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r06_call43; mkdir -p $O
# the TPL hand-off under uneven load: four test processes side by side on the one GPU (each 130 cases incl. 1080p / 4K pictures), beside a process that keeps the chip busy with the ME bench; three rounds
( timeout 600 python bench.py --only-me --steps 3000 --no-cpu --no-pmc > $O/load.txt 2>&1 ) &
LOADPID=$!
for r in 1 2 3; do
  for p in 1 2 3 4; do ( timeout 900 python -m pytest tests/test_tpl.py tests/test_tpl_full.py -q -m gpu -p no:cacheprovider > $O/stress_${r}_$p.txt 2>&1 ) & done
  wait %2 %3 %4 %5 2>/dev/null
  for p in 1 2 3 4; do tail -1 $O/stress_${r}_$p.txt; done
done
kill $LOADPID 2>/dev/null; wait 2>/dev/null
