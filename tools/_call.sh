cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
ulimit -c 0
O=gpurun_out/r06_final2; mkdir -p $O
( time SVT_HIP_FIXTURES=full timeout 2700 python -m pytest tests/test_ref_fixtures.py -q -m gpu ) > $O/pytest_fixtures_full.txt 2>&1; tail -4 $O/pytest_fixtures_full.txt
mkdir -p $O/fixtures_full; cp gpurun_out/ref_fixtures/*.json $O/fixtures_full/ 2>/dev/null
