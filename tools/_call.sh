bash tools/gpu_regression.sh r06_final5
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06_final5
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
for i in 1 2; do timeout 1500 python -m pytest tests/test_tpl.py tests/test_tpl_full.py -q -m gpu > $O/pytest_tpl_again_$i.txt 2>&1; tail -1 $O/pytest_tpl_again_$i.txt; done
