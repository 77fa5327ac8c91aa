bash tools/gpu_regression.sh r06_final6
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06_final6
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
