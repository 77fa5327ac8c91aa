cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r46_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r46_pytest.log
timeout 900 python bench.py --steps 20 --warmup 3 --extra > gpurun_out/r46_bench.json 2> gpurun_out/r46_bench.err; echo "bench rc=$?"
timeout 600 python bench.py > gpurun_out/r46_bench_default.json 2> gpurun_out/r46_bench_default.err; echo "bench default rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r46 -o run -- python bench.py --steps 20 --warmup 3 --no-cpu > gpurun_out/r46_prof.log 2>&1; echo "prof rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc46_fetch -o run -- python bench.py --steps 5 --warmup 1 --no-cpu > gpurun_out/r46_pmc_fetch.log 2>&1; echo "pmcf rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc46_write -o run -- python bench.py --steps 5 --warmup 1 --no-cpu > gpurun_out/r46_pmc_write.log 2>&1; echo "pmcw rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r46x -o run -- python bench.py --steps 10 --warmup 2 --no-cpu --extra > gpurun_out/r46_profx.log 2>&1; echo "profx rc=$?"
