cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_misc.py -m gpu -x -q 2>&1 | tail -3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r29 -o run -- python tools/microbench.py stats --steps 5 --warmup 1 > gpurun_out/r29_prof.log 2>&1; echo "prof rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc29 -o run -- python tools/microbench.py stats --steps 2 --warmup 1 > gpurun_out/r29_pmc.log 2>&1; echo "pmc rc=$?"
python tools/microbench.py stats > gpurun_out/r29_micro.json 2>/dev/null; cat gpurun_out/r29_micro.json | cut -c1-400
