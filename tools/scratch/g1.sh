cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/h1
python tools/microbench.py hmechain mestage mesessionstage --steps 20 > gpurun_out/h1/bench.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d gpurun_out/h1 -o pmc -- python tools/microbench.py hmechain --steps 4 > gpurun_out/h1/pmc_log.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d gpurun_out/h1 -o pmc2 -- python tools/microbench.py hmechain --steps 4 > gpurun_out/h1/pmc2_log.txt 2>&1
ls gpurun_out/h1
