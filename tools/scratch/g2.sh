cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/h2
timeout 300 python -m pytest tests/test_hme.py tests/test_sad.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/h2/pytest.txt
python tools/microbench.py hmechain mestage mesessionstage hme --steps 20 > gpurun_out/h2/bench.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d gpurun_out/h2 -o pmc -- python tools/microbench.py hmechain mesessionstage --steps 4 > gpurun_out/h2/pmc_log.txt 2>&1
