cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc22_$tag -o run -- python tools/microbench.py lr --steps 2 --warmup 1 > gpurun_out/r22_pmc_$tag.log 2>&1; echo "pmc $tag rc=$?"
done
