# SQ counters of the batched log-mel kernel, four counters per pass (development tool; results under gpurun_out/pmcq*)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $R/gpurun_out/pmcq$i -o run --output-format csv -- python3 $R/tools/fe_only.py 256 5 > $R/gpurun_out/pmcq$i.log 2>&1 || echo "pass $i failed"
done
for i in 1 2 3 4 5; do python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcq$i | grep logmel; done
