# usage: bash tools/pmc_run.sh <tag> <python script and args...>   (development tool: SQ counters per kernel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $R/gpurun_out/pmc_${tag}_$i -o run --output-format csv -- python3 $R/"$@" > $R/gpurun_out/pmc_${tag}_$i.log 2>&1 || echo "pass $i failed"
done
for i in 1 2 3 4 5; do python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${tag}_$i; done
