# SQ counters of the Wavenet kernel (split-bf16 unless "fp32" is given as the 2nd argument) at 256 windows, four counters per pass (development tool; results under gpurun_out/pmcw*)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-256}
P=${2:-bf16x3}
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $R/gpurun_out/pmcw$i -o run --output-format csv -- python3 $R/tools/kbench.py wavenet $N 5 x $P > $R/gpurun_out/pmcw$i.log 2>&1 || echo "pass $i failed"
done
for i in 1 2 3 4 5 6; do python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcw$i | grep -i "wavenet"; done
