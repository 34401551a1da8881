# SQ counters of the at-scale CRNN kernels (crnn_rows_kernel, gru_tail_kernel) over 10 min of audio (development tool)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $R/gpurun_out/pmct$i -o run --output-format csv -- python3 $R/tools/slide_throughput.py 10 > $R/gpurun_out/pmct$i.log 2>&1 || echo "pass $i failed"
done
for i in 1 2 3 4; do python3 $R/tools/pmc_summary.py $R/gpurun_out/pmct$i | grep -i "tail\|rows"; done
