# HBM traffic counters per kernel, one counter per pass (development tool; results under gpurun_out/pmct_*)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $R/gpurun_out/pmct_${c}_crnn -o run --output-format csv -- python3 $R/tools/kbench.py crnn 256 5 > $R/gpurun_out/pmct_${c}_crnn.log 2>&1 || echo "failed $c crnn"
  rocprofv3 --pmc $c -d $R/gpurun_out/pmct_${c}_wavenet -o run --output-format csv -- python3 $R/tools/kbench.py wavenet 256 5 bf16x3 > $R/gpurun_out/pmct_${c}_wavenet.log 2>&1 || echo "failed $c wavenet"
done
for c in FETCH_SIZE WRITE_SIZE; do for m in crnn wavenet; do python3 $R/tools/pmc_summary.py $R/gpurun_out/pmct_${c}_$m; done; done
