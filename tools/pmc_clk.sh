cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $R/gpurun_out/pmcclk -o run --output-format csv -- python3 $R/tools/fe_only.py 256 5 > $R/gpurun_out/pmcclk.log 2>&1 || echo "pmc failed"
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcclk | grep logmel
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ktfe -o run --output-format csv -- python3 $R/tools/fe_only.py 256 20 > $R/gpurun_out/ktfe.log 2>&1 || echo "kt failed"
grep -h logmel $R/gpurun_out/ktfe/*kernel_stats.csv
