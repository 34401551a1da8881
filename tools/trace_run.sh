# rocprofv3 kernel trace of the default bench run + overlap histogram (development tool)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/trp -o run --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 400 > $R/gpurun_out/trp.json 2> $R/gpurun_out/trp.err
python3 $R/tools/trace_overlap.py $R/gpurun_out/trp/run_kernel_trace.csv 200
