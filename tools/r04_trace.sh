#!/bin/bash
# kernel trace of the pipelined job (development): usage r04_trace.sh C tag
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r04
rocprofv3 --kernel-trace -d $R/gpurun_out/r04/tr_$2 -o run --output-format csv -- python3 $R/tools/pipe_run.py $1 400 > $R/gpurun_out/r04/tr_$2.log 2>&1
cat $R/gpurun_out/r04/tr_$2.log | grep -v amdgpu.ids
python3 $R/tools/trace_timeline.py $(find $R/gpurun_out/r04/tr_$2 -name "*kernel_trace.csv" | head -1) 400
