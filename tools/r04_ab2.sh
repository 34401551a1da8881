#!/bin/bash
# usage: tools/r04_ab2.sh variant...  pipelined step with 3, 4, 6 contexts (8 HIP hardware queues), 1000 steps each
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
for v in "$@"; do
  if [ $v = default ]; then unset WWHIP_LIB; else export WWHIP_LIB=$PWD/build_variants/libwwhip_$v.so; fi
  for c in 1 3 4 6; do
    echo "$v $(python3 tools/pipe_run.py $c 1000 2>&1 | grep -v amdgpu.ids | tail -1)" | tee -a gpurun_out/r04/ab2.log
  done
done
