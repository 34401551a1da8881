#!/bin/bash
# usage: tools/r04_ab.sh variant...   ("default" = the in-tree library); per-kernel times (single stream, HIP events), the
# front end at scale (4,096 clips) and the pipelined step (4 and 6 contexts, 1,000 steps) on one box (development)
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
for v in "$@"; do
  if [ $v = default ]; then unset WWHIP_LIB; else export WWHIP_LIB=$PWD/build_variants/libwwhip_$v.so; fi
  echo "== $v $(python3 tools/kbench.py crnn 256 200 2>&1 | grep -v amdgpu.ids | head -1) | 4096 clips: $(python3 tools/kbench.py crnn 4096 20 2>&1 | grep -v amdgpu.ids | head -1)" | tee -a gpurun_out/r04/ab.log
  for c in 4 6; do echo "   $(python3 tools/pipe_run.py $c 1000 2>&1 | grep -v amdgpu.ids | tail -1)" | tee -a gpurun_out/r04/ab.log; done
done
