#!/bin/bash
# usage: tools/r04_ab.sh variant...   ("default" = the in-tree library); per-kernel times + pipelined step on one box
set -o pipefail
mkdir -p gpurun_out/r04
for v in "$@"; do
  if [ $v = default ]; then unset WWHIP_LIB; else export WWHIP_LIB=$PWD/build_variants/libwwhip_$v.so; fi
  echo "== $v" | tee -a gpurun_out/r04/ab.log
  timeout -k 10 200 python tools/kbench.py crnn 256 200 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04/ab.log || exit 1
  timeout -k 10 200 python tools/kbench_pipe.py crnn 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04/ab.log || exit 1
done
