#!/bin/bash
# round 4: the tails after the common association (development)
mkdir -p gpurun_out/r04
for v in "$@"; do
  if [ $v = default ]; then unset WWHIP_LIB; else export WWHIP_LIB=$PWD/build_variants/libwwhip_$v.so; fi
  echo "== $v" | tee -a gpurun_out/r04/tails.log
  python3 tools/tail_sweep.py 256 512 1024 2048 4096 8192 9216 12288 16384 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04/tails.log
  python3 tools/slide_throughput.py 10 models=crnn crnn_tail_mfma=2 2>&1 | grep -v amdgpu.ids | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slide 10 min tail16:', d['CRNN']['kernel_ms'])" | tee -a gpurun_out/r04/tails.log
  python3 tools/slide_throughput.py 10 models=crnn crnn_tail_mfma=0 2>&1 | grep -v amdgpu.ids | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slide 10 min valu tail:', d['CRNN']['kernel_ms'])" | tee -a gpurun_out/r04/tails.log
  python3 tools/kbench.py crnn 256 200 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04/tails.log
done
