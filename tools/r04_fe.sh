#!/bin/bash
# round 4: front-end A/B on one box (development)
set -o pipefail
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "logmel or stft or framing" > $O/fe_tests.log 2>&1 || { tail -30 $O/fe_tests.log; exit 1; }
tail -3 $O/fe_tests.log
timeout -k 10 200 python tools/fe_check.py > $O/fe_check.log 2>&1 || { tail -30 $O/fe_check.log; exit 1; }
cat $O/fe_check.log
for v in default feold wpb1 wpb2; do
  if [ $v = default ]; then unset WWHIP_LIB; else export WWHIP_LIB=$PWD/build_variants/libwwhip_$v.so; fi
  echo "== $v" | tee -a $O/fe_ab.log
  timeout -k 10 200 python tools/kbench.py crnn 256 200 2>&1 | tee -a $O/fe_ab.log || exit 1
  timeout -k 10 200 python tools/kbench_pipe.py crnn 2>&1 | tee -a $O/fe_ab.log || exit 1
done
