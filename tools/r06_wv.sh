#!/bin/bash
# Round 6: the split-bf16 Wavenet's phase table and probes on ONE box (variants built by tools/build_variant.sh, see profiles/r06).
set -e
out=gpurun_out/r06/wv; mkdir -p $out
python tools/wv_probe.py time > $out/time_shipped.json
for v in nobar acc2 rcp1 acc2rcp1; do
  WWHIP_LIB=build_variants/libwwhip_wv_$v.so python tools/wv_probe.py time > $out/time_$v.json
done
python tools/wv_probe.py time > $out/time_shipped_again.json
WWHIP_LIB=build_variants/libwwhip_wv_stamps.so python tools/wv_probe.py stamps > $out/stamps.txt
WWHIP_LIB=build_variants/libwwhip_wv_stamps.so python tools/wv_probe.py time > $out/time_stamps_build_no_env.json
cat $out/time_*.json
head -40 $out/stamps.txt
