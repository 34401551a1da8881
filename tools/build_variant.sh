#!/bin/bash
# Development: build libwwhip with extra defines into build_variants/libwwhip_<name>.so (git-ignored, shipped by gpurun) for
# A/B runs on one box:  WWHIP_LIB=build_variants/libwwhip_<name>.so python tools/kbench.py ...
# usage: tools/build_variant.sh <name> "<defines>" [files that see the defines, default: all]
set -e
cd "$(dirname "$0")/.."
name=$1; defs=$2; shift 2 || true
only=${@:-api.hip frontend.hip crnn.hip wavenet.hip posterior.hip streams.hip uploader.hip}
mkdir -p build_variants/obj_$name
objs=""
for f in api.hip frontend.hip crnn.hip wavenet.hip posterior.hip streams.hip uploader.hip; do
  o=build_variants/obj_$name/${f%.hip}.o
  if echo " $only " | grep -q " $f "; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result $defs -c wakeword-detection_amd/csrc/$f -o $o &
  else
    cp wakeword-detection_amd/csrc/build/${f%.hip}.o $o
  fi
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_variants/libwwhip_$name.so $objs
rm -rf build_variants/obj_$name
echo build_variants/libwwhip_$name.so
