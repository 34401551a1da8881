#!/bin/bash
# front end A/B by rocprofv3 on ONE box: usage r05_fe_ab.sh variant...  (development)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05
for v in "$@"; do
  export WWHIP_LIB=$R/build_variants/libwwhip_$v.so
  for n in 256 4096; do
    rm -rf /tmp/kt_$v_$n
    rocprofv3 --kernel-trace --stats -d /tmp/kt_${v}_$n -o run --output-format csv -- python3 $R/tools/kbench.py crnn $n 60 > /dev/null 2>&1
    f=$(find /tmp/kt_${v}_$n -name "*kernel_stats.csv" | head -1)
    echo "$v n=$n $(python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "logmel" in r["Name"]: print(r["Name"][:40], "avg_us", round(float(r["AverageNs"])/1e3,2), "calls", r["Calls"])
PY
)" | tee -a $R/gpurun_out/r05/fe_ab_rocprof.log
  done
done
