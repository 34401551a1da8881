#!/bin/bash
# front-end kernel counters by rocprofv3 --pmc on ONE box: usage r05_fe_pmc.sh variant...  (development)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05
for v in "$@"; do
  export WWHIP_LIB=$R/build_variants/libwwhip_$v.so
  rm -rf /tmp/pm_$v
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_LDS -d /tmp/pm_$v -o run --output-format csv -- python3 $R/tools/kbench.py crnn 4096 3 > /dev/null 2>&1
  python3 - $v $(find /tmp/pm_$v -name "*counter_collection.csv" | head -1) <<'PY' | tee -a $R/gpurun_out/r05/fe_pmc.log
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[2])):
    if "logmel" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:28]][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(r["Kernel_Name"][:28],r["Counter_Name"])]+=1
for k,v in agg.items():
    d={c:x/n[(k,c)] for c,x in v.items()}
    w=d.get("SQ_WAVES",1)
    print(sys.argv[1], k, "waves", int(w), "valu/wave", round(d.get("SQ_INSTS_VALU",0)/w,1), "lds/wave", round(d.get("SQ_INSTS_LDS",0)/w,1), "salu/wave", round(d.get("SQ_INSTS_SALU",0)/w,1),
          "cycles", int(d.get("SQ_BUSY_CYCLES",0)/32), "bank_conflict_cycles", int(d.get("SQ_LDS_BANK_CONFLICT",0)), "active_lds", int(d.get("SQ_ACTIVE_INST_LDS",0)), "wait_lds", int(d.get("SQ_WAIT_INST_LDS",0)))
PY
done
