# Round-6 profile set (development tool): rocprofv3 kernel-trace stats for the clip batch, the streaming tick in its three forms
# (one launch / two launches / Wavenet), the sliding evaluation and the at-scale evaluation; PMC passes (HBM traffic + SQ) of the
# clip batch and the one-launch tick.  Everything goes to gpurun_out/r06p; tools/pmc_collect.py turns the PMC directories into
# gpurun_out/r06p/summary/pmc_counters.json (copied to profiles/r06/).  Programs directly after `--`.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06p
mkdir -p $O/summary
STATS="--kernel-trace --stats --output-format csv"
WHAT=${1:-all}   # kt | pmc | all
if [ $WHAT = kt ] || [ $WHAT = all ]; then
rocprofv3 $STATS -d $O/kt_bench_single -o run -- python3 $R/bench.py --pipeline 1 --no-cpu-baseline --no-extra > $O/bench_single_under_rocprof.json 2> $O/kt_bench_single.err
echo "kt bench single done"
rocprofv3 $STATS -d $O/kt_bench_default -o run -- python3 $R/bench.py --no-cpu-baseline --no-extra > $O/bench_default_under_rocprof.json 2> $O/kt_bench_default.err
echo "kt bench default done"
rocprofv3 $STATS -d $O/kt_stream -o run -- python3 $R/tools/stream_prof.py > $O/stream_under_rocprof.txt 2> $O/kt_stream.err
echo "kt stream done"
rocprofv3 $STATS -d $O/kt_slide -o run -- python3 $R/tools/slide_throughput.py 10 > $O/slide_under_rocprof.json 2> $O/kt_slide.err
echo "kt slide done"
rocprofv3 $STATS -d $O/kt_eval_scale -o run -- python3 $R/tools/eval_at_scale.py 2529 2 > $O/eval_at_scale_under_rocprof.txt 2> $O/kt_eval_scale.err
echo "kt eval at scale done"
rocprofv3 --kernel-trace -d $O/kt_eval_lanes -o run --output-format csv -- python3 $R/tools/eval_at_scale.py 2529 2 > /dev/null 2> $O/kt_eval_lanes.err
python3 $R/tools/eval_lanes_trace.py $(find $O/kt_eval_lanes -name "*kernel_trace.csv" | head -1) > $O/summary/eval_two_lanes_overlap.txt || true
echo "kt eval at scale done"
rocprofv3 --kernel-trace -d $O/kt_pipe4 -o run --output-format csv -- python3 $R/tools/pipe_run.py 4 400 > $O/pipe4_under_rocprof.txt 2> $O/kt_pipe4.err
python3 $R/tools/trace_timeline.py $(find $O/kt_pipe4 -name "*kernel_trace.csv" | head -1) 400 > $O/summary/pipelined_4_contexts_timeline.txt
echo "kt pipelined timeline done"
for d in kt_bench_single kt_bench_default kt_stream kt_slide kt_eval_scale; do
  f=$(ls $O/$d/*/*kernel_stats.csv $O/$d/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/summary/${d}_kernel_stats.csv
done
cp $O/stream_under_rocprof.txt $O/eval_at_scale_under_rocprof.txt $O/summary/ 2>/dev/null || true
fi
if [ $WHAT = pmc ] || [ $WHAT = all ]; then
SQ1="SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQ2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD"
pmc() {  # name, counters, program...
  local name=$1 ctrs=$2; shift 2
  rocprofv3 --pmc $ctrs -d $O/pmc_$name -o run --output-format csv -- "$@" > $O/pmc_$name.log 2>&1 || echo "pmc pass $name failed"
  echo "pmc $name done"
}
for wl in "clips256 python3 $R/tools/kbench.py crnn 256 5" "wave256 python3 $R/tools/kbench.py wavenet 256 5 bf16x3" "wave256f python3 $R/tools/kbench.py wavenet 256 5" "stream128 python3 $R/tools/stream_prof.py"; do
  set -- $wl; name=$1; shift
  pmc ${name}_fetch FETCH_SIZE "$@"
  pmc ${name}_write WRITE_SIZE "$@"
  pmc ${name}_sq1 "$SQ1" "$@"
  pmc ${name}_sq2 "$SQ2" "$@"
done
cd $R
python3 tools/pmc_collect.py $O/summary/pmc_counters.json clips256=$O/pmc_clips256_fetch clips256=$O/pmc_clips256_write clips256=$O/pmc_clips256_sq1 clips256=$O/pmc_clips256_sq2 \
  wave256=$O/pmc_wave256_fetch wave256=$O/pmc_wave256_write wave256=$O/pmc_wave256_sq1 wave256=$O/pmc_wave256_sq2 \
  wave256f=$O/pmc_wave256f_fetch wave256f=$O/pmc_wave256f_write wave256f=$O/pmc_wave256f_sq1 wave256f=$O/pmc_wave256f_sq2 \
  stream128=$O/pmc_stream128_fetch stream128=$O/pmc_stream128_write stream128=$O/pmc_stream128_sq1 stream128=$O/pmc_stream128_sq2
fi
# the raw trace directories go (gpurun merges at most 64 MiB back): the summaries and the logs stay
cd $R
find $O -mindepth 1 -maxdepth 1 -type d ! -name summary -exec rm -rf {} +
ls $O/summary
