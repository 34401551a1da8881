# Round-end profile set (development tool): bench lines + rocprofv3 kernel stats + HBM traffic counters -> gpurun_out/r02e
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02e
mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
cd /tmp
# kernel stats of the bench command itself (4 pipelined contexts: averages inflated by overlap) and of the single-stream form
rocprofv3 --kernel-trace --stats -d $O/prof_default -o run --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra > $O/bench_default_rocprof.json 2> $O/prof_default.err
rocprofv3 --kernel-trace --stats -d $O/prof_single -o run --output-format csv -- python3 $R/bench.py --pipeline 1 --no-cpu-baseline --no-extra > $O/bench_single_rocprof.json 2> $O/prof_single.err
rocprofv3 --kernel-trace --stats -d $O/prof_wave -o run --output-format csv -- python3 $R/bench.py --model wavenet --pipeline 1 --no-cpu-baseline --no-extra > $O/bench_wave_rocprof.json 2> $O/prof_wave.err
# HBM-side traffic, one counter per pass
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $O/pmct_${c}_crnn -o run --output-format csv -- python3 $R/tools/kbench.py crnn 256 5 > $O/pmct_${c}_crnn.log 2>&1 || echo "failed $c crnn"
  rocprofv3 --pmc $c -d $O/pmct_${c}_wavenet -o run --output-format csv -- python3 $R/tools/kbench.py wavenet 256 5 bf16x3 > $O/pmct_${c}_wavenet.log 2>&1 || echo "failed $c wavenet"
done
for c in FETCH_SIZE WRITE_SIZE; do for m in crnn wavenet; do python3 $R/tools/pmc_summary.py $O/pmct_${c}_$m; done; done
ls $O $O/prof_single | head -40
cd $R
python tools/slide_throughput.py 10 > $O/slide_throughput_10min.json 2> /dev/null
bash tools/pmc_crnn.sh 256 > $O/sq_crnn_256.txt 2>&1
bash tools/pmc_crnn.sh 4096 > $O/sq_crnn_4096.txt 2>&1
bash tools/pmc_fe.sh > $O/sq_logmel_256.txt 2>&1
bash tools/pmc_wavenet.sh 256 > $O/sq_wavenet_bf16x3_256.txt 2>&1
for f in sq_crnn_256 sq_crnn_4096 sq_logmel_256 sq_wavenet_bf16x3_256; do echo $f; tail -n 6 $O/$f.txt; done
