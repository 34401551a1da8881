set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01e
mkdir -p $O
cd $R
python bench.py > $O/bench_crnn.json 2> $O/bench_crnn.err
python bench.py --pipeline 1 --no-cpu-baseline > $O/bench_crnn_single.json 2> $O/bench_crnn_single.err
python bench.py --model wavenet --no-cpu-baseline > $O/bench_wave.json 2> $O/bench_wave.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof_crnn -o run --output-format csv -- python3 $R/bench.py --pipeline 1 --no-cpu-baseline > $O/bench_crnn_rocprof.json 2> $O/prof_crnn.err
rocprofv3 --kernel-trace --stats -d $O/prof_wave -o run --output-format csv -- python3 $R/bench.py --model wavenet --pipeline 1 --no-cpu-baseline > $O/bench_wave_rocprof.json 2> $O/prof_wave.err
ls -la $O $O/prof_crnn | head -40
tail -c 600 $O/bench_crnn.json
