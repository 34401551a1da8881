#!/bin/bash
# headline job only, default and the driver's command, with 4 and 8 HIP hardware queues (development)
mkdir -p gpurun_out/r04
for q in 4 8; do
  for p in 3 4 6; do
    export GPU_MAX_HW_QUEUES=$q
    python3 bench.py --no-extra --no-cpu-baseline --pipeline $p 2>/dev/null > gpurun_out/r04/bq_${q}_${p}.json
    python3 -c "
import json,sys; d=json.load(open('gpurun_out/r04/bq_${q}_${p}.json')); print('queues $q pipeline $p: us/step', round(d['ms_per_step']*1e3,2), 'single', round(d['single_stream']['ms_per_step']*1e3,2), 'regions', d['timed_regions']['min_ms'], d['timed_regions']['max_ms'])"
    python3 bench.py --no-extra --no-cpu-baseline --pipeline $p --steps 20 --warmup 5 2>/dev/null > gpurun_out/r04/bq20_${q}_${p}.json
    python3 -c "
import json,sys; d=json.load(open('gpurun_out/r04/bq20_${q}_${p}.json')); print('   steps 20: us/step', round(d['ms_per_step']*1e3,2), 'regions', d['timed_regions']['min_ms'], d['timed_regions']['max_ms'])"
  done
done
