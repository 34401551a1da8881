# usage: bash tools/pipe_sweep.sh [extra bench.py flags]   (development tool: throughput vs pipelined contexts)
for p in 1 2 3 4; do python bench.py --pipeline $p --no-cpu-baseline --steps 300 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipeline', $p, round(d['value']/1e6,1), 'M frames/s', round(d['ms_per_step']*1e3,2), 'us/step')"; done
