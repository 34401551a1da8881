#!/bin/bash
# Round 6: the evaluation flow at hey-snips size and at 16 x, chunks dealt to 1 / 2 / 3 / 4 lanes (contexts), on ONE box.
set -e
out=gpurun_out/r06/eval; mkdir -p $out
for l in 1 2 3 4; do
  WWHIP_EVAL_LANES=$l python tools/eval_share.py 7 > $out/share_lanes$l.json
  WWHIP_EVAL_LANES=$l python tools/eval_share.py 2 16 > $out/share16_lanes$l.json
done
for f in $out/share_lanes?.json $out/share16_lanes?.json; do
  python - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], "world1 min/median ms", d["world1"]["seconds_min"], d["world1"]["seconds_median"], "| share of 8", d["world8"]["seconds_min"], d["world8"]["seconds_median"])
PY
done
