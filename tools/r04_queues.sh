#!/bin/bash
for q in 4 8; do
  export GPU_MAX_HW_QUEUES=$q
  for c in 3 4 5 6 8; do
    echo "GPU_MAX_HW_QUEUES=$q $(python3 tools/pipe_run.py $c 1000 2>&1 | grep -v amdgpu.ids | tail -1)"
  done
done
