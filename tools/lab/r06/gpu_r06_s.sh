#!/bin/bash
# r06 s: the partial flush of the deferred finishes on a bounded grid (it runs beside the step's last chain)
set -u
export TMPDIR=/tmp MVK_TUNE=1; OUT=gpurun_out/r06s; mkdir -p $OUT
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3; do
  for e in "-" "MVK_DEFER_FLUSH_GRID=256" "MVK_DEFER_FLUSH_GRID=512" "MVK_DEFER_FLUSH_GRID=1024" "MVK_DEFER_FLUSH_GRID=2048"; do
    envs=""; [ "$e" != "-" ] && envs="$e"
    env $envs timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line "[$e]" | tee -a $OUT/ab.txt
  done
done
MVK_DEFER_FLUSH_GRID=512 timeout 900 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_golden.py -q -x -k "fullsize or determin or reproduc or replay" 2>&1 | tail -2
