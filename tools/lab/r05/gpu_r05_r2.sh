#!/bin/bash
# order of the decoder's two late weight gradients
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05r2; mkdir -p $OUT
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3 4; do
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line small_first
  MVK_TUNE=1 MVK_LATE_BIG_FIRST=1 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line big_first
done
