#!/bin/bash
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05q; mkdir -p $OUT
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3; do
  timeout 600 python bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline 2>>$OUT/ab.err | line cfg5_ksplit
  MVK_TUNE=1 MVK_HEADS_KSPLIT=0 timeout 600 python bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline 2>>$OUT/ab.err | line cfg5_tiled
done
