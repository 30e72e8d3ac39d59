#!/bin/bash
# split count of the small-batch weight gradients on the step's last chain (MVK_SPLITK_TARGET_1024: workgroups aimed at, default 768)
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05k2; mkdir -p $OUT
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3 4 5; do
  for t in 768 256 128 192; do
    MVK_TUNE=1 MVK_SPLITK_TARGET_1024=$t timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line target_$t
  done
done
