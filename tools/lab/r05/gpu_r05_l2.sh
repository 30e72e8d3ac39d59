#!/bin/bash
# MVK_SPLITK_TARGET_1024 = 256 against 768 on the other configurations
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05l2; mkdir -p $OUT
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3; do
  for t in 768 256; do
    MVK_TUNE=1 MVK_SPLITK_TARGET_1024=$t timeout 600 python bench.py --config cfg2 --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line cfg2_$t
    MVK_TUNE=1 MVK_SPLITK_TARGET_1024=$t timeout 600 python bench.py --config cfg3k1 --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line cfg3k1_$t
  done
done
for i in 1 2; do
  for t in 768 256; do
    MVK_TUNE=1 MVK_SPLITK_TARGET_1024=$t timeout 600 python bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline 2>>$OUT/ab.err | line cfg5_$t
    MVK_TUNE=1 MVK_SPLITK_TARGET_1024=$t timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>>$OUT/ab.err | line cfg4_$t
  done
done
