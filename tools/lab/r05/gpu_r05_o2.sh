#!/bin/bash
# the 512-workgroup target of the tall-skinny split-K launches (MVK_SPLITK_TARGET_512)
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05o2; mkdir -p $OUT
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3; do
  for t in 512 256 384 768; do
    MVK_TUNE=1 MVK_SPLITK_TARGET_512=$t timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line cfg3_t512_$t
  done
done
for i in 1 2; do
  for t in 512 256; do
    MVK_TUNE=1 MVK_SPLITK_TARGET_512=$t timeout 600 python bench.py --config cfg2 --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line cfg2_t512_$t
    MVK_TUNE=1 MVK_SPLITK_TARGET_512=$t timeout 600 python bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline 2>>$OUT/ab.err | line cfg5_t512_$t
    MVK_TUNE=1 MVK_SPLITK_TARGET_512=$t timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>>$OUT/ab.err | line cfg4_t512_$t
  done
done
