#!/bin/bash
# split targets of the 4x4/stride-2 weight gradients (C4) and of the Linear weight gradients (LIN), separately
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05m2; mkdir -p $OUT
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
run() { MVK_TUNE=1 MVK_SPLITK_TARGET_C4=$2 MVK_SPLITK_TARGET_LIN=$3 timeout 600 python bench.py --config $1 --steps $4 --warmup $5 --no-cpu-baseline 2>>$OUT/ab.err | line $1_c4_$2_lin_$3; }
for i in 1 2 3; do
  for pair in "768 768" "256 768" "768 256" "256 256" "256 384"; do
    set -- $pair
    run cfg3 $1 $2 200 20
  done
done
for i in 1 2; do
  for pair in "768 768" "256 768" "768 256" "256 256"; do
    set -- $pair
    run cfg2 $1 $2 200 20
    run cfg4 $1 $2 10 3
  done
done
