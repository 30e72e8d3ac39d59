#!/bin/bash
# in-step A/B of the dense16 one-register-set forms (three alternating rounds on one box)
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05h; mkdir -p $OUT
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3 4; do
  for name in base fos bos; do
    MVK_LIB_PATH=$PWD/multivae_amd/libmvk_d16_$name.so python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>>$OUT/ab.err | line $name
  done
done 2>&1 | tee $OUT/ab_d16.txt
