#!/bin/bash
# r06 j: first decoder layer with 4 rows in flight per thread (MVK_SK_UNROLL) vs the round-5 loop; cfg4 / cfg5 after the igemm_bf body refactor
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r06j; mkdir -p $OUT
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3 4; do
  MVK_LIB_PATH=$PWD/multivae_amd/libmvk_sk1.so timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line unroll1 | tee -a $OUT/ab.txt
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line unroll4 | tee -a $OUT/ab.txt
done
for c in cfg4 cfg5; do for i in 1 2; do
  timeout 900 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>>$OUT/ab.err | line $c | tee -a $OUT/ab.txt
done; done
grep -v amdgpu.ids $OUT/ab.err | grep -v "^  File\|^    " | tail -3
