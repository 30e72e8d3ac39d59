#!/bin/bash
# max |dpre| tracking variants of the fused tail (g0 none, g1 per thread, g2 per wave and image): alone and in the step
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05u; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x -k "small_up" 2>&1 | tail -15
bash tools/suh_run.sh g0 g1 g2 2>&1 | tee $OUT/alone.txt
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac'])"; }
for i in 1 2 3; do
  MVK_LIB_PATH=$PWD/multivae_amd/libmvk_suh_g0.so MVK_TUNE=1 MVK_TAIL_BWD_F16=0 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line g0_bf16bwd
  MVK_LIB_PATH=$PWD/multivae_amd/libmvk_suh_g1.so timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line g1_f16bwd
  MVK_LIB_PATH=$PWD/multivae_amd/libmvk_suh_g2.so timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line g2_f16bwd
done
