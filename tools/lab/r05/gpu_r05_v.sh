#!/bin/bash
# fused tail publishing a bound from its largest row sum (g3) + the image layer's backward on fp16 pairs: tests and step A/B
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05v; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "small_up or fused or fullsize or svhn or mnistsvhn or trainer_with_hip_graph" 2>&1 | tail -5
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac'])"; }
for i in 1 2 3 4; do
  MVK_LIB_PATH=$PWD/multivae_amd/libmvk_suh_g0.so MVK_TUNE=1 MVK_TAIL_BWD_F16=0 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line g0_bf16bwd
  MVK_LIB_PATH=$PWD/multivae_amd/libmvk_suh_g3.so timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line g3_f16bwd
  MVK_LIB_PATH=$PWD/multivae_amd/libmvk_suh_g3.so MVK_TUNE=1 MVK_TAIL_BWD_F16=0 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line g3_bf16bwd
done
