#!/bin/bash
# 3x3 layers with 256 input channels as two 128-channel launches of the register-stationary kernels: tests + cfg4 / cfg5 A/B
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05b2; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "conv3x3 or resnet or jmvae_cub or cfg4 or cfg5 or mmvaeplus or polymnist" 2>&1 | tail -5
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3; do
  timeout 600 python bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline 2>>$OUT/ab.err | line cfg5_split256
  MVK_TUNE=1 MVK_C3_SPLIT256=0 timeout 600 python bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline 2>>$OUT/ab.err | line cfg5_tiled256
done
for i in 1 2 3; do
  timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>>$OUT/ab.err | line cfg4_split256
  MVK_TUNE=1 MVK_C3_SPLIT256=0 timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>>$OUT/ab.err | line cfg4_tiled256
done
