#!/bin/bash
# mvk_conv3x3_s2 (post-activation ResNet blocks: conv2 + LeakyReLU + sum + second store in one launch): tests + cfg4 A/B
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05w; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "conv3x3 or resnet or mmvaeplus or cfg4 or polymnist" 2>&1 | tail -5
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3; do
  timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>>$OUT/ab.err | line cfg4_dual
  MVK_TUNE=1 MVK_C3_DUAL=0 timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>>$OUT/ab.err | line cfg4_axpby
done
