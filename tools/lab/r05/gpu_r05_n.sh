#!/bin/bash
# image-side 3x3 launches publish max |Y|: kernel test, ResNet goldens, cfg5 / cfg4 A/B
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05n; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x -k "image_side_publishes or conv3x3_fwd_bwd or resnet or jmvae_cub or cfg4_cfg5" 2>&1 | tail -4
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3; do
  timeout 600 python bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline 2>>$OUT/ab.err | line cfg5_publish
  MVK_TUNE=1 MVK_C3_Y_AMAX=0 timeout 600 python bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline 2>>$OUT/ab.err | line cfg5_pass
done
for i in 1 2; do
  timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>>$OUT/ab.err | line cfg4_publish
  MVK_TUNE=1 MVK_C3_Y_AMAX=0 timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>>$OUT/ab.err | line cfg4_pass
done
