#!/bin/bash
# first convolution + pack launch in one grid: tests + headline / cfg2 A/B
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05x; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "pack or small_down or fullsize or svhn or mnistsvhn or trainer_with_hip_graph or cfg1 or mmvae" 2>&1 | tail -5
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3 4; do
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line fused
  MVK_TUNE=1 MVK_PACK_FUSED=0 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line separate
done
for i in 1 2; do
  timeout 600 python bench.py --config cfg2 --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line cfg2_fused
  MVK_TUNE=1 MVK_PACK_FUSED=0 timeout 600 python bench.py --config cfg2 --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line cfg2_separate
done
