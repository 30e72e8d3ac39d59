#!/bin/bash
# split-K for small-M 3x3 convolutions: tests + cfg4 A/B (+ cfg5 sanity)
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05s; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "conv3x3 or resnet or jmvae_cub or cfg4_cfg5 or mmvaeplus_resnet or cfg4_full or cfg5_full" 2>&1 | tail -4
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3; do
  timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>>$OUT/ab.err | line cfg4_c3splitk
  MVK_TUNE=1 MVK_C3_SPLITK=0 timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>>$OUT/ab.err | line cfg4_nosplit
done
for i in 1 2; do
  timeout 600 python bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline 2>>$OUT/ab.err | line cfg5_c3splitk
  MVK_TUNE=1 MVK_C3_SPLITK=0 timeout 600 python bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline 2>>$OUT/ab.err | line cfg5_nosplit
done
