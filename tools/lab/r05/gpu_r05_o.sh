#!/bin/bash
# split-K heads forward: tests + cfg4 / cfg5 A/B
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05o; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x -k "linear_fwd or heads_fwd or resnet or jmvae_cub or cfg4_cfg5 or mmvaeplus_resnet" 2>&1 | tail -4
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3; do
  timeout 600 python bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline 2>>$OUT/ab.err | line cfg5_ksplit
  MVK_TUNE=1 MVK_HEADS_KSPLIT=0 timeout 600 python bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline 2>>$OUT/ab.err | line cfg5_nosplit
done
for i in 1 2; do
  timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>>$OUT/ab.err | line cfg4_ksplit
  MVK_TUNE=1 MVK_HEADS_KSPLIT=0 timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>>$OUT/ab.err | line cfg4_nosplit
done
