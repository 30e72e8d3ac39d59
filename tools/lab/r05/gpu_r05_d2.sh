#!/bin/bash
# one-launch backward for two heads of up to 64 outputs (heads_bwd_kernel<128>): tests + cfg5 A/B
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05d2; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "heads or mlp_encoder or jmvae or cfg5 or cub or fullsize" 2>&1 | tail -5
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3; do
  timeout 600 python bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline 2>>$OUT/ab.err | line cfg5_heads64
  MVK_TUNE=1 MVK_HEADS_BWD_MAXN=32 timeout 600 python bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline 2>>$OUT/ab.err | line cfg5_heads32
done
