#!/bin/bash
# the image layer's backward on scaled fp16 pairs (small_up_bwd_h_kernel): tests + headline A/B
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05t; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x -k "small_up or fused or fullsize or svhn or mnistsvhn or trainer_with_hip_graph" 2>&1 | tail -5
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac'])"; }
for i in 1 2 3 4; do
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line tail_bwd_f16
  MVK_TUNE=1 MVK_TAIL_BWD_F16=0 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line tail_bwd_bf16
done
