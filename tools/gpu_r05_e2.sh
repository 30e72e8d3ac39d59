#!/bin/bash
# clock pre-warm of bench.py: the driver-shaped command with and without
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05e2; mkdir -p $OUT
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['prewarm_steps'], d['ms_per_step_min_max'])"; }
for i in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>>$OUT/ab.err | line prewarm100
  python bench.py --gpus 1 --steps 20 --warmup 5 --prewarm-ms 0 --no-cpu-baseline 2>>$OUT/ab.err | line prewarm0
  python bench.py --gpus 1 --steps 20 --warmup 5 --prewarm-ms 400 --no-cpu-baseline 2>>$OUT/ab.err | line prewarm400
done
python bench.py 2>>$OUT/ab.err | grep '^{' | tail -1 | cut -c1-400
timeout 900 python -m pytest tests -m gpu -q -x -k "bench" 2>&1 | tail -3
