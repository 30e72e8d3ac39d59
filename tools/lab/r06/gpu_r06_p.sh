#!/bin/bash
# r06 p: the convolutional encoder's backward-data launches on the tiled engine while the decoder's late leaves hold the CUs
set -u
export TMPDIR=/tmp MVK_TUNE=1; OUT=gpurun_out/r06p; mkdir -p $OUT
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3; do
  for e in "-" "MVK_ENC_BWD_TILED=1" "MVK_ENC_BWD_TILED=2" "MVK_ENC_BWD_TILED=3" "MVK_ENC_BWD_TILED=1 MVK_WGRAD_PAIR=0"; do
    envs=""; [ "$e" != "-" ] && envs="$e"
    env $envs timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line "[$e]" | tee -a $OUT/ab.txt
  done
done
grep -v "amdgpu.ids" $OUT/ab.err | grep -i "capture failed" | sort | uniq -c
