#!/bin/bash
# r06 l: switches whose balance may have moved now that nothing runs beside the image layer's backward launch
set -u
export TMPDIR=/tmp MVK_TUNE=1; OUT=gpurun_out/r06l; mkdir -p $OUT
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3; do
  for e in "-" "MVK_LATE_DENSE=1" "MVK_TAIL_FIRST=0" "MVK_LATE_DW0=0" "MVK_EARLY_DENSE=1" "MVK_WGRAD_PAIR=0" "MVK_FLUSH_SIBLING=0" "MVK_HEADS_BWD_MLP=0"; do
    envs=""; [ "$e" != "-" ] && envs="$e"
    env $envs timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line "[$e]" | tee -a $OUT/ab.txt
  done
done
