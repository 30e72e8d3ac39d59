#!/bin/bash
# r06 w: the MLP decoder's preparation + forward launches behind the large decoder's first layer (MVK_FWD_DEFER=4)
set -u
export TMPDIR=/tmp MVK_TUNE=1; OUT=gpurun_out/r06w; mkdir -p $OUT
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3 4; do
  for e in "-" "MVK_FWD_DEFER=4"; do
    envs=""; [ "$e" != "-" ] && envs="$e"
    env $envs timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line "[$e]" | tee -a $OUT/ab.txt
  done
done
grep -v "amdgpu.ids" $OUT/ab.err | grep -i "capture failed" | sort | uniq -c
