#!/bin/bash
# r06 n: where the MLP decoder's forward launches start (MVK_FWD_DEFER = 1 / 2 / 3) on the new baseline; cfg2 / cfg3k1 with the MLP heads' backward unfused
set -u
export TMPDIR=/tmp MVK_TUNE=1; OUT=gpurun_out/r06n; mkdir -p $OUT
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3; do
  for e in "-" "MVK_FWD_DEFER=1" "MVK_FWD_DEFER=2" "MVK_FWD_DEFER=3" "MVK_HEADS_BWD_MLP=1"; do
    envs=""; [ "$e" != "-" ] && envs="$e"
    env $envs timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line "[$e]" | tee -a $OUT/ab.txt
  done
done
for c in cfg2 cfg3k1 cfg5; do for i in 1 2; do
  for e in "-" "MVK_HEADS_BWD_MLP=1"; do
    envs=""; [ "$e" != "-" ] && envs="$e"
    env $envs timeout 600 python bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline 2>>$OUT/ab.err | line "$c [$e]" | tee -a $OUT/ab.txt
  done
done; done
grep -v "amdgpu.ids" $OUT/ab.err | grep -i "capture failed" | sort | uniq -c
