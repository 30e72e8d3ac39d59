#!/bin/bash
# r06 z: fewer streams for the launch-bound configurations (cfg2, cfg3k1)?
set -u
export TMPDIR=/tmp MVK_TUNE=1; OUT=gpurun_out/r06z; mkdir -p $OUT
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for c in cfg2 cfg3k1; do
  for i in 1 2; do
    for e in "-" "MVK_BRANCH_STREAMS=0" "MVK_LATE_LEAVES=0" "MVK_ASYNC_LOSS=0" "MVK_FLUSH_SIBLING=0" "MVK_BRANCH_STREAMS=0 MVK_LATE_LEAVES=0 MVK_FLUSH_SIBLING=0" "MVK_LATE_DW0=0"; do
      envs=""; [ "$e" != "-" ] && envs="$e"
      env $envs timeout 600 python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line "$c [$e]" | tee -a $OUT/ab.txt
    done
  done
done
