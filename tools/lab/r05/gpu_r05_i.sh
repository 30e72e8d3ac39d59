#!/bin/bash
# cfg4 / cfg5 with the modality branches on fewer streams (MVK_BRANCH_MAX)
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05i; mkdir -p $OUT
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2; do
  for bm in 0 2 3; do
    MVK_TUNE=1 MVK_BRANCH_MAX=$bm timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>>$OUT/ab.err | line cfg4_branchmax$bm
  done
done 2>&1 | tee $OUT/ab_branch.txt
