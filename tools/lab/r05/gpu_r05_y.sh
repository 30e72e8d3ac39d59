#!/bin/bash
# where the MLP decoder's z-independent preparation (dense16 pack + target bound) runs: behind the short encoder (default) or in
# the decoder's own chain behind the posterior (MVK_EARLY_DENSE=0)
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05y; mkdir -p $OUT
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3 4; do
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line early_dense
  MVK_TUNE=1 MVK_EARLY_DENSE=0 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line late_dense
done
rocprofv3 --kernel-trace -d $OUT/tr -o t -- env MVK_TUNE=1 MVK_EARLY_DENSE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/tr.log 2>&1
python tools/step_timeline.py $(find $OUT/tr -name "*_results.db" | head -1) 0 14 > $OUT/timeline_late_dense.txt 2>&1
rm -rf $OUT/tr
head -30 $OUT/timeline_late_dense.txt
