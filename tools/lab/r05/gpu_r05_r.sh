#!/bin/bash
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05r; mkdir -p $OUT
for c in cfg4 cfg5; do
  rocprofv3 --kernel-trace --stats -d $OUT/trace_$c -o t -- python bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline > $OUT/trace_$c.log 2>&1
  db=$(find $OUT/trace_$c -name "*_results.db" | head -1)
  python tools/step_groups.py $db 5 70 > $OUT/${c}_step_groups.md 2>&1
  rm -rf $OUT/trace_$c
done
head -50 $OUT/cfg4_step_groups.md
