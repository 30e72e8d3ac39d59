#!/bin/bash
# r06 b: the SVHN encoder's weight gradients on a leaf stream (MVK_LEAF_STREAM, a round-2 loss) re-measured, with and without the
# decoder's late leaves in the tail (MVK_SKIP_LATE = what a rotated step's tail would look like)
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r06b; mkdir -p $OUT
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3; do
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line shipped | tee -a $OUT/ab.txt
  MVK_TUNE=1 MVK_LEAF_STREAM=1 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line leaf_stream | tee -a $OUT/ab.txt
  MVK_TUNE=1 MVK_SKIP_LATE=1 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line skip_late | tee -a $OUT/ab.txt
  MVK_TUNE=1 MVK_SKIP_LATE=1 MVK_LEAF_STREAM=1 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line skip_late+leaf_stream | tee -a $OUT/ab.txt
done
MVK_TUNE=1 MVK_SKIP_LATE=1 MVK_LEAF_STREAM=1 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
python tools/step_timeline.py $OUT/trace_results.db 0 14 > $OUT/step_timeline.txt 2>/dev/null
rm -f $OUT/trace_results.db
tail -32 $OUT/step_timeline.txt
