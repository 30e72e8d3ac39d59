#!/bin/bash
# r06 r: the partial flush of the deferred finishes on the late-leaf stream (directly behind the decoder's late weight gradients)
set -u
export TMPDIR=/tmp MVK_TUNE=1; OUT=gpurun_out/r06r; mkdir -p $OUT
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3 4; do
  for e in "-" "MVK_FLUSH_ON_LATE=1" "MVK_FLUSH_ON_LATE=1 MVK_LATE_DW0=0"; do
    envs=""; [ "$e" != "-" ] && envs="$e"
    env $envs timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line "[$e]" | tee -a $OUT/ab.txt
  done
done
grep -v "amdgpu.ids" $OUT/ab.err | grep -i "capture failed" | sort | uniq -c
MVK_FLUSH_ON_LATE=1 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
python tools/step_timeline.py $OUT/trace_results.db 0 14 > $OUT/step_timeline.txt 2>/dev/null
rm -f $OUT/trace_results.db
tail -30 $OUT/step_timeline.txt
