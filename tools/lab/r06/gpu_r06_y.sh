#!/bin/bash
# r06 y: the long encoder's backward enqueued before the short one's (node creation order of the encoders)
set -u
export TMPDIR=/tmp MVK_TUNE=1; OUT=gpurun_out/r06y; mkdir -p $OUT
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3 4; do
  for e in "-" "MVK_ENC_SIDE_FIRST=1" "MVK_ENC_SIDE_FIRST=1 MVK_HEADS_BWD_MLP=1"; do
    envs=""; [ "$e" != "-" ] && envs="$e"
    env $envs timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line "[$e]" | tee -a $OUT/ab.txt
  done
done
grep -v "amdgpu.ids" $OUT/ab.err | grep -i "capture failed" | sort | uniq -c
MVK_ENC_SIDE_FIRST=1 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
python tools/step_timeline.py $OUT/trace_results.db 0 14 > $OUT/step_timeline.txt 2>/dev/null
rm -f $OUT/trace_results.db
tail -28 $OUT/step_timeline.txt
