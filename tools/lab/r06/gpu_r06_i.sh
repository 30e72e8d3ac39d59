#!/bin/bash
# r06 i: the rotated step's head branch forked BEHIND the first launch of the main chain (queue assignment of the replayed graph)
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r06i; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_trainer.py -x -q -k "rotat" 2>&1 | tail -3 | tee $OUT/pytest.txt
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3; do
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line unrotated | tee -a $OUT/ab.txt
  MVK_TUNE=1 MVK_ROT_SVHN=2 MVK_ROT_MLP=2 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --rotate 2>>$OUT/ab.err | line rot_small_armed | tee -a $OUT/ab.txt
  MVK_TUNE=1 MVK_ROT_SVHN=2 MVK_ROT_MLP=0 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --rotate 2>>$OUT/ab.err | line rot_small_svhn_armed | tee -a $OUT/ab.txt
  MVK_TUNE=1 MVK_ROT_SVHN=1 MVK_ROT_MLP=0 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --rotate 2>>$OUT/ab.err | line rot_svhn_armed | tee -a $OUT/ab.txt
  MVK_TUNE=1 MVK_ROT_SVHN=1 MVK_ROT_MLP=1 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --rotate 2>>$OUT/ab.err | line rot_both_armed | tee -a $OUT/ab.txt
  MVK_TUNE=1 MVK_ROT_SVHN=2 MVK_ROT_MLP=2 MVK_ROT_ARM=0 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --rotate 2>>$OUT/ab.err | line rot_small_front | tee -a $OUT/ab.txt
done
grep -v amdgpu.ids $OUT/ab.err | grep -v "^  File\|^    " | tail -5
MVK_TUNE=1 MVK_ROT_SVHN=2 MVK_ROT_MLP=2 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --rotate > $OUT/bench_under_rocprof.log 2>&1
python tools/step_timeline.py $OUT/trace_results.db 0 14 > $OUT/step_timeline.txt 2>/dev/null
rm -f $OUT/trace_results.db
head -24 $OUT/step_timeline.txt; tail -3 $OUT/step_timeline.txt
