#!/bin/bash
# r06 h: rotation of the small (first-layer, split-K) leaves only
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r06h; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_trainer.py -x -q -k "rotat" 2>&1 | tail -5 | tee $OUT/pytest.txt
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3 4; do
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line unrotated | tee -a $OUT/ab.txt
  MVK_TUNE=1 MVK_ROT_SVHN=2 MVK_ROT_MLP=2 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --rotate 2>>$OUT/ab.err | line rot_small | tee -a $OUT/ab.txt

  MVK_TUNE=1 MVK_ROT_SVHN=0 MVK_ROT_MLP=2 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --rotate 2>>$OUT/ab.err | line rot_small_mlp | tee -a $OUT/ab.txt
done
grep -v amdgpu.ids $OUT/ab.err | tail -5
MVK_TUNE=1 MVK_ROT_SVHN=2 MVK_ROT_MLP=2 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --rotate > $OUT/bench_under_rocprof.log 2>&1
python tools/step_timeline.py $OUT/trace_results.db 0 14 > $OUT/step_timeline.txt 2>/dev/null
rm -f $OUT/trace_results.db
cat $OUT/step_timeline.txt | head -34; tail -22 $OUT/step_timeline.txt
