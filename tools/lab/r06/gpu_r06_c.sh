#!/bin/bash
# r06 c: the rotated step — correctness first, then same-box A/B of the forms
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r06c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_trainer.py -x -q -k "rotated or graph_replay_matches or optimizer_inside" 2>&1 | tail -15 | tee $OUT/pytest.txt
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3; do
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-rotate 2>>$OUT/ab.err | line unrotated | tee -a $OUT/ab.txt
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line rotated_both | tee -a $OUT/ab.txt
  MVK_TUNE=1 MVK_ROT_MLP=0 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line rotated_svhn | tee -a $OUT/ab.txt
  MVK_TUNE=1 MVK_ROT_SVHN=0 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line rotated_mlp | tee -a $OUT/ab.txt
done
tail -5 $OUT/ab.err
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
python tools/step_timeline.py $OUT/trace_results.db 0 14 > $OUT/step_timeline.txt 2>/dev/null
rm -f $OUT/trace_results.db
tail -70 $OUT/step_timeline.txt
