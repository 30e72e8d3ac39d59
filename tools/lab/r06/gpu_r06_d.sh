#!/bin/bash
# r06 d: the rotated step, remaining forms: the SVHN leaves alone with fewer workgroups (compute units left to the encoders' chain)
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r06d; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_trainer.py -x -q -k "rotated" 2>&1 | tail -5 | tee $OUT/pytest.txt
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3; do
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-rotate 2>>$OUT/ab.err | line unrotated | tee -a $OUT/ab.txt
  MVK_TUNE=1 MVK_ROT_MLP=0 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --rotate 2>>$OUT/ab.err | line rotated_svhn | tee -a $OUT/ab.txt
  MVK_TUNE=1 MVK_ROT_MLP=0 MVK_IMGWGRAD_GRID=192 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --rotate 2>>$OUT/ab.err | line rotated_svhn_grid192 | tee -a $OUT/ab.txt
  MVK_TUNE=1 MVK_ROT_MLP=0 MVK_IMGWGRAD_GRID=128 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --rotate 2>>$OUT/ab.err | line rotated_svhn_grid128 | tee -a $OUT/ab.txt
done
MVK_TUNE=1 MVK_ROT_MLP=0 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
python tools/step_timeline.py $OUT/trace_results.db 0 14 > $OUT/step_timeline_rotated_svhn.txt 2>/dev/null
rm -f $OUT/trace_results.db
head -40 $OUT/step_timeline_rotated_svhn.txt; tail -3 $OUT/step_timeline_rotated_svhn.txt
