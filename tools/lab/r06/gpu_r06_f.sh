#!/bin/bash
# r06 f: the convolutional encoder's two inner weight gradients as one launch (mvk_conv4s2_wgrad_pair)
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r06f; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q -k "svhn or golden or determin or reproduc or replay or rotated or graph" 2>&1 | tail -6 | tee $OUT/pytest.txt
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3 4; do
  MVK_TUNE=1 MVK_WGRAD_PAIR=0 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line two_launches | tee -a $OUT/ab.txt
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line pair | tee -a $OUT/ab.txt
done
for c in cfg2 cfg3k1; do for i in 1 2; do
  MVK_TUNE=1 MVK_WGRAD_PAIR=0 timeout 600 python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line ${c}_two_launches | tee -a $OUT/ab.txt
  timeout 600 python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line ${c}_pair | tee -a $OUT/ab.txt
done; done
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
python tools/step_timeline.py $OUT/trace_results.db 0 14 > $OUT/step_timeline.txt 2>/dev/null
rm -f $OUT/trace_results.db
tail -28 $OUT/step_timeline.txt
