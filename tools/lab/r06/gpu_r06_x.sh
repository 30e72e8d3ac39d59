#!/bin/bash
# r06 x: the batch copy as one launch (mvk_copy_batch) — test + same-box pairs (the previous library = two blits)
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r06x; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_trainer.py -x -q -k "copy_batch or graph or replay or trainer_with" 2>&1 | tail -3
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3 4; do
  MVK_TWO_BLITS=1 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line two_blits | tee -a $OUT/ab.txt
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line copy_batch | tee -a $OUT/ab.txt
done
for c in cfg2 cfg3k1; do for i in 1 2; do
  MVK_TWO_BLITS=1 timeout 600 python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line "$c two_blits" | tee -a $OUT/ab.txt
  timeout 600 python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line "$c copy_batch" | tee -a $OUT/ab.txt
done; done
