#!/bin/bash
# r06 g: pair weight-gradient launch + unrolled first decoder layer, same-box A/B
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r06g; mkdir -p $OUT
python tools/smallk_probe.py 2>&1 | tail -3 | tee $OUT/smallk.txt
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
grep -v amdgpu.ids $OUT/ab.err | tail -5
