#!/bin/bash
# smallcin_wgrad_kernel with a (2 C) x (Cv / 8) block per lane and the waves splitting the positions: tests + headline A/B against the
# previous object + the kernel's time inside the step
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05i2; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "conv_down_up_wgrad or svhn or fullsize or mnistsvhn or trainer_with_hip_graph or cfg2" 2>&1 | tail -4
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
OLD=$PWD/multivae_amd/libmvk_suh_sciold.so
for i in 1 2 3 4; do
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line new
  MVK_LIB_PATH=$OLD timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line old
done
rocprofv3 --kernel-trace -d $OUT/tr -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/tr.log 2>&1
python tools/step_timeline.py $(find $OUT/tr -name "*_results.db" | head -1) 0 14 2>/dev/null | tail -8
rm -rf $OUT/tr
