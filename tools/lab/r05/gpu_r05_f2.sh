#!/bin/bash
# conv3small.hip variants against the previous object: tests + cfg4 / cfg5 A/B + per-kernel times
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05f2; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "conv3x3 or resnet or jmvae_cub or mmvaeplus or polymnist" 2>&1 | tail -4
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
OLD=$PWD/multivae_amd/libmvk_suh_c3sold.so
for i in 1 2 3; do
  timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>>$OUT/ab.err | line cfg4_new
  MVK_LIB_PATH=$OLD timeout 600 python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline 2>>$OUT/ab.err | line cfg4_old
done
for i in 1 2; do
  timeout 600 python bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline 2>>$OUT/ab.err | line cfg5_new
  MVK_LIB_PATH=$OLD timeout 600 python bench.py --config cfg5 --steps 20 --warmup 5 --no-cpu-baseline 2>>$OUT/ab.err | line cfg5_old
done
rocprofv3 --kernel-trace --stats -d $OUT/tr -o t -- python bench.py --config cfg4 --steps 6 --warmup 3 --no-cpu-baseline > $OUT/tr.log 2>&1
python tools/step_groups.py $(find $OUT/tr -name "*_results.db" | head -1) 4 70 2>/dev/null | grep -E "smallcout|small_wgrad|smallcin|dispatches"
rm -rf $OUT/tr
