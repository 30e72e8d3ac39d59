#!/bin/bash
# heads_bwd_kernel with LDS sized by the heads' width (48 KB at 2 x 20 latents instead of 62): tests + headline A/B + timeline
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05p2; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "heads or mlp_encoder or svhn or fullsize or trainer_with_hip_graph" 2>&1 | tail -4
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
OLD=$PWD/multivae_amd/libmvk_suh_skold.so
for i in 1 2 3 4; do
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line new
  MVK_LIB_PATH=$OLD timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line old
done
rocprofv3 --kernel-trace -d $OUT/tr -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/tr.log 2>&1
python tools/step_timeline.py $(find $OUT/tr -name "*_results.db" | head -1) 0 14 2>/dev/null | sed -n 33,48p
rm -rf $OUT/tr
