#!/bin/bash
# the decoders' weight packs behind the short encoder (its stream has slack) instead of in the convolutional encoder's chain
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05c2; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "mopoe or fullsize or trainer or fused" 2>&1 | tail -4
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3 4; do
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line late_dec_pack
  MVK_TUNE=1 MVK_LATE_DEC_PACK=0 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line one_pack
done
rocprofv3 --kernel-trace -d $OUT/tr -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/tr.log 2>&1
python tools/step_timeline.py $(find $OUT/tr -name "*_results.db" | head -1) 0 14 > $OUT/timeline.txt 2>&1
rm -rf $OUT/tr
head -24 $OUT/timeline.txt
