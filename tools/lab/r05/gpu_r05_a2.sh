#!/bin/bash
# one alias of z per decoder (mvk_mopoe_posterior_bwd2): tests + headline A/B + timeline
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05a2; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "mopoe or fullsize or trainer or fused or crmvae or golden" 2>&1 | tail -4
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3 4; do
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line z_aliases
  MVK_TUNE=1 MVK_Z_ALIASES=0 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line autograd_add
done
rocprofv3 --kernel-trace -d $OUT/tr -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/tr.log 2>&1
python tools/step_timeline.py $(find $OUT/tr -name "*_results.db" | head -1) 0 14 > $OUT/timeline.txt 2>&1
rm -rf $OUT/tr
sed -n 28,60p $OUT/timeline.txt
