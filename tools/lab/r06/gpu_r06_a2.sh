#!/bin/bash
# r06 a2: the decoders' gradient into the latent as ONE narrow launch (narrow_fwd_kernel) instead of split-K + reduce
set -u
export TMPDIR=/tmp MVK_TUNE=1; OUT=gpurun_out/r06a2; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -k "svhn or mlp_encoder_decoder or fullsize or mopoe_golden or fused_decoder or heads" 2>&1 | grep -E "passed|failed|Error" | tail -3
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3 4; do
  for e in "MVK_NARROW_DZ=0" "-"; do
    envs=""; [ "$e" != "-" ] && envs="$e"
    env $envs timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line "[$e]" | tee -a $OUT/ab.txt
  done
done
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
python tools/step_timeline.py $OUT/trace_results.db 0 14 > $OUT/step_timeline.txt 2>/dev/null
rm -f $OUT/trace_results.db
grep -n "narrow_fwd\|posterior_bwd\|imgwgrad_kernel<8" $OUT/step_timeline.txt
