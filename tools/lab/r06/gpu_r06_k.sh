#!/bin/bash
# r06 k: the loss assembly postponed to the end of the step (no reader inside it: the posterior takes the KL rows' constant gradient)
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r06k; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q -k "mopoe or golden or trainer or graph or replay or determin or reproduc or rotat or fused_decoder or user_decoder" 2>&1 | tail -6 | tee $OUT/pytest.txt
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
r=d.get('roofline',{})
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'], 'elbo_group_us', (r.get('elbo_group') or {}).get('us_per_step'))"; }
for i in 1 2 3 4; do
  MVK_TUNE=1 MVK_ASSEMBLY_LAST=0 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line head_of_backward | tee -a $OUT/ab.txt
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line last | tee -a $OUT/ab.txt
done
grep -v amdgpu.ids $OUT/ab.err | grep -v "^  File\|^    " | tail -3
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
python tools/step_timeline.py $OUT/trace_results.db 0 14 > $OUT/step_timeline.txt 2>/dev/null
rm -f $OUT/trace_results.db
sed -n 22,32p $OUT/step_timeline.txt; tail -16 $OUT/step_timeline.txt
