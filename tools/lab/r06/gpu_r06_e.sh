#!/bin/bash
# r06 e: the whole GPU suite on the round's code + where the W = 1 cost of the data-parallel step goes
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r06e; mkdir -p $OUT
rm -f gpurun_out/iwae_float64.jsonl gpurun_out/flip_counts.jsonl
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest.txt
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'], d.get('settle_steps'))"; }
for i in 1 2 3; do
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line single | tee -a $OUT/ab.txt
  MVK_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line force_dist | tee -a $OUT/ab.txt
done
MVK_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29518 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT -o trace -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
python tools/step_timeline.py $OUT/trace_results.db 0 14 > $OUT/dp1_timeline.txt 2>/dev/null
rm -f $OUT/trace_results.db
head -8 $OUT/dp1_timeline.txt; tail -12 $OUT/dp1_timeline.txt
