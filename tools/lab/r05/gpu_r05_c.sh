#!/bin/bash
# round 5, call C: kernel trace of the headline step, A/B of the optimizer placement and of the collective overlap (one box)
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05c; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
python tools/rocpd_summary.py $OUT/trace_results.db > $OUT/kernel_stats.md 2>/dev/null
python tools/step_timeline.py $OUT/trace_results.db 0 14 > $OUT/step_timeline.txt 2>/dev/null
rm -f $OUT/trace_results.db
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['n_gpus'], d['config']['launch'][:70])"; }
for i in 1 2 3 4; do
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>>$OUT/ab.err | tee -a $OUT/ab_graph_adam.jsonl | line in_graph_prelude
  MVK_TUNE=1 MVK_ADAM_PRELUDE=0 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>>$OUT/ab.err | tee -a $OUT/ab_graph_adam_noprelude.jsonl | line in_graph_noprelude
  MVK_TUNE=1 MVK_GRAPH_ADAM=0 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>>$OUT/ab.err | tee -a $OUT/ab_host_adam.jsonl | line host_adam
done
for i in 1 2 3; do
  MVK_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>>$OUT/ab.err | tee -a $OUT/ab_dist_overlap.jsonl | line dist_overlap
  MVK_OVERLAP=0 MVK_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29612 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>>$OUT/ab.err | tee -a $OUT/ab_dist_serial.jsonl | line dist_serial
done
