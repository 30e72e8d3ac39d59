#!/bin/bash
# Quick GPU round trip: headline bench line + kernel trace of the same command (gpurun_out/<TAG>/).
TAG=${1:-quick}; OUT=gpurun_out/$TAG
export TMPDIR=/tmp; mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 2> $OUT/bench.err | tail -1 > $OUT/bench_cfg3.json; cat $OUT/bench_cfg3.json
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
python tools/rocpd_summary.py $OUT/trace_results.db > $OUT/kernel_stats.md 2>/dev/null
python tools/step_timeline.py $OUT/trace_results.db 0 14 > $OUT/step_timeline.txt 2>/dev/null
tail -3 $OUT/step_timeline.txt
rm -f $OUT/trace_results.db
