#!/bin/bash
# r06 t: one replayed step of MoPoE MnistSvhn at K = 1 (cfg3k1) and of MMVAE MnistSvhn (cfg2): where the launch-bound configurations spend their step
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r06t; mkdir -p $OUT
for c in cfg3k1 cfg2; do
  rocprofv3 --kernel-trace --stats -d $OUT/$c -o trace -- python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$c.log 2>&1
  python tools/step_timeline.py $OUT/$c/trace_results.db 0 14 > $OUT/step_timeline_$c.txt 2>/dev/null
  rm -rf $OUT/$c
done
cat $OUT/step_timeline_cfg3k1.txt
