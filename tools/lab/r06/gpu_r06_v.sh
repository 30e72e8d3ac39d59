#!/bin/bash
# r06 v: runtime knobs of the HIP graph executor on the launch-bound configurations (cfg2, cfg3k1) and the headline
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r06v; mkdir -p $OUT
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for c in cfg2 cfg3k1 cfg3; do
  for i in 1 2; do
    for e in "-" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=2" "DEBUG_HIP_FORCE_GRAPH_QUEUES=8" "DEBUG_CLR_MAX_BATCH_SIZE=1000" "DEBUG_CLR_MAX_BATCH_SIZE=8" "GPU_MAX_HW_QUEUES=8" "DEBUG_HIP_DYNAMIC_QUEUES=1" "HIP_FORCE_DEV_KERNARG=1" "DEBUG_HIP_KERNARG_COPY_OPT=1"; do
      envs=""; [ "$e" != "-" ] && envs="$e"
      env $envs timeout 600 python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line "$c [$e]" | tee -a $OUT/ab.txt
    done
  done
done
