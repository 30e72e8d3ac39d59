#!/bin/bash
# round 5, call D: fused-tail kernel variants; what the external event node alone costs the replayed step
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05d; mkdir -p $OUT
bash tools/suh_run.sh base xf pair wreg nt xfwr all3 all4 2>&1 | tee $OUT/suh_variants.txt
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['n_gpus'], d['config']['launch'][:70])"; }
D="MVK_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0"
for i in 1 2 3; do
  env $D MASTER_PORT=29611 MVK_OVERLAP=1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>>$OUT/ab.err | line dist_overlap
  env $D MASTER_PORT=29612 MVK_OVERLAP=2 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>>$OUT/ab.err | line dist_eventnode_serial
  env $D MASTER_PORT=29613 MVK_OVERLAP=0 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>>$OUT/ab.err | line dist_serial
done 2>&1 | tee $OUT/ab_overlap.txt
