#!/bin/bash
# round 5, call B: external-event probe, the new tests, in-graph Adam and overlapped-collective A/B on one box
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05b; mkdir -p $OUT
echo "=== extevent probe"; MVK_SYNC_DEBUG=0 timeout 300 python tools/extevent_order_probe.py 2>&1 | tail -4 | tee $OUT/extevent.log
echo "=== new tests"
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -rf -x -k "optimizer_inside or two_ranks or integration_md or reduce_terms or gpus_2 or rccl_path or trainer_with_hip_graph or graph_replay or amsgrad_and_scheduler or resume_from" 2>&1 > $OUT/pytest_new.log; tail -12 $OUT/pytest_new.log
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['config']['launch'][:60], d['roofline'].get('measured_copy_GBs'), d['roofline']['elbo_group']['us_per_step'])"; }
for i in 1 2 3; do
  python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>>$OUT/ab.err | tail -1 | tee -a $OUT/ab_graph_adam.jsonl | line in_graph_adam
  MVK_TUNE=1 MVK_GRAPH_ADAM=0 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>>$OUT/ab.err | tail -1 | tee -a $OUT/ab_host_adam.jsonl | line host_adam
done
for i in 1 2; do
  MVK_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>>$OUT/ab.err | tail -1 | tee -a $OUT/ab_dist_overlap.jsonl | line dist_overlap
  MVK_OVERLAP=0 MVK_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29612 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>>$OUT/ab.err | tail -1 | tee -a $OUT/ab_dist_serial.jsonl | line dist_serial
done
tail -5 $OUT/ab.err
