#!/bin/bash
# persistent amax arena: test + in-step A/B + the whole suite
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05k; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_trainer.py -m gpu -q -x -k "persistent_amax or graph_replay or full_size_step" 2>&1 | tail -4
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3 4; do
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>>$OUT/ab.err | line persist
  MVK_TUNE=1 MVK_ARENA_PERSIST=0 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>>$OUT/ab.err | line private
done
timeout 2700 python -m pytest tests -m gpu -q --timeout=900 -rf 2>&1 > $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log
