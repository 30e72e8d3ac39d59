#!/bin/bash
# round 5, call A: the suite on the new boundary code, the external-event probe, the headline line
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05a; mkdir -p $OUT
echo "=== extevent probe"; timeout 300 python tools/extevent_order_probe.py 2>&1 | tail -5 | tee $OUT/extevent.log
echo "=== pytest gpu"
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 -rf -x 2>&1 > $OUT/pytest_gpu.log; tail -15 $OUT/pytest_gpu.log
echo "=== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
echo "=== bench"; timeout 600 python bench.py --steps 30 --warmup 10 2> $OUT/bench.err | tail -1 > $OUT/bench_cfg3.json; python -c "
import json;d=json.load(open('$OUT/bench_cfg3.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['n_gpus'])"
echo "=== bench --gpus 2 same gpu gloo (self launch)"; MVK_DIST_BACKEND=gloo MVK_BENCH_SAME_GPU=1 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --batch 64 2> $OUT/bench2.err | tail -1 | cut -c1-300
