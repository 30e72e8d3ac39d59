#!/bin/bash
# round 5, call E: the whole GPU suite, smoke, the headline line (defaults after the A/B runs)
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05e; mkdir -p $OUT
timeout 2700 python -m pytest tests -m gpu -q --timeout=900 -rf 2>&1 > $OUT/pytest_gpu.log; tail -15 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 600 python bench.py --steps 30 --warmup 10 2> $OUT/bench.err | grep '^{' | tail -1 > $OUT/bench_cfg3.json; python -c "
import json;d=json.load(open('$OUT/bench_cfg3.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['avg_launch_us'],d['n_gpus'],d['config']['launch'])"
