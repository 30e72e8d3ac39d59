#!/bin/bash
# One GPU round trip: parity tests, smoke, bench, rocprof kernel trace.  Writes everything under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== rocminfo"; rocminfo 2>/dev/null | grep -m2 -E "gfx950|Marketing" 
echo "=== fullsize probe"
timeout 600 python tools/fullsize_probe.py > gpurun_out/fullsize_probe.log 2>&1; tail -30 gpurun_out/fullsize_probe.log
echo "=== pytest gpu"
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 -rf -s 2>&1 > gpurun_out/pytest_gpu.log; tail -60 gpurun_out/pytest_gpu.log
echo "=== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "=== bench"
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -5 | tee gpurun_out/bench.log
