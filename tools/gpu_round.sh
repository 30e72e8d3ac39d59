#!/bin/bash
# One GPU round trip: parity tests, smoke, bench, rocprof kernel trace.  Writes everything under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== rocminfo"; rocminfo 2>/dev/null | grep -m2 -E "gfx950|Marketing" 
echo "=== pytest gpu"
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
echo "=== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "=== bench"
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -5 | tee gpurun_out/bench.log
