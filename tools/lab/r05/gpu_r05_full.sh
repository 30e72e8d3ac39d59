#!/bin/bash
# the whole GPU suite + smoke on the current tree
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05full; mkdir -p $OUT
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $OUT/smoke.txt
