#!/bin/bash
# usage: tools/pmc.sh TAG "COUNTER1 COUNTER2 ..."  — one PMC pass (kernel-trace only) over a short bench run
TAG=$1; shift
/usr/local/graft/bin/gpurun --timeout 900 -- "mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp; rocprofv3 --pmc $* --kernel-trace -d gpurun_out/$TAG -o p -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/$TAG/bench.log 2>&1" 2>&1 | tail -1
