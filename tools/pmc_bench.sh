#!/bin/bash
# usage: tools/pmc_bench.sh TAG FILTER "COUNTERS"  — PMC pass over a short bench run, mean counters per kernel matching FILTER
TAG=$1; FILT=$2; shift; shift
/usr/local/graft/bin/gpurun --timeout 900 -- "mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp; rocprofv3 --pmc $* --kernel-trace -d gpurun_out/$TAG -o p -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/$TAG/bench.log 2>&1" 2>&1 | tail -1
python tools/pmc_agg.py gpurun_out/$TAG/p_results.db $FILT
