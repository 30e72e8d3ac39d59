#!/bin/bash
# usage: tools/pmc_probe.sh TAG LIB "COUNTERS"   — PMC pass over tools/bf_probe.py
TAG=$1; LIBP=$2; shift; shift
/usr/local/graft/bin/gpurun --timeout 900 -- "mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp; LIB=$LIBP rocprofv3 --pmc $* --kernel-trace -d gpurun_out/$TAG -o p -- python tools/bf_probe.py > gpurun_out/$TAG/probe.log 2>&1" 2>&1 | tail -1
python tools/pmc_agg.py gpurun_out/$TAG/p_results.db igemm_bf
