#!/bin/bash
# usage: tools/prof.sh TAG [min_us]  — rocprofv3 kernel trace of bench.py on the GPU box + last-step timeline
TAG=${1:-p}; MIN=${2:-25}
/usr/local/graft/bin/gpurun --timeout 900 -- "mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$TAG/bench.log 2>&1" 2>&1 | tail -1
python tools/step_timeline.py gpurun_out/$TAG/t_results.db $MIN
