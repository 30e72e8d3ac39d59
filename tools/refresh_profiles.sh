#!/bin/bash
# Re-measure and refresh profiles/ (run from the repo root in the build container): rocprofv3 kernel trace of the
# default bench command on the GPU box -> per-kernel summary, last-step timeline, bench log.
set -e
TAG=${1:-r01}
/usr/local/graft/bin/gpurun --timeout 1200 -- "mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG -o trace -- python bench.py --steps 20 --warmup 5 > gpurun_out/$TAG/bench_under_rocprof.log 2>&1; tail -1 gpurun_out/$TAG/bench_under_rocprof.log | cut -c1-300" 2>&1 | tail -3
python tools/rocpd_summary.py gpurun_out/$TAG/trace_results.db > profiles/${TAG}_kernel_stats.md
python tools/step_timeline.py gpurun_out/$TAG/trace_results.db 0 > profiles/${TAG}_step_timeline.txt
grep -v "amdgpu.ids" gpurun_out/$TAG/bench_under_rocprof.log | tail -3 > profiles/${TAG}_bench_under_rocprof.log
head -12 profiles/${TAG}_kernel_stats.md; tail -2 profiles/${TAG}_step_timeline.txt
