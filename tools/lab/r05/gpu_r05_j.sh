#!/bin/bash
# the three-workgroups-per-CU fused tail (MVK_SUH3=1): correctness tests, alone timings, in-step A/B
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05j; mkdir -p $OUT
MVK_SUH3=1 timeout 900 python -m pytest tests -m gpu -q -x -k "small_up_fwd_scaled_fp16 or fused_decoder_tail or mopoe_fullsize_golden" 2>&1 | tail -4
for r in 1 2 3; do
  echo -n "h2 "; python tools/smallup_probe.py 5120 15 2>/dev/null | grep -E "^nll_s|^fwd_s" | cut -c1-62 | tr '\n' '|'; echo
  echo -n "h3 "; MVK_TUNE=1 MVK_SUH3=1 python tools/smallup_probe.py 5120 15 2>/dev/null | grep -E "^nll_s|^fwd_s" | cut -c1-62 | tr '\n' '|'; echo
done
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"; }
for i in 1 2 3; do
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>>$OUT/ab.err | line h2
  MVK_TUNE=1 MVK_SUH3=1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>>$OUT/ab.err | line h3
done
