#!/bin/bash
# split-K block-target comparison (run on the GPU box): alternate two settings to separate them from box noise
for r in 1 2 3 4; do for a in 1024 768; do
  MVK_SPLITK_TARGET_1024=$a python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('target', $a, 'ms/step', d['ms_per_step'])"
done; done
