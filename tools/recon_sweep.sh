#!/bin/bash
# recon-NLL workgroup granularity sweep (run on the GPU box): samples per block
for c in 16 5 3 2; do
  MVK_RECON_CHUNK=$c python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('chunk', $c, 'ms/step', d['ms_per_step'], 'recon us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])"
done
