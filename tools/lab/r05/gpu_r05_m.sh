#!/bin/bash
# loss assembly placement: shipped (multi-workgroup, at the head of the backward pass) / one workgroup / behind the image layer's backward
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05m; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_trainer.py -m gpu -q -x -k "user_decoder" 2>&1 | tail -3
MVK_LOSS_AFTER_TAIL=1 timeout 900 python -m pytest tests -m gpu -q -x -k "mopoe_fullsize_golden or graph_replay_matches or fused_decoder_tail or mopoe_golden" 2>&1 | tail -3
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3 4; do
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>>$OUT/ab.err | line multi_wg_head
  MVK_TUNE=1 MVK_TERMS_WS=0 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>>$OUT/ab.err | line one_wg_head
  MVK_TUNE=1 MVK_LOSS_AFTER_TAIL=1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>>$OUT/ab.err | line multi_wg_after_tail
done
