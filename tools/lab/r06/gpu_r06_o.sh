#!/bin/bash
# r06 o: dense16 one-register-set forms on the new baseline (three workgroups per CU); heads_bwd auto on cfg2 / cfg3k1
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r06o; mkdir -p $OUT
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for i in 1 2 3; do
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line shipped | tee -a $OUT/ab.txt
  for v in fwd1 bwd1 both1; do
    MVK_LIB_PATH=$PWD/multivae_amd/libmvk_d16_$v.so timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>$OUT/ab.err | line d16_$v | tee -a $OUT/ab.txt
  done
done
for c in cfg2 cfg3k1; do for i in 1 2; do
  timeout 600 python bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline 2>>$OUT/ab.err | line "$c auto" | tee -a $OUT/ab.txt
done; done
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "mlp_encoder_decoder" 2>&1 | tail -2
