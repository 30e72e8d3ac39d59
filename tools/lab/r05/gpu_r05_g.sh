#!/bin/bash
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05g; mkdir -p $OUT
for name in base os os4f; do
  echo "== $name"; D16_ABLATE=1 MVK_LIB_PATH=$PWD/multivae_amd/libmvk_d16_$name.so python tools/dense16_probe.py 2>&1 | grep -E "^fwd_nll|^bwd_data|^wgrad|cycle stamps|^first|^pack"
done 2>&1 | tee $OUT/d16_ablate.txt
