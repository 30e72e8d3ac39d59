#!/bin/bash
# ON the GPU box: tools/dense16_probe.py for each variant library, three rounds interleaved
for r in 1 2 3; do
  for name in "$@"; do
    echo -n "$name  "; MVK_LIB_PATH=$PWD/multivae_amd/libmvk_d16_$name.so python tools/dense16_probe.py 2>/dev/null | grep -E "^fwd_nll|^bwd_data|^wgrad" | cut -c1-60 | tr '\n' '|'; echo
  done
done
