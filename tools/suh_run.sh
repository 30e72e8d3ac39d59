#!/bin/bash
# ON the GPU box: time the fused tail (tools/smallup_probe.py nll_s / fwd_s) for each variant library, three rounds interleaved
for r in 1 2 3; do
  for name in "$@"; do
    echo -n "$name  "; MVK_LIB_PATH=$PWD/multivae_amd/libmvk_suh_$name.so python tools/smallup_probe.py 5120 15 2>/dev/null | grep -E "^nll_s|^fwd_s" | cut -c1-62 | tr '\n' '|'; echo
  done
done
