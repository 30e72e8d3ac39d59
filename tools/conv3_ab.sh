#!/bin/bash
# On the GPU box: the 64 -> 64 probe for every variant library under build/v/
for f in build/v/libmvk_*.so; do
  echo "== $f"; MVK_LIB_PATH=$PWD/$f python tools/conv3_probe.py one 2>&1 | tail -2
done
