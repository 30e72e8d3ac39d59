#!/bin/bash
# Variant builds of conv3rs.hip (one instantiation: 64 -> 64 channels) for A/B probes on the GPU box:
#   bash tools/conv3_variants.sh NAME "-DMVK_C3_SCHED=5 ..." [NAME2 "flags2" ...]   -> build/v/libmvk_NAME.so
# build/ is git-ignored but travels with gpurun; tools/conv3_probe.py picks a library through MVK_LIB_PATH.
set -e
cd "$(dirname "$0")/../multivae_amd/csrc"
mkdir -p ../../build/v
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -fno-slp-vectorize -DMVK_C3_PROBE_ONLY $flags \
    -c conv3rs.hip -o ../../build/v/conv3rs_$name.o
  objs=$(ls *.o | grep -v '^conv3rs.o$' | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../../build/v/conv3rs_$name.o -o ../../build/v/libmvk_$name.so
  rm -f ../../build/v/conv3rs_$name.o
  echo built $name
done
