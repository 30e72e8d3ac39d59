#!/bin/bash
# Compile-time variants of csrc/dense16.hip: multivae_amd/libmvk_d16_<name>.so for each "name:flags" argument (built HERE).
set -e
cd "$(dirname "$0")/../multivae_amd/csrc"
make -s >/dev/null
OTHERS="igemm.o imgconv.o smallconv.o smallcin.o elbo.o mmvae.o misc.o utils.o skinny.o conv3small.o conv3rs.o comm.o"
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -c dense16.hip -o /tmp/dense16_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/dense16_$name.o $OTHERS -ldl -o ../libmvk_d16_$name.so ) &
done
wait
ls ../libmvk_d16_*.so
