#!/bin/bash
# Compile-time experiment variants of the GEMM engine: multivae_amd/libmvk_<name>.so for each "name:flags" argument,
# e.g. tools/build_variants.sh noa:-DMVK_X_NOA nob:-DMVK_X_NOB noab:"-DMVK_X_NOA -DMVK_X_NOB"
set -e
cd "$(dirname "$0")/../multivae_amd/csrc"
make -s >/dev/null
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $flags -c igemm.hip -o /tmp/igemm_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/igemm_$name.o smallconv.o smallcin.o elbo.o mmvae.o misc.o -o ../libmvk_$name.so ) &
done
wait
