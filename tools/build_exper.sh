#!/bin/bash
# Experiment build of the library (-DMVK_EXPER: kernels honour mvk_debug_set_flags) -> multivae_amd/libmvk_exper.so
set -e
cd "$(dirname "$0")/../multivae_amd/csrc"
mkdir -p /tmp/mvk_exper
for f in igemm smallconv smallcin elbo mmvae misc; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMVK_EXPER -c $f.hip -o /tmp/mvk_exper/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/mvk_exper/*.o -o ../libmvk_exper.so
