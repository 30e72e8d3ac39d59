#!/bin/bash
# Compile-time variants of small_up_fwd_h_kernel (csrc/smallconv.hip, MVK_SUH_* switches): multivae_amd/libmvk_suh_<name>.so for
# each "name:flags" argument (built HERE, hipcc cross-compiles; the .so files travel with gpurun).
#   tools/suh_variants.sh base: xf:-DMVK_SUH_XFIRST=1 ...     then on the GPU box: tools/suh_run.sh base xf ...
set -e
cd "$(dirname "$0")/../multivae_amd/csrc"
make -s >/dev/null
OTHERS="igemm.o imgconv.o smallcin.o elbo.o mmvae.o misc.o utils.o skinny.o conv3small.o conv3rs.o dense16.o comm.o"
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -c smallconv.hip -o /tmp/smallconv_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/smallconv_$name.o $OTHERS -ldl -o ../libmvk_suh_$name.so ) &
done
wait
ls -la ../libmvk_suh_*.so
