#!/bin/bash
# sweep of the LDS row padding of imgwgrad_kernel<4, 64, 128> (variant libraries built ON the GPU box; rocprofv3 kernel durations)
export TMPDIR=/tmp
for a in 16 32 40 48 56 72; do
(cd multivae_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DMVK_IW_PAD_A=$a -c imgconv.hip -o /tmp/ic_v.o 2>/dev/null &&
 /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o /tmp/ic_v.o smallconv.o smallcin.o elbo.o mmvae.o misc.o utils.o skinny.o -o ../libmvk_v.so) || { echo "PAD $a: build failed"; continue; }
rm -rf /tmp/pp; MVK_LIB_PATH=$PWD/multivae_amd/libmvk_v.so rocprofv3 --kernel-trace --stats -d /tmp/pp -o t -- python tools/imgconv_probe.py prof new > /dev/null 2>&1
echo "PAD_A=$a"; python tools/rocpd_summary.py /tmp/pp/t_results.db | grep "imgwgrad" | cut -c1-100
done
