#!/bin/bash
# A/B of the scheduling pipeline of imgwgrad_kernel: builds variant libraries ON the GPU box and times the kernels
for n in 0 2 3 4 5 6; do
(cd multivae_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DMVK_IW_SCHED=$n -c imgconv.hip -o /tmp/ic_v.o &&
 /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o /tmp/ic_v.o smallconv.o smallcin.o elbo.o mmvae.o misc.o utils.o skinny.o -o ../libmvk_v.so)
echo "IW_SCHED=$n"; MVK_LIB_PATH=$PWD/multivae_amd/libmvk_v.so python tools/imgconv_probe.py 5120 3 2>&1 | grep wgrd | cut -c1-60
done
