#!/bin/bash
# A/B of the "1 MFMA, N others" pipeline width of imgconv_kernel (variant libraries built ON the GPU box)
for n in 3 4 5 6; do
(cd multivae_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DMVK_IC_SCHED=$n -c imgconv.hip -o /tmp/ic_v.o &&
 /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o /tmp/ic_v.o smallconv.o smallcin.o elbo.o mmvae.o misc.o utils.o skinny.o -o ../libmvk_v.so)
echo "IC_SCHED=$n"; MVK_LIB_PATH=$PWD/multivae_amd/libmvk_v.so python tools/imgconv_probe.py 5120 3 2>&1 | grep -E "^h=. (up  |down)" | cut -c1-58
done
