#!/bin/bash
# Which part of small_up_bwd_kernel bounds it (run ON the GPU box, from the repo root).  Builds an ablation library
# (-DMVK_ABLATE: results are wrong by construction) and times the kernel with parts switched off: MVK_ABLATE bits
# 1 = backward-data MFMA loop reduced to one step, 2 = weight-gradient loop reduced to one step, 4 = no dV stores,
# 8 = no prefetch loads after the first unit, 16 = no gradient-tile staging, 32 = no input-tile staging, 64 = no epilogue.
(cd multivae_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMVK_ABLATE -c smallconv.hip -o /tmp/sc_abl.o &&
 /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o imgconv.o /tmp/sc_abl.o smallcin.o elbo.o mmvae.o misc.o utils.o skinny.o -o ../libmvk_abl.so)
export MVK_LIB_PATH=$PWD/multivae_amd/libmvk_abl.so
for a in 0 1 2 3 4 7 8 15 31 47 79 127; do echo -n "abl=$a  "; MVK_ABLATE=$a python tools/smallup_probe.py 5120 7 | grep bwd | cut -c1-40; done
