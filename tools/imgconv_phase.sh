#!/bin/bash
# ICPROF build of the library (cycle counters inside imgconv_kernel) + tools/imgconv_phase.py; run ON the GPU box from the repo root
(cd multivae_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DMVK_ICPROF $ICDEFS -c imgconv.hip -o /tmp/ic_prof.o &&
 /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o /tmp/ic_prof.o smallconv.o smallcin.o elbo.o mmvae.o misc.o utils.o skinny.o conv3small.o conv3rs.o dense16.o comm.o -ldl -o ../libmvk_icprof.so)
MVK_LIB_PATH=$PWD/multivae_amd/libmvk_icprof.so python tools/imgconv_phase.py
