#!/bin/bash
# SUPROF build of the library (cycle counters inside small_up_bwd_kernel) + tools/smallup_phase.py; run ON the GPU box from the repo root
(cd multivae_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMVK_SUPROF -c smallconv.hip -o /tmp/sc_prof.o &&
 /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o imgconv.o /tmp/sc_prof.o smallcin.o elbo.o mmvae.o misc.o utils.o skinny.o conv3small.o -o ../libmvk_suprof.so)
MVK_LIB_PATH=$PWD/multivae_amd/libmvk_suprof.so python tools/smallup_phase.py
