// Does a co-resident wave's MFMA stream delay another wave's global-load returns?  VGPR- vs AGPR-form accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int AGPR>
__global__ __launch_bounds__(256) void k(const float* src, float* out, unsigned long long* cyc, int iters, int nmfma) {
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7ffffff0, 0x00020000);
  float av = threadIdx.x, bv = 2.f;
  float s = 0.f;
  unsigned long long wait_c = 0, tot0 = __builtin_readcyclecounter();
  int base = (blockIdx.x * 256 + threadIdx.x) * 16;
  for (int it = 0; it < iters; ++it) {
    u32x4 v[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (base + ((it * 6 + u) * 1048576) % (64 << 20)), 0, 0);
    for (int m = 0; m < nmfma; m += 2) {
      if (AGPR) {
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc0) : "v"(av), "v"(bv));
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc1) : "v"(av), "v"(bv));
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc1, 0, 0, 0);
      }
    }
    unsigned long long t0 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wait_c += __builtin_readcyclecounter() - t0;
#pragma unroll
    for (int u = 0; u < 6; ++u) s += __uint_as_float(v[u].x);
  }
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) { atomicAdd(cyc, wait_c); atomicAdd(cyc + 1, __builtin_readcyclecounter() - tot0); atomicAdd(cyc + 2, 1ull); }
}
template <int AGPR>
void run(int bpc, int nmfma, float* src, float* out, unsigned long long* cyc) {
  hipMemset(cyc, 0, 24);
  int iters = 64;
  k<AGPR><<<256 * bpc, 256>>>(src, out, cyc, iters, nmfma);
  hipDeviceSynchronize();
  unsigned long long h[3]; hipMemcpy(h, cyc, 24, hipMemcpyDeviceToHost);
  printf("%s acc, %d blocks/CU, %2d MFMA/iter: vmcnt wait %6.0f cyc/iter, total %6.0f cyc/iter (ideal MFMA %d)\n", AGPR ? "AGPR" : "VGPR", bpc, nmfma,
         (double)h[0] / h[2] / iters, (double)h[1] / h[2] / iters, nmfma * 64);
}
int main() {
  float *src, *out; unsigned long long* cyc;
  hipMalloc(&src, 256 << 20); hipMalloc(&out, 256 * 4 * 256 * 4); hipMalloc(&cyc, 24);
  hipMemset(src, 0, 256 << 20);
  for (int bpc = 1; bpc <= 3; ++bpc) { run<0>(bpc, 32, src, out, cyc); run<1>(bpc, 32, src, out, cyc); }
  run<0>(3, 0, src, out, cyc);
  return 0;
}
