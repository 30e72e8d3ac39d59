// micro-benchmark: sustained rate of v_mfma_f32_32x32x16_bf16 with NO memory traffic (operands and accumulators in registers),
// one wave per SIMD as in the register-stationary kernels, for launches of 50 us ... 1 ms: what the matrix pipes deliver once the
// chip has settled at its power-limited clock.  The split-bf16 product costs 6 of these per fp32 MAC: the fp32-equivalent
// ceiling is this number / 6.   hipcc --offload-arch=gfx950 -O3 mfma_bf16_sustained.hip -o mfma_bf16_sustained.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int RND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(float* out, unsigned long long* cyc, int iters, unsigned seed) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // RND = 0: constant operands (few toggling bits, the best case for power); RND = 1: 8 operand pairs of hashed values in
  // [-2, 2) per lane, a different pair for every MFMA (what real data does to the multiplier arrays)
  bf16x8 av[8], bv[8];
  for (int j = 0; j < 8; ++j) {
    u32x4 ua, ub;
    for (int e = 0; e < 4; ++e) {
      unsigned h = (seed + threadIdx.x * 2654435761u + j * 40503u + e * 9176u + blockIdx.x * 7919u) * 2246822519u;
      h ^= h >> 15; h *= 3266489917u; h ^= h >> 13;
      const unsigned lo = 0x3f80u | (h & 0x807fu), hi = 0x3f80u | ((h >> 16) & 0x807fu);  // +-[1, 2)
      ua[e] = RND ? (lo | (hi << 16)) : 0x3f803f80u;
      ub[e] = RND ? ((hi ^ 0x8000u) | (lo << 16)) : 0x3f003f00u;
    }
    av[j] = __builtin_bit_cast(bf16x8, ua);
    bv[j] = __builtin_bit_cast(bf16x8, ub);
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[(u + i) & 7], bv[(u * 3 + i) & 7], acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC, int RND>
void run(int iters, float* out, unsigned long long* cyc) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC, RND><<<256, 256>>>(out, cyc, 10, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) k<NACC, RND><<<256, 256>>>(out, cyc, iters, 7u + r);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n_mfma = 256.0 * 4 * iters * 16 * NACC, flops = n_mfma * 32768.0 * 20;
  const double us = ms * 1e3 / 20;
  printf("%s NACC=%d iters=%5d : %7.1f us/launch  %7.1f TFLOP/s bf16 = %6.1f fp32-equivalent (/6)  shader clock %.2f GHz  pipe busy %.0f %%\n", RND ? "random  " : "constant", NACC, iters, us,
         flops / ms / 1e9, flops / ms / 1e9 / 6, c / us / 1e3, 100.0 * iters * 16 * NACC * 32 / (double)c);
}
int main() {
  float* out; unsigned long long* cyc; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  for (int rep = 0; rep < 2; ++rep)
    for (int iters : {100, 300, 1000}) { run<4, 0>(iters, out, cyc); run<4, 1>(iters, out, cyc); }
  return 0;
}
