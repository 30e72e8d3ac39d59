// micro-benchmark: v_mfma_f32_32x32x2_f32 issue rate vs number of independent accumulators and waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float av = a + threadIdx.x, bv = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu, float* out) {
  int iters = 2000;
  int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<grid, 256>>>(out, 10, 1.f, 2.f);
  hipEventRecord(e0);
  k<NACC><<<grid, 256>>>(out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)grid * 4 * iters * 8 * NACC * 4096.0;
  printf("NACC=%d waves/SIMD=%d : %.1f TFLOP/s (%.3f ms)\n", NACC, blocks_per_cu, flops / ms / 1e9, ms);
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  for (int w = 1; w <= 4; ++w) { run<1>(w, out); run<2>(w, out); run<4>(w, out); }
  return 0;
}
