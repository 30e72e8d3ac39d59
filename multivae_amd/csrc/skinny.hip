// Narrow-output linear layers in ONE launch: the (embedding, log-covariance) heads of the encoders
// (reference: models/nn/default_architectures.py Encoder_VAE_MLP.embedding / .log_var, models/nn/svhn.py:29-30 the two
// Conv2d(128, L, 4, 2, 0) heads of Encoder_VAE_SVHN).  N <= 32 outputs per head from K = 400 ... 2048 inputs over a batch
// of a few hundred rows is 0.04-0.09 GFLOP: on the tiled GEMM engine each head was a split-K launch plus its reduce
// (4 launches of 5-10 us in a launch-latency-bound part of the step).  Here a workgroup owns a [16 rows] x [16 columns]
// tile of one head, its 4 waves split K, v_mfma_f32_16x16x4_f32 (exact fp32) accumulates, and the 4 partial tiles are
// added in a fixed order through LDS (deterministic).
#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct HeadsArgs {
  const float* X;
  const float* W[2];
  const float* b[2];
  float* Y[2];
  int M, N, K, tiles_per_head;
  long long w_sk, w_sn;  // W(k, n) = W[k * w_sk + n * w_sn]
};

__global__ __launch_bounds__(256) void heads_fwd_kernel(const HeadsArgs g) {
  __shared__ float red[4][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  const int head = blockIdx.y / g.tiles_per_head, n0 = (blockIdx.y % g.tiles_per_head) * 16;
  const int m0 = blockIdx.x * 16;
  const float* __restrict__ W = g.W[head];
  const int K = g.K, N = g.N;
  const int kw = ((K + 63) / 64) * 16;  // k range of a wave, a multiple of 16
  const int k0 = wave * kw, k1 = min(K, k0 + kw);
  const int row = min(m0 + l15, g.M - 1);  // clamped: rows past M are computed and not stored
  const int n = n0 + l15;
  const bool nok = n < N;
  const float* __restrict__ xrow = g.X + (long long)row * K;
  const float* __restrict__ wcol = W + (long long)(nok ? n : 0) * g.w_sn;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int kb = k0; kb < k1; kb += 16) {
    const int k = kb + 4 * lq;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    float b[4] = {0.f, 0.f, 0.f, 0.f};
    if (k < k1) {  // K % 4 == 0 and k1 % 4 == 0: a float4 is entirely inside or outside
      a = *reinterpret_cast<const f32x4*>(xrow + k);
      if (nok) {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = wcol[(long long)(k + j) * g.w_sk];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][r * 64 + lane] = acc[r];
  __syncthreads();
  if (wave != 0 || !nok) return;
  const float bias = g.b[head] ? g.b[head][n] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + 4 * lq + r;  // D[i = 4 lq + r][j = l15]
    if (m < g.M) {
      const int o = r * 64 + lane;
      g.Y[head][(long long)m * N + n] = ((red[0][o] + red[1][o]) + (red[2][o] + red[3][o])) + bias;
    }
  }
}

}  // namespace

extern "C" int mvk_heads_fwd(const float* X, const float* W0, const float* b0, float* Y0, const float* W1, const float* b1,
                             float* Y1, int M, int N, int K, int64_t w_sk, int64_t w_sn, void* stream) {
  if (M == 0) return MVK_OK;
  if (!X || !W0 || !Y0 || (W1 && !Y1) || M < 0 || N <= 0 || N > 32 || K <= 0 || K % 4 != 0 || !mvk_aligned16(X))
    return MVK_EINVAL;
  HeadsArgs a{};
  a.X = X;
  a.W[0] = W0;
  a.b[0] = b0;
  a.Y[0] = Y0;
  a.W[1] = W1;
  a.b[1] = b1;
  a.Y[1] = Y1;
  a.M = M;
  a.N = N;
  a.K = K;
  a.tiles_per_head = (N + 15) / 16;
  a.w_sk = w_sk;
  a.w_sn = w_sn;
  const int heads = W1 ? 2 : 1;
  hipLaunchKernelGGL(heads_fwd_kernel, dim3((M + 15) / 16, heads * a.tiles_per_head), dim3(256), 0, mvk_stream(stream), a);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}
