// Narrow-output linear layers in ONE launch: the (embedding, log-covariance) heads of the encoders
// (reference: models/nn/default_architectures.py Encoder_VAE_MLP.embedding / .log_var, models/nn/svhn.py:29-30 the two
// Conv2d(128, L, 4, 2, 0) heads of Encoder_VAE_SVHN).  N <= 32 outputs per head from K = 400 ... 2048 inputs over a batch
// of a few hundred rows is 0.04-0.09 GFLOP: on the tiled GEMM engine each head was a split-K launch plus its reduce
// (4 launches of 5-10 us in a launch-latency-bound part of the step).  Here a workgroup owns a [16 rows] x [16 columns]
// tile of one head, its 4 waves split K, v_mfma_f32_16x16x4_f32 (exact fp32) accumulates, and the 4 partial tiles are
// added in a fixed order through LDS (deterministic).
#include <cstdlib>

#include "bf3.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct HeadsArgs {
  const float* X;
  const float* W[2];
  const float* b[2];
  float* Y[2];
  int M, N, K, tiles_per_head, act, wvec;  // wvec: 16-byte weight loads (w_sk == 1, rows and bases 16-byte aligned)
  long long w_sk, w_sn;  // W(k, n) = W[k * w_sk + n * w_sn]
  // K split over gridDim.z workgroups (few output tiles, long reductions: the heads of the ResNet encoders, K = 12544 ... 65536
  // from 32 ... 128 rows — four workgroups walked K = 12544 in 57 us, latency-bound): raw sums go to part[z][head][M][N],
  // heads_finish_kernel adds the slices in order and applies bias + activation
  float* part;
  int kz;  // k range of a z slice (a multiple of 16)
};

template <int NW>  // waves per workgroup = K slices (4 for short reductions, 16 for K >= 1024: the loop is latency-bound)
__global__ __launch_bounds__(NW * 64) void heads_fwd_kernel(const HeadsArgs g) {
  __shared__ float red[NW][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  const int head = blockIdx.y / g.tiles_per_head, n0 = (blockIdx.y % g.tiles_per_head) * 16;
  const int m0 = blockIdx.x * 16;
  const float* __restrict__ W = g.W[head];
  const int K = g.K, N = g.N;
  const int kbeg = g.part ? (int)blockIdx.z * g.kz : 0, kend = g.part ? min(K, kbeg + g.kz) : K;
  const int kw = ((kend - kbeg + 16 * NW - 1) / (16 * NW)) * 16;  // k range of a wave, a multiple of 16
  const int k0 = kbeg + wave * kw, k1 = min(kend, k0 + kw);
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
        if (g.wvec) {  // torch Linear rows: 4 consecutive k
          const f32x4 t = *reinterpret_cast<const f32x4*>(wcol + k);
#pragma unroll
          for (int j = 0; j < 4; ++j) b[j] = t[j];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) b[j] = wcol[(long long)(k + j) * g.w_sk];
        }
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
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; w += 4) t += (red[w][o] + red[w + 1][o]) + (red[w + 2][o] + red[w + 3][o]);
      if (g.part)
        g.part[(((long long)blockIdx.z * (gridDim.y / g.tiles_per_head) + head) * g.M + m) * N + n] = t;
      else
        g.Y[head][(long long)m * N + n] = mvk_act(t + bias, g.act);
    }
  }
}


// ONE head of up to 32 outputs over MANY rows (the decoders' gradient into the latent at the decoder batch: 5120 rows, N = 20, K = 2048 /
// 512 — until round 6 a split-K launch of the tiled engine plus its reduce, 23 + 27 us at the end of the decoder window's chains):
// a workgroup owns 16 rows, its 4 waves split K, and every wave accumulates BOTH 16-column tiles, so X is read once (the
// two-tile grid of heads_fwd_kernel read it twice).  Same exact-fp32 MFMA, same fixed summation order per output.
__global__ __launch_bounds__(256) void narrow_fwd_kernel(const HeadsArgs g) {
  __shared__ float red[4][2][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  const int m0 = blockIdx.x * 16;
  const float* __restrict__ W = g.W[0];
  const int K = g.K, N = g.N;
  const int kw = ((K + 63) / 64) * 16;  // k range of a wave, a multiple of 16
  const int k0 = wave * kw, k1 = min(K, k0 + kw);
  const int row = min(m0 + l15, g.M - 1);
  const int n_0 = l15, n_1 = 16 + l15;
  const bool ok0 = n_0 < N, ok1 = n_1 < N;
  const float* __restrict__ xrow = g.X + (long long)row * K;
  const float* __restrict__ w0 = W + (long long)(ok0 ? n_0 : 0) * g.w_sn;
  const float* __restrict__ w1 = W + (long long)(ok1 ? n_1 : 0) * g.w_sn;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int kb = k0; kb < k1; kb += 16) {
    const int k = kb + 4 * lq;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    float b0[4] = {0.f, 0.f, 0.f, 0.f}, b1[4] = {0.f, 0.f, 0.f, 0.f};
    if (k < k1) {
      a = *reinterpret_cast<const f32x4*>(xrow + k);
      if (g.wvec) {
        if (ok0) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(w0 + k);
#pragma unroll
          for (int j = 0; j < 4; ++j) b0[j] = t[j];
        }
        if (ok1) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(w1 + k);
#pragma unroll
          for (int j = 0; j < 4; ++j) b1[j] = t[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (ok0) b0[j] = w0[(long long)(k + j) * g.w_sk];
          if (ok1) b1[j] = w1[(long long)(k + j) * g.w_sk];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b0[j], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b1[j], acc1, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    red[wave][0][r * 64 + lane] = acc0[r];
    red[wave][1][r * 64 + lane] = acc1[r];
  }
  __syncthreads();
  if (wave > 1) return;  // wave 0 finishes the first column tile, wave 1 the second
  const int n = wave * 16 + l15;
  if (n >= N) return;
  const float bias = g.b[0] ? g.b[0][n] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + 4 * lq + r;
    if (m < g.M) {
      const int o = r * 64 + lane;
      const float t = (red[0][wave][o] + red[1][wave][o]) + (red[2][wave][o] + red[3][wave][o]);
      g.Y[0][(long long)m * N + n] = mvk_act(t + bias, g.act);
    }
  }
}

// Y[head][m][n] = act(sum_z part[z][head][m][n] + b[head][n]), the slices in order (deterministic)
__global__ __launch_bounds__(256) void heads_finish_kernel(const HeadsArgs g, int heads, int nz) {
  const long long per = (long long)g.M * g.N;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= per * heads) return;
  const int head = (int)(i / per);
  const long long r = i - head * per;
  const int n = (int)(r % g.N);
  float t = 0.f;
  int z = 0;
  for (; z + 3 < nz; z += 4) {  // four loads in flight, added in z order
    const float a0 = g.part[((long long)(z + 0) * heads + head) * per + r], a1 = g.part[((long long)(z + 1) * heads + head) * per + r];
    const float a2 = g.part[((long long)(z + 2) * heads + head) * per + r], a3 = g.part[((long long)(z + 3) * heads + head) * per + r];
    t = (((t + a0) + a1) + a2) + a3;
  }
  for (; z < nz; ++z) t += g.part[((long long)z * heads + head) * per + r];
  g.Y[head][r] = mvk_act(t + (g.b[head] ? g.b[head][n] : 0.f), g.act);
}

// ---- backward of the heads in ONE launch -----------------------------------------------------------------------------------
// Six launches of the tiled engine before (two weight gradients with their split-K finishes, two bias column sums, two
// backward-data GEMMs accumulating into one buffer) in the launch-latency-bound encoder backward.  A workgroup owns
// HB_KC = 16 columns k of X and HB_MR = 128 rows m:
//   dX[m][k]      = (sum_h sum_n dY_h[m][n] W_h(n, k)) * act'(X[m][k])                   (exact fp32 FMA chains, fixed order)
//   wslab_h[rg][.] = sum over its rows of dY_h[m][n] X[m][k]      (rows in order; the RG row groups are added by the caller's
//   bslab_h[rg][n] = sum over its rows of dY_h[m][n]               ordered finish: deterministic)
//   pslab[rg][k]   = sum over its rows of dX[m][k]                 (the bias gradient of the layer that produced X)
constexpr int HB_KC = 16, HB_MR = 128, HB_XS = 20;
struct HeadsBwdArgs {
  const float* X;
  const float* dY[2];
  const float* W[2];
  float* dX;
  float* wslab[2];
  float* bslab[2];
  float* pslab;
  int M, N, K, nh, x_act, flat_c;
  long long w_sk, w_sn;  // W_h(n, k) = W_h[n * w_sn + k * w_sk]
};

#ifndef MVK_CHAIN_PRIO
#define MVK_CHAIN_PRIO 3  // wave priority of the latency-critical chain kernels (s_setprio; 0 = default)
#endif
__global__ __launch_bounds__(256) void heads_bwd_kernel(const HeadsBwdArgs g) {
  // This launch heads the step's last dependent chain and runs beside the decoder's weight gradients (one 512-register MFMA wave
  // per SIMD, all 256 CUs): 17 us alone, 60-90 us there.  A higher wave priority wins the SIMD's issue arbitration against that wave.
  if (MVK_CHAIN_PRIO) __builtin_amdgcn_s_setprio(MVK_CHAIN_PRIO);
  // LDS sized by the heads' width (dynamic): 48 KB for the 2 x 20 latents of the MnistSvhn encoders instead of a fixed 62 KB — the
  // launch runs beside the decoder's weight gradients (95 / 112 KB of a CU's 160 KB): at 62 KB no workgroup of it fits beside the
  // larger one
  extern __shared__ __attribute__((aligned(16))) float hb_smem[];
  const int tid = threadIdx.x, N = g.N, K = g.K, NN = g.nh * N, NNP = NN | 1;
  float* const dYs = hb_smem;                                  // [HB_MR][NNP]
  float* const Ws = dYs + ((HB_MR * NNP + 3) & ~3);            // [NN][HB_KC]
  float* const Xs = Ws + NN * HB_KC;                           // [HB_MR][HB_XS]
  float* const Ds = Xs + HB_MR * HB_XS;                        // [HB_MR][HB_XS]
  float* const red = Ds + HB_MR * HB_XS;                       // [16 * 64]
  const int k0 = blockIdx.x * HB_KC, rg = blockIdx.y, r0 = rg * HB_MR;
  const int rows = min(HB_MR, g.M - r0);
  {  // rows past the end are zero everywhere below: every loop over the rows has the fixed trip count HB_MR
    const int c = tid & 63, mq = tid >> 6;
    if (c < NN) {
      const int h = c >= N ? 1 : 0, n = c - h * N;
      const float* __restrict__ src = g.dY[h] + (long long)r0 * N + n;
#pragma unroll 8
      for (int m = mq; m < HB_MR; m += 4) dYs[m * NNP + c] = m < rows ? src[(long long)m * N] : 0.f;
    }
  }
  for (int i = tid; i < NN * HB_KC; i += 256) {
    const int c = i / HB_KC, kk = i % HB_KC, h = c >= N ? 1 : 0, n = c - h * N;
    Ws[i] = g.W[h][(long long)n * g.w_sn + (long long)(k0 + kk) * g.w_sk];
  }
#pragma unroll
  for (int i = tid; i < HB_MR * (HB_KC / 4); i += 256) {
    const int m = i / (HB_KC / 4), q = i % (HB_KC / 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (m < rows) v = *reinterpret_cast<const f32x4*>(g.X + (long long)(r0 + m) * K + k0 + 4 * q);
    *reinterpret_cast<f32x4*>(Xs + m * HB_XS + 4 * q) = v;
  }
  __syncthreads();
  {  // backward data: thread = (row, half of the 16 columns)
    const int m = tid >> 1, kh = (tid & 1) * 8;
    // (packed fp32 FMAs: the same chain per column in the same order, half the vector-ALU instructions)
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    f32x2_ pa[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll 4
    for (int c = 0; c < NN; ++c) {
      const float dy = dYs[m * NNP + c];
      const f32x2_ dd = {dy, dy};
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(Ws + c * HB_KC + kh);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(Ws + c * HB_KC + kh + 4);
      pa[0] = __builtin_elementwise_fma(dd, f32x2_{w0[0], w0[1]}, pa[0]);
      pa[1] = __builtin_elementwise_fma(dd, f32x2_{w0[2], w0[3]}, pa[1]);
      pa[2] = __builtin_elementwise_fma(dd, f32x2_{w1[0], w1[1]}, pa[2]);
      pa[3] = __builtin_elementwise_fma(dd, f32x2_{w1[2], w1[3]}, pa[3]);
    }
    float acc[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[2 * e] = pa[e][0];
      acc[2 * e + 1] = pa[e][1];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      acc[e] *= mvk_act_grad_from_out(Xs[m * HB_XS + kh + e], g.x_act);
      Ds[m * HB_XS + kh + e] = acc[e];
    }
    if (m < rows && g.dX) {
      float* o = g.dX + (long long)(r0 + m) * K + k0 + kh;
      *reinterpret_cast<f32x4*>(o) = f32x4{acc[0], acc[1], acc[2], acc[3]};
      *reinterpret_cast<f32x4*>(o + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
    }
  }
  if (tid < NN * (HB_KC / 4)) {  // weight gradients: thread = (head column, 4 columns of X), rows in order
    const int c = tid / (HB_KC / 4), q = tid % (HB_KC / 4), h = c >= N ? 1 : 0, n = c - h * N;
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    f32x2_ g01 = {0.f, 0.f}, g23 = {0.f, 0.f};
#pragma unroll 8
    for (int m = 0; m < HB_MR; ++m) {
      const float dy = dYs[m * NNP + c];
      const f32x2_ dd = {dy, dy};
      const f32x4 x = *reinterpret_cast<const f32x4*>(Xs + m * HB_XS + 4 * q);
      g01 = __builtin_elementwise_fma(dd, f32x2_{x[0], x[1]}, g01);
      g23 = __builtin_elementwise_fma(dd, f32x2_{x[2], x[3]}, g23);
    }
    const f32x4 acc = {g01[0], g01[1], g23[0], g23[1]};
    float* slab = g.wslab[h] + (long long)rg * N * K + (long long)n * K;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + 4 * q + e;
      slab[g.flat_c > 0 ? (k % g.flat_c) * (K / g.flat_c) + k / g.flat_c : k] = acc[e];
    }
  }
  __syncthreads();  // Ds complete
  // column sums: 16 row groups of 8 rows, then the groups in order (fixed order: deterministic)
  const bool want_b = blockIdx.x == 0;
  if (g.pslab) {
    const int k = tid & 15, mg = tid >> 4;
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += Ds[(mg * 8 + j) * HB_XS + k];
    red[mg * 16 + k] = t;
    __syncthreads();
    if (tid < HB_KC) {
      float u = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) u += red[q * 16 + tid];
      g.pslab[(long long)rg * K + k0 + tid] = u;
    }
    if (want_b) __syncthreads();
  }
  if (want_b) {  // bias gradients of the heads: 4 row groups of 32 rows
    const int c = tid & 63, mq = tid >> 6;
    float t = 0.f;
    if (c < NN) {
#pragma unroll 8
      for (int j = 0; j < 32; ++j) t += dYs[(mq * 32 + j) * NNP + c];
    }
    red[mq * 64 + c] = t;
    __syncthreads();
    if (tid < NN) {
      const int h = tid >= N ? 1 : 0, n = tid - h * N;
      g.bslab[h][rg * N + n] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
    }
  }
}

}  // namespace

namespace mvk {
// Y_h = X W_h + b_h for 1 or 2 narrow heads; 1 = shape not covered
int heads_launch(const float* X, const float* W0, const float* b0, float* Y0, const float* W1, const float* b1, float* Y1,
                 int M, int N, int K, long long w_sk, long long w_sn, int act, hipStream_t s, float* ws, long long ws_floats) {
  if ((W1 && N > 32) || N < 1 || K % 4 != 0 || K < 4 || !mvk_aligned16(X) || M < 1) return 1;
  const int wvec = w_sk == 1 && (w_sn & 3) == 0 && mvk_aligned16(W0) && (!W1 || mvk_aligned16(W1));
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
  a.act = act;
  a.wvec = wvec;
  a.w_sk = w_sk;
  a.w_sn = w_sn;
  const int heads = W1 ? 2 : 1;
  dim3 grid((M + 15) / 16, heads * a.tiles_per_head);
  static const int nw_env = mvk_tune("MVK_HEADS_WAVES") ? atoi(mvk_tune("MVK_HEADS_WAVES")) : 0;  // A/B switch
  // few tiles and a long reduction: split K over workgroups (MVK_HEADS_KSPLIT=0 switches it off)
  static const int ksplit_on = mvk_tune("MVK_HEADS_KSPLIT") ? atoi(mvk_tune("MVK_HEADS_KSPLIT")) : 1;
  const int tiles = (int)(grid.x * grid.y);
  int nz = 1;
  if (ksplit_on && ws && tiles <= 64 && K >= 2048) {
    nz = K / 256;                                   // >= 256 k per slice: one pass of 16 waves x 16
    const int cap = tiles >= 32 ? 64 : 128;         // ~2-4 k workgroups at most
    if (nz > cap) nz = cap;
    a.kz = ((K + nz - 1) / nz + 15) & ~15;
    nz = (K + a.kz - 1) / a.kz;
    if (nz < 2 || (long long)nz * heads * M * N > ws_floats) nz = 1;
  }
  static const int narrow_on = mvk_tune("MVK_NARROW_FWD") ? atoi(mvk_tune("MVK_NARROW_FWD")) : 1;  // A/B: 0 = the two-tile grid
  if (narrow_on && heads == 1 && N > 16 && N <= 32 && M >= 1024) {  // many rows, one head: both column tiles in one workgroup
    hipLaunchKernelGGL(narrow_fwd_kernel, dim3((M + 15) / 16), dim3(256), 0, s, a);
    MVK_CHECK_LAUNCH();
    return MVK_OK;
  }
  if (nz > 1) {
    a.part = ws;
    grid.z = nz;
    hipLaunchKernelGGL(heads_fwd_kernel<16>, grid, dim3(1024), 0, s, a);
    MVK_CHECK_LAUNCH();
    const long long total = (long long)heads * M * N;
    hipLaunchKernelGGL(heads_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, heads, nz);
    MVK_CHECK_LAUNCH();
    return MVK_OK;
  }
  if (nw_env ? nw_env == 16 : K >= 1024)
    hipLaunchKernelGGL(heads_fwd_kernel<16>, grid, dim3(1024), 0, s, a);
  else
    hipLaunchKernelGGL(heads_fwd_kernel<4>, grid, dim3(256), 0, s, a);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

// slabs: wslab_h [rg][N * K] (in the layout of dW_h), bslab_h [rg][N], pslab [rg][K] (optional) with rg = ceil(M / 128)
// row groups (*nz); 1 = shape not covered
int heads_bwd_launch(const float* X, int x_act, const float* dY0, const float* dY1, const float* W0, const float* W1,
                     long long w_sk, long long w_sn, int flat_c, float* dX, float* wslab0, float* wslab1, float* bslab0,
                     float* bslab1, float* pslab, int M, int N, int K, int* nz, hipStream_t s) {
  const int nh = dY1 ? 2 : 1;
  if (M < 1 || N < 1 || N > 32 || K % HB_KC != 0 || K < HB_KC || !mvk_aligned16(X) || (dX && !mvk_aligned16(dX))) return 1;
  if (flat_c > 0 && K % flat_c != 0) return 1;
  const int rgs = (M + HB_MR - 1) / HB_MR;
  if (rgs > 64) return 1;
  HeadsBwdArgs a{};
  a.X = X;
  a.dY[0] = dY0;
  a.dY[1] = dY1;
  a.W[0] = W0;
  a.W[1] = W1;
  a.dX = dX;
  a.wslab[0] = wslab0;
  a.wslab[1] = wslab1;
  a.bslab[0] = bslab0;
  a.bslab[1] = bslab1;
  a.pslab = pslab;
  a.M = M;
  a.N = N;
  a.K = K;
  a.nh = nh;
  a.x_act = x_act;
  a.flat_c = flat_c;
  a.w_sk = w_sk;
  a.w_sn = w_sn;
  const int nn = nh * N;
  const size_t lds = sizeof(float) * (size_t)(((HB_MR * (nn | 1) + 3) & ~3) + nn * HB_KC + 2 * HB_MR * HB_XS + 16 * 64);
  hipLaunchKernelGGL(heads_bwd_kernel, dim3(K / HB_KC, rgs), dim3(256), lds, s, a);
  MVK_CHECK_LAUNCH();
  *nz = rgs;
  return MVK_OK;
}
}  // namespace mvk

extern "C" int mvk_heads_fwd(const float* X, const float* W0, const float* b0, float* Y0, const float* W1, const float* b1,
                             float* Y1, int M, int N, int K, int64_t w_sk, int64_t w_sn, void* stream) {
  if (M == 0) return MVK_OK;
  if (!X || !W0 || !Y0 || (W1 && !Y1) || M < 0 || N <= 0 || N > 32 || K <= 0 || K % 4 != 0 || !mvk_aligned16(X))
    return MVK_EINVAL;
  const int rc = mvk::heads_launch(X, W0, b0, Y0, W1, b1, Y1, M, N, K, w_sk, w_sn, MVK_ACT_NONE, mvk_stream(stream), nullptr, 0);
  return rc == 1 ? MVK_EINVAL : rc;
}

// ---------------------------------------------------------------------------------------------------------------------
// Short-reduction linear layers: Y[M][N] = act(X[M][K] W + b) with K <= 32 (the first layer of every decoder: z of
// latent_dim 20 -> 400 hidden units / the 4x4x128 map of ConvTranspose2d(L, 128, 4, 1, 0), models/nn/svhn.py:51,
// default_architectures.py Decoder_AE_MLP).  At K = 20 the tiled MFMA engine pads the reduction to 32, splits both
// operands into bf16 pieces and spends 31 us on 0.4 GFLOP; the layer is a streaming write of the output (42 MB at
// K B = 5120 rows) with 20 FMAs per element.  Here a thread keeps W[:, n..n+3] in registers (K float4), a workgroup stages
// 32 rows of X in LDS (broadcast reads) and every thread writes 16 bytes per row: exact fp32 FMA chains in k order.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct SmallKArgs {
  const float* X;
  const float* W;
  const float* bias;
  float* Y;
  int M, N, K, bias_mod, act;
  long long w_sk, w_sn;  // W(k, n) = W[k * w_sk + n * w_sn]
  const float* mask_src;  // optional [M][N]: the result is multiplied by act'(mask_src) (backward-data into a hidden layer)
  int mask_act, accumulate;
  float* y_amax;          // optional: max |Y| is published here (atomic max, one per workgroup; must hold 0 before the launch)
};

// R rows per workgroup.  Every workgroup first loads its weight columns (40-80 KB at K = 20): with 8 rows the 1280 workgroups of
// the decoder batch (M = 5120, N = 2048) read 100 MB of weights from L2 to write 42 MB of output (29 us); the launcher picks R
// 16 rows there (640 workgroups; the rest of the 28 us were 1280 same-address atomics of the published maximum: amax_publish).
template <int K4, int R>  // K4 = ceil(K / 4)
__global__ __launch_bounds__(256) void smallk_fwd_kernel(const SmallKArgs g) {
  constexpr int KP = K4 * 4;
  __shared__ __attribute__((aligned(16))) float xs[R][KP];
  const int CT = g.N / 4;                    // float4 column groups
  const int ctb = CT < 256 ? CT : 256;       // column groups of this workgroup
  const int rgn = 256 / ctb;                 // row groups (threads beyond ctb * rgn idle)
  const int cgi = threadIdx.x % ctb, rg = threadIdx.x / ctb;
  const int cg = blockIdx.y * ctb + cgi;
  const bool active = rg < rgn && cg < CT;
  const int n = (cg < CT ? cg : 0) * 4;
  const int m0 = blockIdx.x * R;
  f32x4 w[KP];  // w[k] = W(k, n .. n+3)
#ifndef MVK_SK_ABL
#define MVK_SK_ABL 0  // subtraction builds (tools/smallk_probe.py): 1 no weight loads, 2 no stores, 4 no FMA loop
#endif
  if (MVK_SK_ABL & 1) {
#pragma unroll
    for (int k = 0; k < KP; ++k) w[k] = f32x4{0.01f * k, 0.02f, 0.03f, 0.04f * threadIdx.x};
  } else
  if (g.w_sn == 1 && (g.w_sk & 3) == 0 && mvk_dev_aligned16(g.W)) {  // [K][N] rows: one 16-byte load per k
#pragma unroll
    for (int k = 0; k < KP; ++k)
      w[k] = (k < g.K) ? *reinterpret_cast<const f32x4*>(g.W + (long long)k * g.w_sk + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  } else if (g.w_sk == 1 && (g.w_sn & 3) == 0 && (g.K & 3) == 0 && mvk_dev_aligned16(g.W)) {  // torch Linear [N][K]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < K4; ++q) {
        const f32x4 t = (4 * q < g.K) ? *reinterpret_cast<const f32x4*>(g.W + (long long)(n + j) * g.w_sn + 4 * q)
                                      : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) w[4 * q + e][j] = t[e];
      }
  } else {
#pragma unroll
    for (int k = 0; k < KP; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) w[k][j] = (k < g.K) ? g.W[(long long)k * g.w_sk + (long long)(n + j) * g.w_sn] : 0.f;
  }
  f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
  if (g.bias)
#pragma unroll
    for (int j = 0; j < 4; ++j) b4[j] = g.bias[(n + j) % g.bias_mod];
  for (int i = threadIdx.x; i < R * KP; i += 256) {
    const int r = i / KP, k = i - r * KP;
    xs[r][k] = (m0 + r < g.M && k < g.K) ? g.X[(long long)(m0 + r) * g.K + k] : 0.f;
  }
  __syncthreads();
  float amax_l = 0.f;
#ifndef MVK_SK_UNROLL
#define MVK_SK_UNROLL 1  // rows in flight per thread (a row is ONE dependent chain of K packed FMAs): 4 measured, no difference in the step
// (0.9735 / 0.9757 / 0.9742 / 0.9795 against 0.9761 / 0.9736 / 0.9772 / 0.9783 ms, tools/lab/r06/gpu_r06_j.sh): the round-5 loop stays
#endif
  if (active)
#pragma unroll MVK_SK_UNROLL
  for (int r = rg; r < R; r += rgn) {
    if (m0 + r >= g.M) continue;
    // packed fp32 FMAs (v_pk_fma_f32: two columns per instruction, the row's x broadcast to both halves): the same FMA chain per
    // column in the same k order, half the vector-ALU instructions of a loop that is bound by them (80 FMAs per 16-byte store)
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    f32x2_ a01 = {b4[0], b4[1]}, a23 = {b4[2], b4[3]};
#pragma unroll
    for (int q = 0; q < ((MVK_SK_ABL & 4) ? 1 : K4); ++q) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(&xs[r][4 * q]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x2_ xx = {x[e], x[e]};
        a01 = __builtin_elementwise_fma(xx, f32x2_{w[4 * q + e][0], w[4 * q + e][1]}, a01);
        a23 = __builtin_elementwise_fma(xx, f32x2_{w[4 * q + e][2], w[4 * q + e][3]}, a23);
      }
    }
    f32x4 acc = {a01[0], a01[1], a23[0], a23[1]};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = mvk_act(acc[j], g.act);
    f32x4* const dst = reinterpret_cast<f32x4*>(g.Y + (long long)(m0 + r) * g.N + n);
    if (g.mask_src) {
      const f32x4 ms = *reinterpret_cast<const f32x4*>(g.mask_src + (long long)(m0 + r) * g.N + n);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] *= mvk_act_grad_from_out(ms[j], g.mask_act);
    }
    if (g.accumulate) {
      const f32x4 old = *dst;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += old[j];
    }
    if (!(MVK_SK_ABL & 2) || acc[0] == 123.456f) *dst = acc;
    amax_l = fmaxf(fmaxf(amax_l, fmaxf(fabsf(acc[0]), fabsf(acc[1]))), fmaxf(fabsf(acc[2]), fabsf(acc[3])));
  }
  if (g.y_amax) mvk::amax_publish(amax_l, g.y_amax, &xs[0][0]);  // uniform; xs is dead (amax_publish synchronises first)
}

}  // namespace

namespace mvk {
// 1: shape not covered (the caller continues with the tiled engine)
int smallk_fwd(const float* X, const float* W, long long w_sk, long long w_sn, const float* bias, int bias_mod, int act,
               float* Y, int M, int N, int K, hipStream_t s, const float* mask_src, int mask_act, int accumulate, float* y_amax) {
  static const int off = mvk_tune("MVK_SMALLK") ? atoi(mvk_tune("MVK_SMALLK")) == 0 : 0;
  if (off || K > 32 || K < 1 || N % 4 != 0 || N < 4 || !mvk_aligned16(Y) || M < 1 || (mask_src && !mvk_aligned16(mask_src)))
    return 1;
  SmallKArgs a{X, W, bias, Y, M, N, K, bias_mod > 0 ? bias_mod : 1, act, w_sk, w_sn, mask_src, mask_act, accumulate, y_amax};
  const int CT = N / 4, ctb = CT < 256 ? CT : 256;
  const int gy = (CT + ctb - 1) / ctb;
  static const int r_env = mvk_tune("MVK_SMALLK_ROWS") ? atoi(mvk_tune("MVK_SMALLK_ROWS")) : 0;
  const long long want = r_env > 0 ? r_env : ((long long)M * gy + 255) / 256;  // rows per workgroup for ~256 workgroups
  // measured at M = 5120, N = 2048, K = 20: 8 rows 19.2 us, 16 rows 18.0, 24 rows 20.4, 40 rows 28.8 (one wave per SIMD: latency-bound)
  const int R = r_env > 0 ? (want <= 8 ? 8 : (want <= 16 ? 16 : (want <= 24 ? 24 : 40))) : (want <= 8 ? 8 : 16);
  const dim3 grid((M + R - 1) / R, gy);
#define MVK_SMALLK_CASE(K4_)                                                                                    \
  case K4_:                                                                                                     \
    if (R == 8) hipLaunchKernelGGL((smallk_fwd_kernel<K4_, 8>), grid, dim3(256), 0, s, a);                      \
    else if (R == 16) hipLaunchKernelGGL((smallk_fwd_kernel<K4_, 16>), grid, dim3(256), 0, s, a);               \
    else if (R == 24) hipLaunchKernelGGL((smallk_fwd_kernel<K4_, 24>), grid, dim3(256), 0, s, a);               \
    else hipLaunchKernelGGL((smallk_fwd_kernel<K4_, 40>), grid, dim3(256), 0, s, a);                            \
    break;
  switch ((K + 3) / 4 > 8 ? 8 : (K + 3) / 4) {
    MVK_SMALLK_CASE(1) MVK_SMALLK_CASE(2) MVK_SMALLK_CASE(3) MVK_SMALLK_CASE(4)
    MVK_SMALLK_CASE(5) MVK_SMALLK_CASE(6) MVK_SMALLK_CASE(7) MVK_SMALLK_CASE(8)
  }
#undef MVK_SMALLK_CASE
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}
}  // namespace mvk
