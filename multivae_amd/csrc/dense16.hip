// Dense layers on PRE-SPLIT fp16 pair planes: the MLP decoder of the MnistSvhn models at the decoder batch n = K B
// (reference: models/nn/default_architectures.py:225-258 Decoder_AE_MLP — Linear(L, 512) + ReLU, Linear(512, prod(input_dim)) +
// Sigmoid —, likelihood models/base/base_utils.py:62-87).
//
// The tiled engine (igemm_bf.hpp) converts every fp32 operand element into bf16 pieces in EVERY workgroup that touches it:
// for a [5120 x 512] x [512 x 784] GEMM on 128 x 64 tiles that is ~10 vector instructions per MFMA, and the three 4.1-GFLOP
// GEMMs of this decoder ran at 0.10 of the split-bf16 ceiling inside the step.  Here every operand reaches its GEMM already
// split into the two fp16 planes of bf3.hpp (x s = hi + lo / 2048; a product = hi hi' + (hi lo' + lo hi') / 2048, three
// v_mfma_f32_32x32x16_f16), written ONCE by its producer:
//   * the weight W1 [N][K] by d16_pack_kernel, once per step, in both orientations ([N][K] for the forward, [K][N] for
//     backward data), with one power-of-two scale PER ROW of the plane (the row index is never the reduction index of the
//     GEMM that reads the plane, so the scale is a column factor of the result: local, no pass for a global maximum);
//   * the hidden activation h = relu(z W0^T + b0) by d16_first_kernel (K <= 32: weights in registers), under the a-priori
//     bound  max_n ||W0[n, :]||_1 max|z| + max|b0|  (every workgroup holds the whole W0 and derives the same number);
//   * the gradient of the likelihood by the forward GEMM itself: its epilogue applies bias + sigmoid, scores the row
//     against the data (Normal(scale): NLL row sums) and stores  gw d NLL / d pre-activation  as planes under the bound
//     gw (1 + max|x|) / (4 scale^2)  — neither the reconstruction nor its gradient exists as an fp32 tensor.
// A scale needs an upper bound only (a loose bound costs range, not precision: bf3.hpp), so no pass over a tensor is added.
// The GEMM kernels therefore move 16-byte pieces global -> registers -> LDS -> MFMA with NO arithmetic in the loop.
//
//   d16_nt_kernel   C[m][n] = sum_k A[m][k] B[n][k]   (forward + fused tail; backward data with the ReLU mask of h)
//   d16_tn_kernel   C[j][i] = sum_m D[m][j] H[m][i]   (weight gradient: both operands reduction-major, transposing LDS reads)
#include <cmath>
#include <cstdlib>

#include "bf3.hpp"

namespace {
using mvk::f16x2;
using mvk::f16x8;
using mvk::f32x16;
using mvk::f32x2;
using mvk::f32x4;
using mvk::u32x2;
using mvk::u32x4;

typedef _Float16 half_t;

// ---------------------------------------------------------------------------------------------------------------------
// Weight planes.  role NK: plane row n = W[n][:], role KN: plane row k = W[:][k]; inv[row] = 1 / (power-of-two scale of the row)
// ---------------------------------------------------------------------------------------------------------------------
struct D16Pack {
  const float* W;  // [N][K]
  int N, K, nk_blocks;
  half_t *nk_hi, *nk_lo;
  float* nk_inv;   // [N]
  half_t *kn_hi, *kn_lo;
  float* kn_inv;   // [K]
};

__global__ __launch_bounds__(256) void d16_pack_kernel(const D16Pack g) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((int)blockIdx.x < g.nk_blocks) {  // a wave per row n
    const int n = blockIdx.x * 4 + wave;
    if (n >= g.N) return;
    const float* row = g.W + (long long)n * g.K;
    float m = 0.f;
    for (int k = lane * 4; k < g.K; k += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(row + k);
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    m = wave_max(m);
    const float s = mvk::f16_scale_of(m);
    if (lane == 0) g.nk_inv[n] = mvk::f16_inv_scale(s);
    for (int k = lane * 4; k < g.K; k += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(row + k);
      unsigned h0, l0, h1, l1;
      mvk::f16_split(v[0] * s, v[1] * s, h0, l0);
      mvk::f16_split(v[2] * s, v[3] * s, h1, l1);
      *reinterpret_cast<u32x2*>(g.nk_hi + (long long)n * g.K + k) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(g.nk_lo + (long long)n * g.K + k) = u32x2{l0, l1};
    }
    return;
  }
  // KN: this workgroup owns 16 columns k0 .. k0+15 of W = 16 rows of the transposed plane; thread = (n lane 0..63, k quad 0..3)
  __shared__ float red[64][17];
  const int k0 = (blockIdx.x - g.nk_blocks) * 16;
  const int nr = tid >> 2, kq = tid & 3;
  float m[4] = {0.f, 0.f, 0.f, 0.f};
  for (int n = nr; n < g.N; n += 64) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(g.W + (long long)n * g.K + k0 + 4 * kq);
#pragma unroll
    for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], fabsf(v[e]));
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[nr][4 * kq + e] = m[e];
  __syncthreads();
  if (tid < 16) {
    float mm = 0.f;
    for (int r = 0; r < 64; ++r) mm = fmaxf(mm, red[r][tid]);
    const float s = mvk::f16_scale_of(mm);
    red[0][tid] = s;  // only column tid of row 0 is rewritten, by the thread that read it last
    g.kn_inv[k0 + tid] = mvk::f16_inv_scale(s);
  }
  __syncthreads();
  float s4[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) s4[e] = red[0][4 * kq + e];
  for (int n = nr; n < g.N; n += 64) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(g.W + (long long)n * g.K + k0 + 4 * kq);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned h, l;
      mvk::f16_split(v[e] * s4[e], 0.f, h, l);
      const long long o = (long long)(k0 + 4 * kq + e) * g.N + n;
      reinterpret_cast<unsigned short*>(g.kn_hi)[o] = (unsigned short)h;
      reinterpret_cast<unsigned short*>(g.kn_lo)[o] = (unsigned short)l;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// First layer: H planes = act(Z[M][K] W0[N][K]^T + b0), K <= 32, N <= 1024 (every workgroup holds all of W0 in registers)
// ---------------------------------------------------------------------------------------------------------------------
struct D16First {
  const float* Z;
  const float* W;  // [N][K] (torch Linear)
  const float* bias;
  const float* z_amax;  // device scalar >= max |Z|
  half_t *hi, *lo;      // [M][N]
  float* bound;         // receives the bound the planes are scaled by
  int M, N, K, act;
};

// R rows per workgroup: every workgroup loads all of W (40 KB at 512 x 20) first; the launcher picks R for ~256 workgroups
template <int K4, int R>
__global__ __launch_bounds__(256) void d16_first_kernel(const D16First g) {
  constexpr int KP = K4 * 4;
  __shared__ __attribute__((aligned(16))) float xs[R][KP];
  __shared__ float red[8];
  const int CT = g.N / 4;       // <= 256 column groups: thread = (column group, row group)
  const int rgn = 256 / CT;
  const int cg = threadIdx.x % CT, rg = threadIdx.x / CT;
  const bool active = rg < rgn;
  const int n = cg * 4;
  const int m0 = blockIdx.x * R;
  f32x4 w[KP];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int q = 0; q < K4; ++q) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      if (4 * q < g.K) {  // K % 4 == 0 (checked by the launcher)
        t = *reinterpret_cast<const f32x4*>(g.W + (long long)(n + j) * g.K + 4 * q);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) w[4 * q + e][j] = t[e];
    }
  f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
  if (g.bias) b4 = *reinterpret_cast<const f32x4*>(g.bias + n);
  // the a-priori bound: max_n sum_k |W(n, k)| * max|z| + max_n |b(n)| (identical in every workgroup: same data, same order)
  float l1 = 0.f, bm = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float sj = 0.f;
#pragma unroll
    for (int k = 0; k < KP; ++k) sj += fabsf(w[k][j]);
    l1 = fmaxf(l1, sj);
    bm = fmaxf(bm, fabsf(b4[j]));
  }
  l1 = wave_max(l1);
  bm = wave_max(bm);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = l1;
    red[4 + (threadIdx.x >> 6)] = bm;
  }
  for (int i = threadIdx.x; i < R * KP; i += 256) {
    const int r = i / KP, k = i - r * KP;
    xs[r][k] = (m0 + r < g.M && k < g.K) ? g.Z[(long long)(m0 + r) * g.K + k] : 0.f;
  }
  __syncthreads();
  l1 = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  bm = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
  const float bound = l1 * (*g.z_amax) * 1.0001f + bm;  // 1.0001: the fp32 FMA chain may round above the exact bound
  if (blockIdx.x == 0 && threadIdx.x == 0) *g.bound = bound;
  const float s = mvk::f16_scale_of(bound);
  if (active)
    for (int r = rg; r < R && m0 + r < g.M; r += rgn) {
      f32x4 acc = b4;
#pragma unroll
      for (int q = 0; q < K4; ++q) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(&xs[r][4 * q]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(x[e], w[4 * q + e][j], acc[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = mvk_act(acc[j], g.act);
      unsigned h0, l0, h1, l1_;
      mvk::f16_split(acc[0] * s, acc[1] * s, h0, l0);
      mvk::f16_split(acc[2] * s, acc[3] * s, h1, l1_);
      const long long o = (long long)(m0 + r) * g.N + n;
      *reinterpret_cast<u32x2*>(g.hi + o) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(g.lo + o) = u32x2{l0, l1_};
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// C[m][n] = sum_k A[m][k] B[n][k] on planes.  128 (or 64) x 128 tile, 4 waves as 2 x 2, a wave = (BM / 2) x 64 = TM x 2 MFMA
// tiles with a main and a cross accumulator each.  k-tile = 32: a plane tile is [rows][64 bytes] in LDS, the 16-byte k-octet
// `o` of row `r` at o ^ ((r >> 2) & 3) (conflict-free ds_write_b128 of the staging and ds_read_b128 of the fragments); two
// LDS stages, one barrier per k-tile; the next tile's global loads are issued before the MFMAs of the current one.
// ---------------------------------------------------------------------------------------------------------------------
enum { D16_NLL = 0, D16_BWD = 1 };
#ifndef D16_NT128_OCC
#define D16_NT128_OCC 1  // workgroups per CU of the 128-row tile (its two register sets put it above 256 registers)
#endif

// (mvk_fast_sigmoid: common.hpp)
#define d16_sigmoid mvk_fast_sigmoid

struct D16Nt {
  const half_t *Ah, *Al;  // [M][K]
  const half_t *Bh, *Bl;  // [N][K]
  const float* a_bound;   // device scalar: the planes of A hold A * f16_scale_of(*a_bound)
  const float* b_inv;     // [N]: 1 / scale of plane row n
  int M, N, K;
  // D16_NLL: v = C + bias; r = sigmoid(v); rows += (r - x)^2 / (2 s^2); G = gw (r - x) / s^2 r (1 - r) -> planes
  const float* bias;      // [N]
  const float* X;         // [xrows][N]; row m is scored against X[m % xrows]
  const float* x_amax;    // device scalar >= max |X|
  int xrows;
  float inv_s2, gw, row_const;
  half_t *Gh, *Gl;        // [M][N]
  float* g_bound;         // receives the bound of G's planes
  float* rows_part;       // [gridDim.y][M] partial NLL row sums (row_const added in column tile 0)
  float* colsum_part;     // [gridDim.x][N] column sums of G per row tile (the bias gradient's partials)
  // D16_BWD: out = C * (mask_hi > 0)
  const half_t* mask_hi;  // [M][N] hi plane of the activation the result lands in (ReLU), or null
  float* out;             // [M][N] fp32
  mvk_prof_slot* prof;    // device-timestamp record (null: profiler off)
  int dbg;                // experiment switches (mvk_dense16_debug): 1 no global loads, 2 no MFMAs, 4 no LDS writes, 8 no epilogue
  float* stamps;          // dbg & 16: 4 floats of cycle stamps (mvk_dense16_debug_stamps; null = none are written)
};

template <int BM, int EPI>
#ifndef D16_NT64_BWD_OCC
#define D16_NT64_BWD_OCC 2  // occupancy bound of the 64-row backward-data form.  4 = at most 128 registers, so that a wave fits beside a
// register-stationary one (imgconv DOWN forms hold 348-368 of a SIMD's 512 registers; this kernel's 173 do not fit and its launch
// waits for theirs to end): measured, it DOES run beside them then — with 87 spilled registers, 227 us instead of 36, step +9 %.
#endif
#ifndef D16_NT64_FWD_OCC
#define D16_NT64_FWD_OCC 2
#endif
__global__ __launch_bounds__(256, BM == 64 ? (EPI == 1 ? D16_NT64_BWD_OCC : D16_NT64_FWD_OCC) : D16_NT128_OCC) void d16_nt_kernel(const D16Nt g) {
  constexpr int BN = 128, TM = BM / 64, TN = 2;
  constexpr int APL = BM * 64, BPL = BN * 64;  // bytes per plane tile
  constexpr int STAGE = 2 * APL + 2 * BPL;
  constexpr int NA = BM * 4 / 256, NB = BN * 4 / 256;  // 16-byte chunks per plane and thread
  extern __shared__ __attribute__((aligned(16))) char lds[];
  mvk_prof_begin(g.prof);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int K = g.K;

  const __amdgpu_buffer_rsrc_t rsAh = __builtin_amdgcn_make_buffer_rsrc((void*)g.Ah, 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsAl = __builtin_amdgcn_make_buffer_rsrc((void*)g.Al, 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsBh = __builtin_amdgcn_make_buffer_rsrc((void*)g.Bh, 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsBl = __builtin_amdgcn_make_buffer_rsrc((void*)g.Bl, 0, 0x7ffffff0, 0x00020000);

  int a_g[NA], a_l[NA], a_kmax[NA], b_g[NB], b_l[NB], b_kmax[NB];
#pragma unroll
  for (int u = 0; u < NA; ++u) {
    const int idx = tid + u * 256, row = idx >> 2, oct = idx & 3;
    a_l[u] = row * 64 + ((oct ^ ((row >> 2) & 3)) << 4);
    a_g[u] = ((m0 + row) * K + oct * 8) * 2;
    a_kmax[u] = (m0 + row < g.M) ? K - oct * 8 : 0;  // the chunk at k0 is inside the row iff k0 < kmax
  }
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const int idx = tid + u * 256, row = idx >> 2, oct = idx & 3;
    b_l[u] = row * 64 + ((oct ^ ((row >> 2) & 3)) << 4);
    b_g[u] = ((n0 + row) * K + oct * 8) * 2;
    b_kmax[u] = (n0 + row < g.N) ? K - oct * 8 : 0;
  }
  // two register sets: a k-tile's loads are issued two iterations before its LDS write (one iteration = the MFMAs of one k-tile
  // is shorter than the latency of an L2 miss: with one set every iteration waited for its loads)
  struct Raw {
    u32x4 ah[NA], al[NA], bh[NB], bl[NB];
  };
#ifndef D16_BWD_ONESET
#define D16_BWD_ONESET 0  // 1: the 64-row backward-data form loads ONE k-tile ahead (one register set: 173 -> 149 registers, no spills).
// It then fits beside imgconv<DOWN, 4, 64, 128> (352 + 152) and runs there: 106 -> 133 us while that launch goes 106 -> 130 us,
// step +0.7 % (four same-box pairs).  Two MFMA-heavy kernels on one SIMD share a power-limited matrix pipe: co-residency is zero-sum.
#endif
#ifndef D16_FWD_ONESET
#define D16_FWD_ONESET 0  // the same for the 64-row forward form (fused tail)
#endif
  constexpr bool ONESET = BM == 64 && ((EPI == 1 && D16_BWD_ONESET) || (EPI == 0 && D16_FWD_ONESET));
  Raw R0, R1;
  auto sel = [](bool ok, int off) { return __builtin_unpredictable(ok) ? off : 0x7fffffff; };  // out of range = zero fill
  auto gload = [&](Raw& r, int k0) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int off = sel(k0 < a_kmax[u], a_g[u] + k0 * 2);
      r.ah[u] = __builtin_amdgcn_raw_buffer_load_b128(rsAh, off, 0, 0);
      r.al[u] = __builtin_amdgcn_raw_buffer_load_b128(rsAl, off, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int off = sel(k0 < b_kmax[u], b_g[u] + k0 * 2);
      r.bh[u] = __builtin_amdgcn_raw_buffer_load_b128(rsBh, off, 0, 0);
      r.bl[u] = __builtin_amdgcn_raw_buffer_load_b128(rsBl, off, 0, 0);
    }
  };
  auto lwrite = [&](const Raw& r, char* st) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      *reinterpret_cast<u32x4*>(st + a_l[u]) = r.ah[u];
      *reinterpret_cast<u32x4*>(st + APL + a_l[u]) = r.al[u];
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      *reinterpret_cast<u32x4*>(st + 2 * APL + b_l[u]) = r.bh[u];
      *reinterpret_cast<u32x4*>(st + 2 * APL + BPL + b_l[u]) = r.bl[u];
    }
  };

  f32x16 accm[TM][TN], accc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) accm[a][b][r] = accc[a][b][r] = 0.f;

  // fragment addresses: row (wave tile + 32 a + l31) * 64 + ((2 ks + lhi) ^ swz) * 16, swz = (row >> 2) & 3 = (l31 >> 2) & 3
  const int swz = (l31 >> 2) & 3;
  const int so[2] = {((0 + lhi) ^ swz) << 4, ((2 + lhi) ^ swz) << 4};
  const int afr = (wm * (BM / 2) + l31) * 64, bfr = 2 * APL + (wn * 64 + l31) * 64;
  constexpr int NF = 2 * (TM + TN);  // fragments of one 16-wide k-step: A hi, A lo, B hi, B lo
  auto rfrag = [&](f16x8 (&F)[NF], const char* st, int ks) {
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      F[a] = *reinterpret_cast<const f16x8*>(st + afr + a * 32 * 64 + so[ks]);
      F[TM + a] = *reinterpret_cast<const f16x8*>(st + APL + afr + a * 32 * 64 + so[ks]);
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      F[2 * TM + b] = *reinterpret_cast<const f16x8*>(st + bfr + b * 32 * 64 + so[ks]);
      F[2 * TM + TN + b] = *reinterpret_cast<const f16x8*>(st + BPL + bfr + b * 32 * 64 + so[ks]);
    }
  };
  auto mfmas = [&](const f16x8 (&F)[NF]) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) accm[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[a], F[2 * TM + b], accm[a][b], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) accc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[a], F[2 * TM + TN + b], accc[a][b], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) accc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[TM + a], F[2 * TM + b], accc[a][b], 0, 0, 0);
  };
  // "1 MFMA, N others" (imgconv_kernel): hipcc otherwise clusters the MFMAs of a k-step behind all of its LDS reads
#define D16_INTERLEAVE(OTHERS)                                 \
  _Pragma("unroll") for (int i_ = 0; i_ < 3 * TM * TN; ++i_) { \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         \
    __builtin_amdgcn_sched_group_barrier(0x496, OTHERS, 0);    \
  }

  // Software pipeline: the k-tile in LDS stage `cur` is multiplied while tile t+1 goes from registers to stage `nxt` and
  // tile t+2 is fetched; the fragments of a k-step are read one k-step ahead; ONE barrier per k-tile, in its middle.
  const int nt = (K + 31) / 32;
  f16x8 FA[NF], FB[NF];
  const unsigned long long tk0 = (g.dbg & 16) ? __builtin_readcyclecounter() : 0ull;
  gload(R0, 0);
  lwrite(R0, lds);
  if (ONESET) {
    gload(R0, 32);
  } else {
    gload(R1, 32);
    gload(R0, 64);
  }
  __syncthreads();
  rfrag(FA, lds, 0);
  // one k-tile: LDS stage `cur` is multiplied; tile t+1 goes from `rw` to stage `nxt`, then tile t+3 is fetched into `rw`
#ifndef D16_ONEFRAG
#define D16_ONEFRAG 0  // 1 (with ONESET): ONE fragment set, read where it is used — 24 registers less, for 4 waves per SIMD (A/B)
#endif
  auto ktile = [&](int t, const char* cur, char* nxt, Raw& rw) {
    lwrite(rw, nxt);               // tile t+1 (the last readers of `nxt` passed the barrier of iteration t-1)
    gload(rw, (t + (ONESET ? 2 : 3)) * 32);  // past the end of K: out of range, zero fill, no traffic
    if (D16_ONEFRAG && ONESET) {
      rfrag(FA, cur, 0);
      mfmas(FA);
      rfrag(FA, cur, 1);
      mfmas(FA);
      __syncthreads();
      return;
    }
    rfrag(FB, cur, 1);
    mfmas(FA);
    D16_INTERLEAVE(2)
    __syncthreads();               // tile t+1 is complete in `nxt`; every wave has its fragments of `cur`
    rfrag(FA, nxt, 0);
    mfmas(FB);
    D16_INTERLEAVE(1)
  };
  for (int t = 0; t < nt; t += 2) {  // an odd nt runs one k-tile of zeros (zero-filled loads)
    ktile(t, lds, lds + STAGE, ONESET ? R0 : R1);
    ktile(t + 1, lds + STAGE, lds, R0);
  }

  const unsigned long long tk1 = (g.dbg & 16) ? __builtin_readcyclecounter() : 0ull;
  if (g.dbg & 8) {  // tools/dense16_probe.py: the main loop alone
    if (accm[0][0][0] == 123.456f) g.rows_part[0] = accc[0][0][0];
    return;
  }
  // ---- epilogue: 64 rows at a time through LDS ([64][BN + 4] floats), a thread then owns 8 consecutive columns of a row ----
  constexpr int LDT = BN + 4;
  float* tile = reinterpret_cast<float*>(lds);
  float* csred = reinterpret_cast<float*>(lds) + 64 * LDT;  // [16][BN] column-sum partials (behind the tile)
  const float inv_sa = mvk::f16_inv_scale(mvk::f16_scale_of(*g.a_bound));
  const int cg = tid & 15, rl = tid >> 4;
  const int n = n0 + cg * 8;
  const bool nok = n < g.N;  // N % 8 == 0
  float cinv[8], bia[8], csum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    cinv[e] = nok ? g.b_inv[n + e] * inv_sa : 0.f;
    bia[e] = (EPI == D16_NLL && nok && g.bias) ? g.bias[n + e] : 0.f;
    csum[e] = 0.f;
  }
  float gs = 1.f;
  if (EPI == D16_NLL) {
    const float gb = g.gw * g.inv_s2 * 0.25f * (1.f + *g.x_amax) * 1.0001f;
    gs = mvk::f16_scale_of(gb);
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *g.g_bound = gb;
  }
  // the side inputs of all 4 TM passes (target rows / the ReLU mask) are fetched up front: a load issued inside a pass would
  // be waited for there, 8 dependent memory latencies per thread (measured: 21 k cycles of epilogue, ~5 k of arithmetic)
  const int nc = nok ? n : 0;
  u32x4 side[TM][4][2];
#pragma unroll
  for (int h = 0; h < TM; ++h)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int lrow = rl + 16 * p;
      int m = m0 + (lrow >> 5) * (BM / 2) + h * 32 + (lrow & 31);
      m = m < g.M ? m : g.M - 1;
      if (EPI == D16_NLL) {
        const u32x4* xr = reinterpret_cast<const u32x4*>(g.X + (long long)(m % g.xrows) * g.N + nc);
        side[h][p][0] = xr[0];
        side[h][p][1] = xr[1];
      } else if (g.mask_hi) {
        side[h][p][0] = *reinterpret_cast<const u32x4*>(g.mask_hi + (long long)m * g.N + nc);
      }
    }
#pragma unroll
  for (int h = 0; h < TM; ++h) {
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int R = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        tile[(wm * 32 + R) * LDT + wn * 64 + b * 32 + l31] = fmaf(accc[h][b][r], 1.f / 2048.f, accm[h][b][r]);
      }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int lrow = rl + 16 * p;
      const int m = m0 + (lrow >> 5) * (BM / 2) + h * 32 + (lrow & 31);
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(tile + lrow * LDT + cg * 8);
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(tile + lrow * LDT + cg * 8 + 4);
      float v[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
      const bool ok = nok && m < g.M;
      if (EPI == D16_NLL) {
        float part = 0.f, gq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xv = __uint_as_float(side[h][p][e >> 2][e & 3]);
          const float pre = fmaf(v[e], cinv[e], bia[e]);
          const float rr = d16_sigmoid(pre);
          const float d = rr - xv;
          part = fmaf(0.5f * g.inv_s2 * d, d, part);
          gq[e] = g.gw * g.inv_s2 * d * (rr * (1.f - rr));
          csum[e] += ok ? gq[e] : 0.f;
        }
        part = ok ? part : 0.f;
        unsigned hh[4], ll[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) mvk::f16_split(gq[2 * e] * gs, gq[2 * e + 1] * gs, hh[e], ll[e]);
        if (ok) {
          const long long o = (long long)m * g.N + n;
          *reinterpret_cast<u32x4*>(g.Gh + o) = u32x4{hh[0], hh[1], hh[2], hh[3]};
          *reinterpret_cast<u32x4*>(g.Gl + o) = u32x4{ll[0], ll[1], ll[2], ll[3]};
        }
        // row sum over this tile's 128 columns: the 16 threads of a row are 16 consecutive lanes
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        if (cg == 0 && m < g.M) g.rows_part[(long long)blockIdx.y * g.M + m] = part + (blockIdx.y == 0 ? g.row_const : 0.f);
      } else {
        if (g.mask_hi) {
          const u32x4 mk = side[h][p][0];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const unsigned hw = (mk[e >> 1] >> ((e & 1) * 16)) & 0xffffu;
            // ReLU'(h): h > 0  <=>  its hi plane is a positive fp16 number (sign clear, not zero)
            v[e] = (hw != 0u && (hw & 0x8000u) == 0u) ? v[e] * cinv[e] : 0.f;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= cinv[e];
        }
        if (ok) {
          const long long o = (long long)m * g.N + n;
#pragma unroll
          for (int e = 0; e < 8; ++e) csum[e] += v[e];
          *reinterpret_cast<f32x4*>(g.out + o) = f32x4{v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(g.out + o + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
      }
    }
    __syncthreads();
  }
  if ((g.dbg & 16) && g.stamps && tid == 0 && EPI == D16_NLL) {  // cycle stamps of two workgroups: [main loop, epilogue]
    const unsigned long long tk2 = __builtin_readcyclecounter();
    const int slot = (blockIdx.x == 0 && blockIdx.y == 0) ? 0 : ((blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1) ? 1 : -1);
    if (slot >= 0) {
      g.stamps[2 * slot] = (float)(tk1 - tk0);  // a buffer of their own: g_bound is ONE float of the caller's
      g.stamps[2 * slot + 1] = (float)(tk2 - tk1);
    }
  }
  if (g.colsum_part) {  // fixed order: row lanes 0..15 through LDS
#pragma unroll
    for (int e = 0; e < 8; ++e) csred[rl * BN + cg * 8 + e] = csum[e];
    __syncthreads();
    if (tid < BN && n0 + tid < g.N) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s += csred[r * BN + tid];
      g.colsum_part[(long long)blockIdx.x * g.N + n0 + tid] = s;
    }
  }
  mvk_prof_end(g.prof);
}

// ---------------------------------------------------------------------------------------------------------------------
// C[j][i] = sum_m D[m][j] H[m][i]  (weight gradient): both operands are stored with the reduction index m as the ROW, i.e.
// transposed for the matrix cores.  A stage is 32 rows m of D[:, j0 .. j0+127] and of H[:, i0 .. i0+127] (two planes each,
// 256-byte rows + 64 bytes of padding: the four rows a 16-lane group of ds_read_b64_tr_b16 touches sit 16 banks apart); a
// fragment (32 rows j x 16 k) is two transposing reads (imgwgrad_kernel's address recipe).  blockIdx.z splits m; the
// partial results go to slabs, added in a fixed order by the caller's finish.
// ---------------------------------------------------------------------------------------------------------------------
struct D16Tn {
  const half_t *Dh, *Dl;  // [M][N]
  const half_t *Hh, *Hl;  // [M][K]
  const float* d_bound;
  const float* h_bound;
  int M, N, K, mchunks;   // mchunks: 32-row chunks per z slice
  int jt, it, S;          // column tiles of D / of H, z slices: the grid is 1-D (jt * it * S workgroups, XCD-aware mapping)
  float* slab;            // [gridDim.z][N][K]
  // bias gradient: workgroups (jt, 0, 0) add the forward's column-sum partials in order
  const float* colsum_part;  // [cs_rows][N] or null
  int cs_rows;
  float* db;              // [N], accumulated (+=)
  mvk_prof_slot* prof;
};

__device__ __forceinline__ f16x8 d16_tr_pair(const char* p0, const char* p1) {
  typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) h4* lp;
  const f16x4 lo = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(p0)));
  const f16x4 hi = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(p1)));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(256, 1) void d16_tn_kernel(const D16Tn g) {
  constexpr int BJ = 128, BI = 128, ROWB = 256 + 64, PL = 32 * ROWB, STAGE = 4 * PL;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  mvk_prof_begin(g.prof);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int wj = wave >> 1, wi = wave & 1;
  // Workgroup -> (tile, slice).  Consecutive workgroup ids go to the 8 XCDs round-robin, each with its own 4 MB L2: with
  // S % 8 == 0 the slice index is the XCD index, so every tile of a slice (the same 32-row chunks of D and H: ~3 MB at
  // S = 8) meets its operands in ONE L2.  (grid (jt, it, S): 146 MB of fabric reads for 26.5 MB of operands, r04_pmc_hbm.md.)
  const int tiles = g.jt * g.it, L = blockIdx.x;
  int tile, bz;
  if (g.S % 8 == 0) {
    const int q = L >> 3;
    tile = q % tiles;
    bz = (q / tiles) * 8 + (L & 7);
  } else {
    tile = L % tiles;
    bz = L / tiles;
  }
  const int bx = tile % g.jt, by = tile / g.jt;
  const int j0 = bx * BJ, i0 = by * BI;
  const int mbeg = bz * g.mchunks * 32;
  int nt = (g.M - mbeg + 31) / 32;
  nt = nt < g.mchunks ? nt : g.mchunks;
  if (nt < 0) nt = 0;

  const __amdgpu_buffer_rsrc_t rsDh = __builtin_amdgcn_make_buffer_rsrc((void*)g.Dh, 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsDl = __builtin_amdgcn_make_buffer_rsrc((void*)g.Dl, 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsHh = __builtin_amdgcn_make_buffer_rsrc((void*)g.Hh, 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsHl = __builtin_amdgcn_make_buffer_rsrc((void*)g.Hl, 0, 0x7ffffff0, 0x00020000);
  // staging: chunk idx = tid + 256 u -> row idx >> 4 (32 rows), column octet idx & 15
  int d_g[2], h_g[2], s_l[2];
  bool d_ok[2], h_ok[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int idx = tid + u * 256, mr = idx >> 4, oct = idx & 15;
    s_l[u] = mr * ROWB + oct * 16;
    d_ok[u] = j0 + oct * 8 < g.N;
    h_ok[u] = i0 + oct * 8 < g.K;
    d_g[u] = ((mbeg + mr) * g.N + j0 + oct * 8) * 2;
    h_g[u] = ((mbeg + mr) * g.K + i0 + oct * 8) * 2;
  }
  struct Raw {  // two register sets: loads run two k-tiles ahead of their LDS write (d16_nt_kernel)
    u32x4 dh[2], dl[2], hh[2], hl[2];
  };
  Raw R0, R1;
  auto sel = [](bool ok, int off) { return __builtin_unpredictable(ok) ? off : 0x7fffffff; };
  auto gload = [&](Raw& r, int t) {
    const int mrow = mbeg + t * 32;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const bool rok = t < nt && mrow + ((tid + u * 256) >> 4) < g.M;
      const int od = sel(rok && d_ok[u], d_g[u] + t * 32 * g.N * 2), oh = sel(rok && h_ok[u], h_g[u] + t * 32 * g.K * 2);
      r.dh[u] = __builtin_amdgcn_raw_buffer_load_b128(rsDh, od, 0, 0);
      r.dl[u] = __builtin_amdgcn_raw_buffer_load_b128(rsDl, od, 0, 0);
      r.hh[u] = __builtin_amdgcn_raw_buffer_load_b128(rsHh, oh, 0, 0);
      r.hl[u] = __builtin_amdgcn_raw_buffer_load_b128(rsHl, oh, 0, 0);
    }
  };
  auto lwrite = [&](const Raw& r, char* st) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      *reinterpret_cast<u32x4*>(st + s_l[u]) = r.dh[u];
      *reinterpret_cast<u32x4*>(st + PL + s_l[u]) = r.dl[u];
      *reinterpret_cast<u32x4*>(st + 2 * PL + s_l[u]) = r.hh[u];
      *reinterpret_cast<u32x4*>(st + 3 * PL + s_l[u]) = r.hl[u];
    }
  };
  // transposing-read addresses: 16-lane group gq reads [4 rows m][16 columns]; lane lp supplies row (lp >> 2), columns
  // 4 (lp & 3) .. +3 and receives column lp of the block, rows 0..3
  const int gq = lane >> 4, lp = lane & 15;
  const int kb = 8 * (gq >> 1) + (lp >> 2), cch = 16 * (gq & 1) + 4 * (lp & 3);
  int fa[2][2], fb[2][2];  // [tile][ks]: byte offset of the first read (the second: + 4 rows)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      fa[a][ks] = (ks * 16 + kb) * ROWB + (wj * 64 + a * 32 + cch) * 2;
      fb[a][ks] = 2 * PL + (ks * 16 + kb) * ROWB + (wi * 64 + a * 32 + cch) * 2;
    }
  f32x16 accm[2][2], accc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) accm[a][b][r] = accc[a][b][r] = 0.f;

  // fragments of one 16-wide k-step: D hi [2], D lo [2], H hi [2], H lo [2]
  auto rfrag = [&](f16x8 (&F)[8], const char* st, int ks) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      F[a] = d16_tr_pair(st + fa[a][ks], st + fa[a][ks] + 4 * ROWB);
      F[2 + a] = d16_tr_pair(st + PL + fa[a][ks], st + PL + fa[a][ks] + 4 * ROWB);
      F[4 + a] = d16_tr_pair(st + fb[a][ks], st + fb[a][ks] + 4 * ROWB);
      F[6 + a] = d16_tr_pair(st + PL + fb[a][ks], st + PL + fb[a][ks] + 4 * ROWB);
    }
  };
  auto mfmas = [&](const f16x8 (&F)[8]) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) accm[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[a], F[4 + b], accm[a][b], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) accc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[a], F[6 + b], accc[a][b], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) accc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[2 + a], F[4 + b], accc[a][b], 0, 0, 0);
  };
#define D16_TN_INTERLEAVE(OTHERS)                           \
  _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) {       \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      \
    __builtin_amdgcn_sched_group_barrier(0x496, OTHERS, 0); \
  }
  f16x8 FA[8], FB[8];
  gload(R0, 0);
  lwrite(R0, lds);
  gload(R1, 1);
  gload(R0, 2);
  __syncthreads();
  rfrag(FA, lds, 0);
  auto ktile = [&](int t, const char* cur, char* nxt, Raw& rw) {  // the pipeline of d16_nt_kernel
    lwrite(rw, nxt);
    gload(rw, t + 3);
    rfrag(FB, cur, 1);
    mfmas(FA);
    D16_TN_INTERLEAVE(3)
    __syncthreads();
    rfrag(FA, nxt, 0);
    mfmas(FB);
    D16_TN_INTERLEAVE(2)
  };
  for (int t = 0; t < nt; t += 2) {  // an odd nt runs one k-tile of zeros
    ktile(t, lds, lds + STAGE, R1);
    ktile(t + 1, lds + STAGE, lds, R0);
  }
  const float inv = mvk::f16_inv_scale(mvk::f16_scale_of(*g.d_bound)) * mvk::f16_inv_scale(mvk::f16_scale_of(*g.h_bound));
  float* slab = g.slab + (long long)bz * g.N * g.K;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int i = i0 + wi * 64 + b * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = j0 + wj * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (j < g.N && i < g.K) slab[(long long)j * g.K + i] = fmaf(accc[a][b][r], 1.f / 2048.f, accm[a][b][r]) * inv;
      }
    }
  if (g.colsum_part && by == 0 && bz == 0 && tid < BJ && j0 + tid < g.N) {
    float s = 0.f;
    for (int r = 0; r < g.cs_rows; ++r) s += g.colsum_part[(long long)r * g.N + j0 + tid];
    g.db[j0 + tid] += s;
  }
  mvk_prof_end(g.prof);
}

// out[i] += sum_z part[z * stride + i], z in order (the finish of the slabs when no deferred-finish arena takes them)
__global__ __launch_bounds__(256) void d16_sum_kernel(float* out, const float* part, long long count, int nz, long long stride) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  float s = 0.f;
  for (int z = 0; z < nz; ++z) s += part[z * stride + i];
  out[i] += s;
}

// planes -> fp32 (tests, and the general backward path: a row factor that is not the one folded into the planes)
__global__ __launch_bounds__(256) void d16_unsplit_kernel(const half_t* hi, const half_t* lo, const float* bound, const float* rowf,
                                                          float rowf_scale, int rowf_tile, int M, int N, float* out) {
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= (long long)M * N) return;
  const float inv = mvk::f16_inv_scale(mvk::f16_scale_of(*bound));
  const long long row = i4 / N;  // N % 4 == 0: the four elements share a row (and, rowf_tile % 4 == 0, a column tile)
  const float rf = rowf ? rowf[(rowf_tile > 0 ? (long long)((i4 - row * N) / rowf_tile) * M : 0) + row] * rowf_scale : 1.f;
  const u32x2 h = *reinterpret_cast<const u32x2*>(hi + i4), l = *reinterpret_cast<const u32x2*>(lo + i4);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned short hb = (unsigned short)(h[e >> 1] >> ((e & 1) * 16)), lb = (unsigned short)(l[e >> 1] >> ((e & 1) * 16));
    const float hv = (float)__builtin_bit_cast(half_t, hb), lv = (float)__builtin_bit_cast(half_t, lb);
    o[e] = fmaf(lv, 1.f / 2048.f, hv) * inv * rf;
  }
  *reinterpret_cast<f32x4*>(out + i4) = o;
}

template <typename Kern>
int set_lds(Kern k, int bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess
             ? MVK_OK
             : MVK_ELAUNCH;
}

}  // namespace

static int g_d16_dbg = 0;
static float* g_d16_stamps = nullptr;

extern "C" {

void mvk_dense16_debug(int flags) { g_d16_dbg = flags; }
void mvk_dense16_debug_stamps(float* four_floats) { g_d16_stamps = four_floats; }

int mvk_dense16_ok(int M, int N, int K) {
  // planes are addressed with 32-bit byte offsets; rows are cut into 16-byte pieces
  return M > 0 && N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0 && (long long)M * N < (1ll << 29) && (long long)M * K < (1ll << 29) &&
                 (long long)N * K < (1ll << 29)
             ? 1
             : 0;
}

int mvk_dense16_pack(const float* W, int N, int K, void* nk_hi, void* nk_lo, float* nk_inv, void* kn_hi, void* kn_lo,
                     float* kn_inv, void* stream) {
  if (!W || !nk_hi || !nk_lo || !nk_inv || !kn_hi || !kn_lo || !kn_inv || N <= 0 || K <= 0 || K % 16 != 0 || N % 8 != 0 ||
      !mvk_aligned16(W) || !mvk_aligned16(nk_hi) || !mvk_aligned16(nk_lo))
    return MVK_EINVAL;
  D16Pack a{W, N, K, (N + 3) / 4, (half_t*)nk_hi, (half_t*)nk_lo, nk_inv, (half_t*)kn_hi, (half_t*)kn_lo, kn_inv};
  hipLaunchKernelGGL(d16_pack_kernel, dim3(a.nk_blocks + K / 16), dim3(256), 0, mvk_stream(stream), a);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_dense16_first(const float* Z, const float* W, const float* bias, const float* z_amax, void* hi, void* lo, float* bound,
                      int M, int N, int K, int act, void* stream) {
  if (!Z || !W || !z_amax || !hi || !lo || !bound || M <= 0 || K < 4 || K > 32 || K % 4 != 0 || N < 4 || N > 1024 || N % 8 != 0 ||
      256 % (N / 4) != 0 || !mvk_aligned16(W) || !mvk_aligned16(hi) || !mvk_aligned16(lo) || (bias && !mvk_aligned16(bias)))
    return MVK_EINVAL;
  D16First a{Z, W, bias, z_amax, (half_t*)hi, (half_t*)lo, bound, M, N, K, act};
  const int want = (M + 255) / 256;
  const int R = want <= 8 ? 8 : (want <= 20 ? 20 : 40);
  const dim3 grid((M + R - 1) / R);
  hipStream_t s = mvk_stream(stream);
#define D16_FIRST_CASE(K4_)                                                                      \
  case K4_:                                                                                      \
    if (R == 8) hipLaunchKernelGGL((d16_first_kernel<K4_, 8>), grid, dim3(256), 0, s, a);        \
    else if (R == 20) hipLaunchKernelGGL((d16_first_kernel<K4_, 20>), grid, dim3(256), 0, s, a); \
    else hipLaunchKernelGGL((d16_first_kernel<K4_, 40>), grid, dim3(256), 0, s, a);              \
    break;
  switch (K / 4) {
    D16_FIRST_CASE(1) D16_FIRST_CASE(2) D16_FIRST_CASE(3) D16_FIRST_CASE(4)
    D16_FIRST_CASE(5) D16_FIRST_CASE(6) D16_FIRST_CASE(7) D16_FIRST_CASE(8)
  }
#undef D16_FIRST_CASE
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

// row tile of the forward launch: 64 rows (two workgroups per CU overlap each other's epilogue) or 128 (MVK_D16_FWD_BM)
static int d16_fwd_bm() {
  static const int bm = mvk_tune("MVK_D16_FWD_BM") ? atoi(mvk_tune("MVK_D16_FWD_BM")) : 64;
  return bm == 128 ? 128 : 64;
}
int mvk_dense16_fwd_nll_rows(int N) { return (N + 127) / 128; }
int mvk_dense16_colsum_rows(int M) { return (M + d16_fwd_bm() - 1) / d16_fwd_bm(); }

int mvk_dense16_fwd_nll(const void* h_hi, const void* h_lo, const float* h_bound, const void* w_hi, const void* w_lo,
                        const float* w_inv, const float* bias, const float* X, int xrows, const float* x_amax, float scale,
                        float grad_weight, void* g_hi, void* g_lo, float* g_bound, float* rows_part, float* colsum_part, int M,
                        int N, int K, void* stream) {
  if (!h_hi || !h_lo || !h_bound || !w_hi || !w_lo || !w_inv || !X || !x_amax || !g_hi || !g_lo || !g_bound || !rows_part ||
      xrows <= 0 || !(scale > 0.f) || !mvk_dense16_ok(M, N, K) || !mvk_aligned16(X) || !mvk_aligned16(g_hi) || !mvk_aligned16(g_lo))
    return MVK_EINVAL;
  constexpr int LDS128 = 2 * 128 * (128 + 128), LDS64 = 2 * 128 * (64 + 128);
  static bool attr = false;
  if (!attr) {
    if (set_lds(d16_nt_kernel<128, D16_NLL>, LDS128) != MVK_OK || set_lds(d16_nt_kernel<64, D16_NLL>, LDS64) != MVK_OK) return MVK_ELAUNCH;
    attr = true;
  }
  const int BM = d16_fwd_bm();
  D16Nt a{};
  a.Ah = (const half_t*)h_hi, a.Al = (const half_t*)h_lo, a.Bh = (const half_t*)w_hi, a.Bl = (const half_t*)w_lo;
  a.a_bound = h_bound, a.b_inv = w_inv, a.M = M, a.N = N, a.K = K;
  a.bias = bias, a.X = X, a.x_amax = x_amax, a.xrows = xrows;
  a.inv_s2 = 1.f / (scale * scale), a.gw = grad_weight;
  a.row_const = (float)N * (logf(scale) + 0.918938533204672742f);
  a.Gh = (half_t*)g_hi, a.Gl = (half_t*)g_lo, a.g_bound = g_bound, a.rows_part = rows_part, a.colsum_part = colsum_part;
  a.dbg = g_d16_dbg;
  a.stamps = g_d16_stamps;
  // profiler kind 9: the planes in (activation, weight), the targets, the gradient planes and the partial rows out
  a.prof = mvk::prof_next(9, 4.0 * M * K + 4.0 * N * K + 4.0 * xrows * N + 4.0 * M * N + 4.0 * M * ((N + 127) / 128));
  if (BM == 64)
    hipLaunchKernelGGL((d16_nt_kernel<64, D16_NLL>), dim3((M + 63) / 64, (N + 127) / 128), dim3(256), LDS64, mvk_stream(stream), a);
  else
    hipLaunchKernelGGL((d16_nt_kernel<128, D16_NLL>), dim3((M + 127) / 128, (N + 127) / 128), dim3(256), LDS128, mvk_stream(stream), a);
  MVK_CHECK_LAUNCH();
  mvk::prof_fold(a.prof, mvk_stream(stream));
  return MVK_OK;
}

/* dA[M][N] = (G[M][K] W^T planes [N][K]) * relu'(mask) ; db (optional) += column sums of dA */
int mvk_dense16_bwd_data(const void* g_hi, const void* g_lo, const float* g_bound, const void* wt_hi, const void* wt_lo,
                         const float* wt_inv, const void* mask_hi, float* dA, float* db, float* ws, int64_t ws_floats, int M, int N,
                         int K, void* stream) {
  if (!g_hi || !g_lo || !g_bound || !wt_hi || !wt_lo || !wt_inv || !dA || !mvk_dense16_ok(M, N, K) || !mvk_aligned16(dA))
    return MVK_EINVAL;
  hipStream_t s = mvk_stream(stream);
  constexpr int BM = 64;
  constexpr int LDS = 2 * 128 * (BM + 128) > (64 * 132 + 16 * 128) * 4 ? 2 * 128 * (BM + 128) : (64 * 132 + 16 * 128) * 4;
  static bool attr = false;
  if (!attr) {
    if (set_lds(d16_nt_kernel<BM, D16_BWD>, LDS) != MVK_OK) return MVK_ELAUNCH;
    attr = true;
  }
  const int mt = (M + BM - 1) / BM;
  float* part = nullptr;
  bool deferred = false;
  if (db) {
    part = mvk::defer_scratch(db, (long long)mt * N, s);
    deferred = part != nullptr;
    if (!part) {
      if (!ws || ws_floats < (long long)mt * N) return MVK_EINVAL;
      part = ws;
    }
  }
  D16Nt a{};
  a.Ah = (const half_t*)g_hi, a.Al = (const half_t*)g_lo, a.Bh = (const half_t*)wt_hi, a.Bl = (const half_t*)wt_lo;
  a.a_bound = g_bound, a.b_inv = wt_inv, a.M = M, a.N = N, a.K = K;
  a.mask_hi = (const half_t*)mask_hi, a.out = dA, a.colsum_part = part;
  a.dbg = g_d16_dbg;
  a.stamps = g_d16_stamps;
  a.prof = mvk::prof_next(10, 2.0 * M * N * K);  // kind 10: GEMM FLOP of the dense16 backward launches
  hipLaunchKernelGGL((d16_nt_kernel<BM, D16_BWD>), dim3(mt, (N + 127) / 128), dim3(256), LDS, s, a);
  MVK_CHECK_LAUNCH();
  mvk::prof_fold(a.prof, s);
  if (db) {
    if (deferred) return mvk::defer_push_plain(db, part, N, mt, N, s);
    hipLaunchKernelGGL(d16_sum_kernel, dim3((N + 255) / 256), dim3(256), 0, s, db, part, (long long)N, mt, (long long)N);
    MVK_CHECK_LAUNCH();
  }
  return MVK_OK;
}

/* dW[N][K] += G^T H, db[N] += sum over the colsum_part rows (the forward's partials), both in a fixed order */
int mvk_dense16_wgrad(const void* g_hi, const void* g_lo, const float* g_bound, const void* h_hi, const void* h_lo,
                      const float* h_bound, const float* colsum_part, int cs_rows, float* dW, float* db, float* ws,
                      int64_t ws_floats, int M, int N, int K, void* stream) {
  if (!g_hi || !g_lo || !g_bound || !h_hi || !h_lo || !h_bound || !dW || !mvk_dense16_ok(M, N, K) || (colsum_part && !db))
    return MVK_EINVAL;
  hipStream_t s = mvk_stream(stream);
  constexpr int LDS = 2 * 4 * 32 * (256 + 64);
  static bool attr = false;
  if (!attr) {
    if (set_lds(d16_tn_kernel, LDS) != MVK_OK) return MVK_ELAUNCH;
    attr = true;
  }
  const int jt = (N + 127) / 128, it = (K + 127) / 128;
  const int chunks = (M + 31) / 32;
  int S = 256 / (jt * it);  // ~one workgroup per compute unit ...
  S = S < 1 ? 1 : (S > chunks ? chunks : S);
  if (S >= 6 && chunks >= 8) S = S >= 12 && chunks >= 16 ? 16 : 8;  // ... in multiples of the 8 XCDs where that is close
  int per = (chunks + S - 1) / S;
  if (S % 8 != 0) S = (chunks + per - 1) / per;  // (a multiple of 8 keeps its empty tail slices: they write zero slabs)
  const long long total = (long long)N * K;
  float* slab = mvk::defer_scratch(dW, (long long)S * total, s);
  const bool deferred = slab != nullptr;
  if (!slab) {
    if (!ws || ws_floats < (long long)S * total) return MVK_EINVAL;
    slab = ws;
  }
  D16Tn a{(const half_t*)g_hi, (const half_t*)g_lo, (const half_t*)h_hi, (const half_t*)h_lo, g_bound, h_bound, M, N, K, per, jt, it, S,
          slab, colsum_part, cs_rows, db, mvk::prof_next(10, 2.0 * M * N * K)};
  hipLaunchKernelGGL(d16_tn_kernel, dim3(jt * it * S), dim3(256), LDS, s, a);
  MVK_CHECK_LAUNCH();
  mvk::prof_fold(a.prof, s);
  if (deferred) return mvk::defer_push_plain(dW, slab, total, S, total, s);
  hipLaunchKernelGGL(d16_sum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dW, slab, total, S, total);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_dense16_unsplit(const void* hi, const void* lo, const float* bound, const float* rowf, float rowf_scale, int rowf_tile,
                        int M, int N, float* out, void* stream) {
  if (!hi || !lo || !bound || !out || M <= 0 || N <= 0 || N % 4 != 0 || rowf_tile < 0 || rowf_tile % 4 != 0 || !mvk_aligned16(out) || !mvk_aligned16(hi) || !mvk_aligned16(lo))
    return MVK_EINVAL;
  const long long n4 = (long long)M * N / 4;
  hipLaunchKernelGGL(d16_unsplit_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, mvk_stream(stream), (const half_t*)hi,
                     (const half_t*)lo, bound, rowf, rowf_scale, rowf_tile, M, N, out);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

}  // extern "C"
