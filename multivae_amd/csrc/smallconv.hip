// The image-producing transposed convolution (ConvTranspose2d(Cv, Cu<=4, 4, 2, 1) + Sigmoid, svhn.py:58-60) and
// its whole backward as two per-image MFMA kernels.
//
// With only Cu = 3 output channels an output-stationary GEMM wastes 13/16 of every MFMA tile.  Instead each
// workgroup takes one image and computes the *column* matrix cols[pos][(cu,kh,kw)] = V[pos][:] . W[:, (cu,kh,kw)]
// (M = h*w positions, N = 16*Cu, K = Cv: every multiply is useful), parks it in LDS and gathers the 2x2 taps of
// every output pixel from there (col2im without atomics).  The backward kernel stages the pre-activation
// gradient tile once (sigmoid' applied while loading, zero halo for the padding) and runs both GEMMs that need
// it — backward-data (im2col gather from LDS) and backward-weight (reduction over positions, accumulated in
// registers across a persistent loop over images) — plus the bias gradient, so dU / U / V are read from HBM
// exactly once.  v_mfma_f32_16x16x4_f32: exact fp32.
#include <cstdlib>

#include "bf3.hpp"
#include "common.hpp"

#ifndef MVK_SMALL_FWD_THREADS
// Threads per workgroup (LDS allows 2 workgroups per CU either way).  Measured inside the MoPoE step (B=512, 3x32 ch):
// forward 89 us at 256 threads -> 67 us at 512 -> 62 us at 1024 (more loads in flight, 56 VGPRs); backward 152 us at 256 -> 188 us at 512 (the
// per-wave weight-gradient accumulators are replicated over twice the waves), so the two kernels differ.
#define MVK_SMALL_FWD_THREADS 1024
#endif
#ifndef MVK_SMALL_DOWN_THREADS
#define MVK_SMALL_DOWN_THREADS 512
#endif
#ifndef MVK_SMALL_BWD_THREADS
#define MVK_SMALL_BWD_THREADS 256
#endif

#ifdef MVK_SUPROF
__device__ unsigned long long* g_su_dbg = nullptr;
extern "C" int mvk_smallup_debug_buffer(unsigned long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_su_dbg), &p, sizeof(p)) == hipSuccess ? 0 : -2;
}
#endif

#ifndef MVK_SUF_ABL
#define MVK_SUF_ABL 0  // subtraction builds of small_up_fwd_bf_kernel: 1 no piece split, 2 one MFMA per tile instead of six, 4 a quarter of the column-matrix writes, 8 one tap read instead of four, 16 no activation
#endif
namespace {

using mvk::f32x4;

template <int CU, int CV>
struct SmallCfg {
  static constexpr int NC = 16 * CU;   // columns (cu, kh, kw)
  static constexpr int VS = CV + 4;    // V tile row stride in LDS (16-byte aligned rows)
  static constexpr int CS = NC + 1;    // column-matrix row stride
  static constexpr int WT = CV + 16;   // transposed-weight row stride (bank shift of 16 between k rows)
  static constexpr int MT = 4;         // 16-row tiles per wave (64 positions)
  static constexpr int NTV = CV / 16;  // 16-wide tiles over Cv
};

// ---------------------------------------------------------------------------------------------------------
// forward: U[n,Cu,2h,2w] = act(convT(V[n,h,w,Cv]) + b)
// ---------------------------------------------------------------------------------------------------------
// DENSE: every per-element guard is a compile-time fact (16x16 inputs, the tile sizes divide the workgroup evenly)
template <int CU, int CV, int NT, bool DENSE>
__global__ __launch_bounds__(NT) void small_up_fwd_kernel(const float* __restrict__ V, const float* __restrict__ Wref,
                                                           const float* __restrict__ bias, float* __restrict__ U, int n,
                                                           int h, int w, int act, mvk_prof_slot* prof) {
  mvk_prof_begin(prof);
  using C = SmallCfg<CU, CV>;
  constexpr int WP = 256 / (NT / 64);  // positions per wave (NT = 256: 64, NT = 512: 32)
  constexpr int MT = WP / 16;          // 16-row MFMA tiles per wave
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                 // [CV][NC]  == Wref[cv][cu][tap] as is
  float* buf = smem + CV * C::NC;   // V tile [P][VS], later the column matrix [P][CS]
  const int P = h * w;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  for (int i = tid; i < CV * C::NC; i += NT) Ws[i] = Wref[i];
  constexpr int NV = 256 * CV / 4 / NT;  // float4 per thread for a full 256-position tile
  const int n4 = P * CV / 4;
  f32x4 pre[NV];  // ext-vector type: HIP's float4 struct arrays are not promoted to registers here
  auto prefetch = [&](long long img) __attribute__((always_inline)) {
    const f32x4* src = reinterpret_cast<const f32x4*>(V + img * P * CV);
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int idx = tid + u * NT;
      pre[u] = src[(DENSE || idx < n4) ? idx : n4 - 1];  // clamped: unconditional loads keep pre[] in registers
    }
  };
  const bool active = DENSE || wave * WP < P;
  const int H2 = 2 * h, W2 = 2 * w;
  const int per_img = CU * H2 * W2;
  // Output geometry is the same for every image: this thread's outputs o = tid + 256 t and, for each, the LDS
  // word of its (up to) 2x2 contributing column-matrix entries.  Missing taps (image border) point at a zero word.
  constexpr int NO = (CU * 32 * 32 + NT - 1) / NT;  // h, w <= 16
  const int zidx = P * (C::VS > C::CS ? C::VS : C::CS);  // one float past both uses of buf, kept at 0
  int tap[NO][4];
  float bia[NO];
#pragma unroll
  for (int t = 0; t < NO; ++t) {
    const int o = tid + t * NT;
    const int oc = o < per_img ? o : 0;
    const int cu = oc / (H2 * W2);
    const int rem = oc - cu * (H2 * W2);
    const int oh = rem / W2, ow = rem - oh * W2;
    const int ph = oh & 1, pw = ow & 1, i0 = oh >> 1, j0 = ow >> 1;
    bia[t] = bias ? bias[cu] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int ih = i0 + ph - a, kh = (1 - ph) + 2 * a;
        const int iw = j0 + pw - b, kw = (1 - pw) + 2 * b;
        const bool ok = ih >= 0 && ih < h && iw >= 0 && iw < w;
        tap[t][a * 2 + b] = ok ? (ih * w + iw) * C::CS + cu * 16 + kh * 4 + kw : zidx;
      }
  }
  if (tid == 0) buf[zidx] = 0.f;
  long long img = blockIdx.x;
  if (img < n) prefetch(img);
  for (; img < n; img += gridDim.x) {
    // stage this image's V tile (prefetched one iteration ago), then start fetching the next image
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int idx = tid + u * NT;
      if (DENSE || idx < n4) {
        const int pos = idx / (CV / 4), q = idx - pos * (CV / 4);
        *reinterpret_cast<f32x4*>(buf + pos * C::VS + 4 * q) = pre[u];
      }
    }
    __syncthreads();
    if (img + gridDim.x < n) prefetch(img + gridDim.x);
    f32x4 acc[MT][CU];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < CU; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (active) {
#pragma unroll 2
      for (int ks = 0; ks < CV / 4; ++ks) {
        const int k = ks * 4 + lq;
        float av[MT], bv[CU];
#pragma unroll
        for (int a = 0; a < MT; ++a) av[a] = buf[(wave * WP + a * 16 + l15) * C::VS + k];
#pragma unroll
        for (int b = 0; b < CU; ++b) bv[b] = Ws[k * C::NC + b * 16 + l15];
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int b = 0; b < CU; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
      }
    }
    __syncthreads();  // every wave is done with the V tile
    if (active) {
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < CU; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) buf[(wave * WP + a * 16 + lq * 4 + r) * C::CS + b * 16 + l15] = acc[a][b][r];
    }
    __syncthreads();
    float* out = U + img * per_img;
#pragma unroll
    for (int t = 0; t < NO; ++t) {
      const int o = tid + t * NT;
      if (DENSE || o < per_img) {
        const float sum = ((buf[tap[t][0]] + buf[tap[t][1]]) + (buf[tap[t][2]] + buf[tap[t][3]])) + bia[t];
        out[o] = mvk_act(sum, act);
      }
    }
    __syncthreads();  // the column matrix is overwritten by the next image's V tile
  }
  mvk_prof_end(prof);
}

// The same forward for 16x16 inputs with Cv = 32 on the bf16 matrix cores: the column-matrix GEMM (M = 256, N = 16 Cu, K = 32) is
// 44 % of the fp32 kernel's time in v_mfma_f32_16x16x4_f32 alone (3072 of ~7000 cycles per image and SIMD).  Here V is split into
// three bf16 pieces while it is staged (x = x0 + x1 + x2, bf3.hpp; planes [pos][32 k] of 64-byte rows, 16-byte k-octet o of row r at
// o ^ ((r >> 1) & 3): conflict-free ds_write_b64 staging and ds_read_b128 fragments), the weights once per workgroup, and a product
// is the 6 piece products of order <= 2 on v_mfma_f32_16x16x32_bf16 — ONE instruction covers the whole K: 6 x 16 cycles per
// 16 x 16 tile instead of 8 x 32, fp32-level error (same scheme and error bound as the tiled engine, igemm_bf.hpp).
// Everything around the GEMM (prefetch, column matrix in LDS, output gather, activation) is the fp32 kernel's.
// FUSED TAIL (X != nullptr; round 3): the layer is the last one of a decoder whose output is scored against the data by a Normal
// likelihood (models/base/base_utils.py:62-87 `set_decoder_dist`, mopoe_model.py:192-199).  The workgroup that holds a whole
// reconstructed image then scores it in its epilogue: rows[img] = sum_d (r_d - x_d)^2 / (2 s^2) + D (log s + 1/2 log 2 pi) (fixed
// summation order: lanes, waves) and U receives d rows[img] / d pre-activation = (r - x) / s^2 * act'(r) instead of the image
// — the reconstruction and its gradient never travel to HBM and back (3 of the 5 passes over the [K B, 3, 32, 32] tensor gone).
// X: [xrows][CU * 1024] targets, image img is scored against row img % xrows (the K samples of a data point share it).
// (NLL is a template argument: the plain kernel sits exactly at its 64-register budget of 8 waves per SIMD, and the tail's
// few extra live values made BOTH paths spill when it was a run-time branch: 47 -> 54 us for the plain forward.)
template <int CU, int NT, bool NLL = false>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT / 128, NT / 128))) void small_up_fwd_bf_kernel(const float* __restrict__ V, const float* __restrict__ Wref,
                                                              const float* __restrict__ bias, float* __restrict__ U, int n,
                                                              int act, mvk_prof_slot* prof, const float* __restrict__ X = nullptr,
                                                              int xrows = 1, float inv_s2 = 1.f, float lconst = 0.f,
                                                              float* __restrict__ rows = nullptr, float g_inv_s2 = 1.f) {
  // g_inv_s2 = grad_weight / scale^2: the stored gradient carries the weight its row enters the loss with
  mvk_prof_begin(prof);
  using mvk::bf16x8;
  using mvk::u32x2;
  constexpr int CV = 32, NC = 16 * CU, CS = NC + 1, P = 256, h = 16, w = 16;
  constexpr int WP = P / (NT / 64), MT = WP / 16;
  constexpr int PLANE = P * 64, WPLANE = NC * 64;  // bytes per piece plane
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* Wb = reinterpret_cast<char*>(smem);  // 3 x [NC][32 k] bf16
  float* buf = smem + 3 * WPLANE / 4;       // 3 x [P][32 k] bf16, later the column matrix [P][CS] (+ the zero word)
  char* Vb = reinterpret_cast<char*>(buf);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  for (int i = tid; i < NC * 16; i += NT) {  // weight pieces: column nn, k = 2 kp, 2 kp + 1
    const int nn = i >> 4, kp = i & 15;
    unsigned p0, p1, p2;
    mvk::bf3_split(Wref[(2 * kp) * NC + nn], Wref[(2 * kp + 1) * NC + nn], p0, p1, p2);
    const int off = nn * 64 + (((kp >> 2) ^ ((nn >> 1) & 3)) << 4) + (kp & 3) * 4;
    *reinterpret_cast<unsigned*>(Wb + off) = p0;
    *reinterpret_cast<unsigned*>(Wb + WPLANE + off) = p1;
    *reinterpret_cast<unsigned*>(Wb + 2 * WPLANE + off) = p2;
  }
  constexpr int NV = P * CV / 4 / NT;
  static_assert((P * CV / 4) % NT == 0 && (CU * 1024) % NT == 0 && MT >= 1, "workgroup size");
  f32x4 pre[NV];
  auto prefetch = [&](long long img) __attribute__((always_inline)) {
    const f32x4* src = reinterpret_cast<const f32x4*>(V + img * P * CV);
#pragma unroll
    for (int u = 0; u < NV; ++u) pre[u] = src[tid + u * NT];
  };
  constexpr int H2 = 2 * h, W2 = 2 * w, per_img = CU * H2 * W2;
  constexpr int NO = per_img / NT;
  constexpr int zidx = P * CS;  // one float past the column matrix (and past the piece planes), kept at 0
  static_assert(3 * PLANE <= zidx * 4, "the zero word must survive the staging");
  int tap[NO][4];
  float bia[NO];
#pragma unroll
  for (int t = 0; t < NO; ++t) {
    const int o = tid + t * NT;
    const int cu = o / (H2 * W2);
    const int rem = o - cu * (H2 * W2);
    const int oh = rem / W2, ow = rem - oh * W2;
    const int ph = oh & 1, pw = ow & 1, i0 = oh >> 1, j0 = ow >> 1;
    bia[t] = bias ? bias[cu] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int ih = i0 + ph - a, kh = (1 - ph) + 2 * a;
        const int iw = j0 + pw - b, kw = (1 - pw) + 2 * b;
        const bool ok = ih >= 0 && ih < h && iw >= 0 && iw < w;
        tap[t][a * 2 + b] = ok ? (ih * w + iw) * CS + cu * 16 + kh * 4 + kw : zidx;
      }
  }
  if (tid == 0) buf[zidx] = 0.f;
  // fragment addresses (bytes inside a plane)
  int aoff[MT], boff[CU];
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    const int r = wave * WP + a * 16 + l15;
    aoff[a] = r * 64 + ((lq ^ ((r >> 1) & 3)) << 4);
  }
#pragma unroll
  for (int b = 0; b < CU; ++b) {
    const int c = b * 16 + l15;
    boff[b] = c * 64 + ((lq ^ ((c >> 1) & 3)) << 4);
  }
  long long img = blockIdx.x;
  if (img < n) prefetch(img);
  for (; img < n; img += gridDim.x) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int idx = tid + u * NT;
      const int pos = idx >> 3, q = idx & 7;
      unsigned a0, a1, a2, b0, b1, b2;
      if (MVK_SUF_ABL & 1) {
        a0 = a1 = __float_as_uint(pre[u][0]), a2 = __float_as_uint(pre[u][1]);
        b0 = b1 = __float_as_uint(pre[u][2]), b2 = __float_as_uint(pre[u][3]);
      } else {
        mvk::bf3_split(pre[u][0], pre[u][1], a0, a1, a2);
        mvk::bf3_split(pre[u][2], pre[u][3], b0, b1, b2);
      }
      const int off = pos * 64 + (((q >> 1) ^ ((pos >> 1) & 3)) << 4) + (q & 1) * 8;
      *reinterpret_cast<u32x2*>(Vb + off) = u32x2{a0, b0};
      *reinterpret_cast<u32x2*>(Vb + PLANE + off) = u32x2{a1, b1};
      *reinterpret_cast<u32x2*>(Vb + 2 * PLANE + off) = u32x2{a2, b2};
    }
    __syncthreads();
    if (img + gridDim.x < n) prefetch(img + gridDim.x);
    float xv[NO];  // fused tail: this thread's target pixels, loaded here and first touched behind the GEMM
    if (NLL) {
      const float* xt = X + (img % xrows) * per_img;
#pragma unroll
      for (int t = 0; t < NO; ++t) xv[t] = xt[tid + t * NT];
    }
    f32x4 acc[MT][CU];
    {
      bf16x8 af[MT][3];
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int p = 0; p < 3; ++p) af[a][p] = *reinterpret_cast<const bf16x8*>(Vb + p * PLANE + aoff[a]);
      constexpr int PA[6] = {0, 1, 2, 0, 1, 0};  // smallest terms first
      constexpr int PB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
      for (int b = 0; b < CU; ++b) {  // one column tile's weight pieces at a time (64 registers per lane at 8 waves per SIMD)
        bf16x8 bfr[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) bfr[p] = *reinterpret_cast<const bf16x8*>(Wb + p * WPLANE + boff[b]);
#pragma unroll
        for (int a = 0; a < MT; ++a) {
          f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < ((MVK_SUF_ABL & 2) ? 1 : 6); ++t) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a][PA[t]], bfr[PB[t]], c, 0, 0, 0);
          acc[a][b] = c;
        }
      }
    }
    __syncthreads();  // every wave is done with the piece planes
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < CU; ++b)
#pragma unroll
        for (int r = 0; r < ((MVK_SUF_ABL & 4) ? 1 : 4); ++r) buf[(wave * WP + a * 16 + lq * 4 + r) * CS + b * 16 + l15] = acc[a][b][r];
    __syncthreads();
    float* out = U + img * per_img;
    if (NLL) {
      float part = 0.f;
#pragma unroll
      for (int t = 0; t < NO; ++t) {
        const float sum = ((buf[tap[t][0]] + buf[tap[t][1]]) + (buf[tap[t][2]] + buf[tap[t][3]])) + bia[t];
        // (the sigmoid on v_exp_f32 / v_rcp_f32 with a two-term exponent, 9 instructions instead of expf + an IEEE division)
        const float r = act == MVK_ACT_SIGMOID ? mvk_fast_sigmoid(sum) : mvk_act(sum, act), dlt = r - xv[t];
        part = fmaf(0.5f * inv_s2 * dlt, dlt, part);
        out[tid + t * NT] = dlt * g_inv_s2 * mvk_act_grad_from_out(r, act);
      }
      part = wave_sum_dpp(part);
      if (lane == 0) buf[zidx + 1 + wave] = part;
    } else {
#pragma unroll
      for (int t = 0; t < NO; ++t) {
        const float sum = (MVK_SUF_ABL & 8) ? buf[tap[t][0]] + bia[t] : ((buf[tap[t][0]] + buf[tap[t][1]]) + (buf[tap[t][2]] + buf[tap[t][3]])) + bia[t];
        out[tid + t * NT] = (MVK_SUF_ABL & 16) ? sum : mvk_act(sum, act);
      }
    }
    __syncthreads();  // the column matrix is overwritten by the next image's pieces
    if (NLL && wave == 0) {  // the wave partials are rewritten three barriers from now at the earliest
      float tot = lane < NT / 64 ? buf[zidx + 1 + lane] : 0.f;  // a fixed shuffle tree over the partials (small_up_fwd_h_kernel)
#pragma unroll
      for (int off = NT / 128; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64);
      if (lane == 0) rows[img] = tot + lconst;
    }
  }
  mvk_prof_end(prof);
}


// The same forward on SCALED fp16 PAIRS with the operand roles swapped (round 4).  Subtraction builds of the kernel above at
// n = 5120 (48.5 us alone, plain form): five of its six MFMAs per tile cost 7.7 us, three quarters of its column-matrix writes 4.4 us,
// three of its four tap reads 3.1 us, the IEEE sigmoid 5.3 us, the piece split 1.4 us; everything off: 35.8 us = 6.4 TB/s, the
// streaming skeleton.  Here:
//   * V is split into (hi, lo) fp16 pairs under the bound its producer published (v_amax, bf3.hpp), W under its own maximum
//     (computed once per workgroup): 3 MFMAs per tile instead of 6, two planes instead of three;
//   * the weights are the A operand and the positions the B operand: D[nn][pos], a lane holds FOUR CONSECUTIVE columns nn of one
//     position, i.e. one ds_write_b128 per tile into the column matrix (row stride 52 floats: conflict-free) instead of four
//     ds_write_b32;
//   * the fast sigmoid in both forms.
// Same sums otherwise (tap order, bias, NLL tail, wave-ordered row sum).
// Round 5 experiment switches of the kernel below (compile-time; tools/suh_variants.sh builds and times them; medians of the fused
// tail alone over three interleaved rounds on one box, base 58.2 us: XFIRST 56.2, PAIR 55.6, WREG 54.0 (shipped), NT 57.5,
// XFIRST + WREG 65.1 (!), XFIRST + WREG + PAIR 57.3, all four 61.1 — the combinations are worse than their parts: the kernel is
// bound by how its four barrier phases of two resident workgroups interleave, not by any one instruction stream):
//   MVK_SUH_XFIRST  the targets' loads are issued BEFORE the next image's prefetch: vmcnt counts in order, so waiting for a
//                   load issued behind the prefetch is waiting for the prefetch as well
//   MVK_SUH_PAIR    a thread owns PAIRS of horizontally adjacent pixels (8-byte target loads and gradient stores, one tap table
//                   of 8 entries instead of 24: the channel is the unrolled index)
//   MVK_SUH_WREG    the weight fragments stay in registers for the whole launch (24 VGPRs) instead of 6 ds_read_b128 per image
//   MVK_SUH_NT      nontemporal loads of the input map (read once) and stores of the gradient (read ~300 us later)
#ifndef MVK_SUH_XFIRST
#define MVK_SUH_XFIRST 0
#endif
#ifndef MVK_SUH_PAIR
#define MVK_SUH_PAIR 0
#endif
#ifndef MVK_SUH_WREG
#define MVK_SUH_WREG 1  // measured (tools/suh_run.sh, n = 5120, three interleaved rounds): fused tail 58.2 -> 54.0 us, plain 48.8 -> 45.7
#endif
#ifndef MVK_SUH_NT
#define MVK_SUH_NT 0
#endif
template <int CU, int NT, bool NLL = false>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT / 128, NT / 128))) void small_up_fwd_h_kernel(
    const float* __restrict__ V, const float* __restrict__ Wref, const float* __restrict__ bias, float* __restrict__ U, int n, int act,
    mvk_prof_slot* prof, const float* __restrict__ v_amax, const float* __restrict__ X = nullptr, int xrows = 1, float inv_s2 = 1.f,
    float lconst = 0.f, float* __restrict__ rows = nullptr, float g_inv_s2 = 1.f, float* __restrict__ du_amax = nullptr) {
  mvk_prof_begin(prof);
#ifndef MVK_SUH_GMAX
// how the fused tail bounds max |stored gradient|: 0 = not at all (A/B), 1 = exact, per thread, 2 = exact, per wave and image,
// 3 = from the largest row sum (wave 0 has it anyway): |g| <= |g_inv_s2| max act' sqrt(2 max_i tot_i / inv_s2), up to
// sqrt(3072) = 55 times the true maximum (6 bits of fp16 range, no precision), no instruction outside wave 0
#define MVK_SUH_GMAX 3
#endif
  float gmax = 0.f;       // ... of this thread (published at the end when du_amax is given)
  unsigned gmax_u = 0u;   // ... of this wave, as bits (wave-uniform: a scalar register)
  using mvk::f16x8;
  using mvk::u32x2;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int CV = 32, NC = 16 * CU, CS = NC + 4, P = 256, h = 16, w = 16;
  constexpr int WP = P / (NT / 64), MT = WP / 16;
  constexpr int PLANE = P * 64, WPLANE = NC * 64;  // bytes per piece plane
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* Wb = reinterpret_cast<char*>(smem);  // 2 x [NC][32 k] fp16
  float* buf = smem + 2 * WPLANE / 4;       // 2 x [P][32 k] fp16, later the column matrix [P][CS] (+ the zero word, wave partials)
  char* Vb = reinterpret_cast<char*>(buf);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  constexpr int zidx = P * CS;  // one float past the column matrix (and past the piece planes), kept at 0
  static_assert(2 * PLANE <= zidx * 4, "the zero word must survive the staging");
  // weight scale: max |W| over the NC * 32 weights (every thread visits its pairs twice: maximum, then split)
  float wmax = 0.f;
  for (int i = tid; i < NC * 16; i += NT) {
    const int nn = i >> 4, kp = i & 15;
    wmax = fmaxf(wmax, fmaxf(fabsf(Wref[(2 * kp) * NC + nn]), fabsf(Wref[(2 * kp + 1) * NC + nn])));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off, 64));
  if (lane == 0) buf[zidx + 1 + wave] = wmax;
  __syncthreads();
  wmax = 0.f;
  for (int wv = 0; wv < NT / 64; ++wv) wmax = fmaxf(wmax, buf[zidx + 1 + wv]);
  const float sw = mvk::f16_scale_of(wmax), sv = mvk::f16_scale_of(*v_amax);
  const float inv_s = mvk::f16_inv_scale(sw) * mvk::f16_inv_scale(sv);
  __syncthreads();
  for (int i = tid; i < NC * 16; i += NT) {  // weight pieces: column nn, k = 2 kp, 2 kp + 1
    const int nn = i >> 4, kp = i & 15;
    unsigned p0, p1;
    mvk::f16_split(Wref[(2 * kp) * NC + nn] * sw, Wref[(2 * kp + 1) * NC + nn] * sw, p0, p1);
    const int off = nn * 64 + (((kp >> 2) ^ ((nn >> 1) & 3)) << 4) + (kp & 3) * 4;
    *reinterpret_cast<unsigned*>(Wb + off) = p0;
    *reinterpret_cast<unsigned*>(Wb + WPLANE + off) = p1;
  }
  constexpr int NV = P * CV / 4 / NT;
  static_assert((P * CV / 4) % NT == 0 && (CU * 1024) % NT == 0 && MT >= 1, "workgroup size");
  f32x4 pre[NV];
  auto prefetch = [&](long long img) __attribute__((always_inline)) {
    const f32x4* src = reinterpret_cast<const f32x4*>(V + img * P * CV);
#pragma unroll
    for (int u = 0; u < NV; ++u) pre[u] = MVK_SUH_NT ? __builtin_nontemporal_load(src + tid + u * NT) : src[tid + u * NT];
  };
  constexpr int H2 = 2 * h, W2 = 2 * w, per_img = CU * H2 * W2;
  constexpr int NO = per_img / NT;
  // output -> (tap addresses in the column matrix, bias).  Plain mapping: output o = tid + t NT of the NCHW image.  PAIR mapping
  // (NT = 512): thread = (output row tid >> 4, pixel pair tid & 15), the unrolled index is the channel: o = cu 1024 + 2 tid + e.
  constexpr bool PAIR = MVK_SUH_PAIR && NT == 512 && H2 * W2 == 2 * NT;
  constexpr int NTAP = PAIR ? 2 : NO;
  int tap[NTAP][4];
  float bia[NO];
#pragma unroll
  for (int t = 0; t < NO; ++t) {
    const int cu = PAIR ? t / 2 : (tid + t * NT) / (H2 * W2);
    bia[t] = bias ? bias[cu] : 0.f;
  }
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    const int o = PAIR ? 2 * tid + t : tid + t * NT;
    const int cu = o / (H2 * W2);
    const int rem = o - cu * (H2 * W2);
    const int oh = rem / W2, ow = rem - oh * W2;
    const int ph = oh & 1, pw = ow & 1, i0 = oh >> 1, j0 = ow >> 1;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int ih = i0 + ph - a, kh = (1 - ph) + 2 * a;
        const int iw = j0 + pw - b, kw = (1 - pw) + 2 * b;
        const bool ok = ih >= 0 && ih < h && iw >= 0 && iw < w;
        tap[t][a * 2 + b] = ok ? (ih * w + iw) * CS + cu * 16 + kh * 4 + kw : zidx;
      }
  }
  if (tid < 3) buf[zidx + 16 * tid] = 0.f;
  // fragment addresses (bytes inside a plane): positions = B operand, weights = A operand (both [row][32 k], k-octet lq)
  int poff[MT], woff[CU];
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    const int r = wave * WP + a * 16 + l15;
    poff[a] = r * 64 + ((lq ^ ((r >> 1) & 3)) << 4);
  }
#pragma unroll
  for (int b = 0; b < CU; ++b) {
    const int c = b * 16 + l15;
    woff[b] = c * 64 + ((lq ^ ((c >> 1) & 3)) << 4);
  }
#if MVK_SUH_WREG
  __syncthreads();  // the weight planes are complete
  f16x8 wreg[CU][2];
#pragma unroll
  for (int b = 0; b < CU; ++b)
#pragma unroll
    for (int p = 0; p < 2; ++p) wreg[b][p] = *reinterpret_cast<const f16x8*>(Wb + p * WPLANE + woff[b]);
#endif
  long long img = blockIdx.x;
  if (img < n) prefetch(img);
  for (; img < n; img += gridDim.x) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int idx = tid + u * NT;
      const int pos = idx >> 3, q = idx & 7;
      unsigned a0, a1, b0, b1;
      mvk::f16_split(pre[u][0] * sv, pre[u][1] * sv, a0, a1);
      mvk::f16_split(pre[u][2] * sv, pre[u][3] * sv, b0, b1);
      const int off = pos * 64 + (((q >> 1) ^ ((pos >> 1) & 3)) << 4) + (q & 1) * 8;
      *reinterpret_cast<u32x2*>(Vb + off) = u32x2{a0, b0};
      *reinterpret_cast<u32x2*>(Vb + PLANE + off) = u32x2{a1, b1};
    }
    __syncthreads();
    if (!MVK_SUH_XFIRST && img + gridDim.x < n) prefetch(img + gridDim.x);
    float xv[NO];  // fused tail: this thread's target pixels, loaded here and first touched behind the GEMM
    if (NLL) {
      const float* xt = X + (img % xrows) * per_img;
      if (PAIR) {
#pragma unroll
        for (int c = 0; c < NO / 2; ++c) {
          const f32x2 v2 = reinterpret_cast<const f32x2*>(xt + c * H2 * W2)[tid];
          xv[2 * c] = v2[0], xv[2 * c + 1] = v2[1];
        }
      } else {
#pragma unroll
        for (int t = 0; t < NO; ++t) xv[t] = (MVK_SUF_ABL & 32) ? 0.25f : xt[tid + t * NT];
      }
    }
    if (MVK_SUH_XFIRST && img + gridDim.x < n) prefetch(img + gridDim.x);
    f32x4 res[MT][CU];  // D[nn = b * 16 + 4 lq + r][pos = wave * WP + a * 16 + l15]
    {
      f16x8 pf[MT][2];
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int p = 0; p < 2; ++p) pf[a][p] = *reinterpret_cast<const f16x8*>(Vb + p * PLANE + poff[a]);
#pragma unroll
      for (int b = 0; b < CU; ++b) {
        f16x8 wf[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#if MVK_SUH_WREG
          wf[p] = wreg[b][p];
#else
          wf[p] = *reinterpret_cast<const f16x8*>(Wb + p * WPLANE + woff[b]);
#endif
        }
#pragma unroll
        for (int a = 0; a < MT; ++a) {
          f32x4 cm = f32x4{0.f, 0.f, 0.f, 0.f}, cx = f32x4{0.f, 0.f, 0.f, 0.f};
          cx = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0], pf[a][1], cx, 0, 0, 0);
          cm = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0], pf[a][0], cm, 0, 0, 0);
          cx = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1], pf[a][0], cx, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) res[a][b][r] = fmaf(cx[r], 1.f / 2048.f, cm[r]) * inv_s;
        }
      }
    }
    __syncthreads();  // every wave is done with the piece planes
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < CU; ++b)
        *reinterpret_cast<f32x4*>(buf + (wave * WP + a * 16 + l15) * CS + b * 16 + 4 * lq) = res[a][b];
    __syncthreads();
    float* out = U + img * per_img;
    // output t of this thread: its four taps (fixed order), bias, activation
    auto pixel = [&](int t) __attribute__((always_inline)) {
      const int tt = PAIR ? (t & 1) : t, co = PAIR ? (t >> 1) * 16 : 0;  // PAIR: the channel is a column offset of the same taps
      // (the zero word has a copy at zidx + 16 and zidx + 32: a border tap plus a channel offset still reads a zero)
      const float sum = ((buf[tap[tt][0] + co] + buf[tap[tt][1] + co]) + (buf[tap[tt][2] + co] + buf[tap[tt][3] + co])) + bia[t];
      return act == MVK_ACT_SIGMOID ? mvk_fast_sigmoid(sum) : mvk_act(sum, act);
    };
    if (NLL) {
      float part = 0.f;
      float gv[NO];
#pragma unroll
      for (int t = 0; t < NO; ++t) {
        const float r = pixel(t), dlt = r - xv[t];
        if (!(MVK_SUF_ABL & 64)) part = fmaf(0.5f * inv_s2 * dlt, dlt, part);
        gv[t] = dlt * g_inv_s2 * mvk_act_grad_from_out(r, act);
      }
      if (MVK_SUH_GMAX == 1) {
#pragma unroll
        for (int t = 0; t < NO; ++t) gmax = fmaxf(gmax, fabsf(gv[t]));
      } else if (MVK_SUH_GMAX == 2) {
        float gm = fabsf(gv[0]);
#pragma unroll
        for (int t = 1; t < NO; ++t) gm = fmaxf(gm, fabsf(gv[t]));
        const unsigned gu = wave_max_dpp_bits(gm);
        gmax_u = gu > gmax_u ? gu : gmax_u;
      }
      if (PAIR) {
#pragma unroll
        for (int c = 0; c < NO / 2; ++c) {
          f32x2* dst = reinterpret_cast<f32x2*>(out + c * H2 * W2) + tid;
          if (MVK_SUH_NT) __builtin_nontemporal_store(f32x2{gv[2 * c], gv[2 * c + 1]}, dst);
          else *dst = f32x2{gv[2 * c], gv[2 * c + 1]};
        }
      } else {
#pragma unroll
        for (int t = 0; t < NO; ++t) {
          if (MVK_SUH_NT) __builtin_nontemporal_store(gv[t], out + tid + t * NT);
          else out[tid + t * NT] = gv[t];
        }
      }
      if (!(MVK_SUF_ABL & 64)) {
        part = wave_sum_dpp(part);
        if (lane == 0) buf[zidx + 1 + wave] = part;
      }
    } else if (PAIR) {
#pragma unroll
      for (int c = 0; c < NO / 2; ++c) reinterpret_cast<f32x2*>(out + c * H2 * W2)[tid] = f32x2{pixel(2 * c), pixel(2 * c + 1)};
    } else {
#pragma unroll
      for (int t = 0; t < NO; ++t) out[tid + t * NT] = pixel(t);
    }
    __syncthreads();  // the column matrix is overwritten by the next image's pieces
    if (NLL && wave == 0 && !(MVK_SUF_ABL & 64)) {  // the wave partials are rewritten three barriers from now at the earliest
      // (one LDS latency + a fixed shuffle tree over the NT / 64 partials: a serial loop of dependent LDS reads in thread 0 held
      // its wave back ~3 us per launch at the next barrier)
      unsigned ones = ~0u;
      asm volatile("" : "+s"(ones));  // the lane index recomputed HERE (mbcnt of an opaque mask): kept live across the image loop it
      const int ln = __builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));  // is the register hipcc spills
      float tot = ln < NT / 64 ? buf[zidx + 1 + ln] : 0.f;
#pragma unroll
      for (int off = NT / 128; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64);
      if (lane == 0) rows[img] = tot + lconst;
      if (MVK_SUH_GMAX == 3) {
        const unsigned tb = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(tot));  // a sum of squares: >= 0
        gmax_u = tb > gmax_u ? tb : gmax_u;
      }
    }
  }
  if (MVK_SUH_GMAX == 3) {
    if (NLL && du_amax && tid == 0) {  // one atomic per workgroup, only when it raises the published value (bf3.hpp)
      const float cmax = act == MVK_ACT_SIGMOID ? 0.25f : 1.f;
      const float bound = 1.001f * fabsf(g_inv_s2) * cmax * sqrtf(2.f * __uint_as_float(gmax_u) / inv_s2);
      unsigned* const d = reinterpret_cast<unsigned*>(du_amax);
      const unsigned mu = __float_as_uint(bound);
      if (mu > __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(d, mu);
    }
  } else if (NLL && du_amax) mvk::amax_publish_wave(MVK_SUH_GMAX == 2 ? __uint_as_float(gmax_u) : __uint_as_float(wave_max_dpp_bits(gmax)), du_amax, buf + zidx + 1);  // uniform branch; the wave-partial words are free here
  mvk_prof_end(prof);
}

// ---------------------------------------------------------------------------------------------------------
// backward: dV = down(dUpre) * act'(V),  partial dW / db per workgroup (persistent over images)
// ---------------------------------------------------------------------------------------------------------
// A work unit is an image (PU = 256 positions) or, for 16x16 inputs, half an image (PU = 128: rows [8 half, 8 half + 8);
// the gradient tile then carries the neighbouring half's two rows as halo instead of zeros).  Half units need 35 KB of LDS
// and ~120 registers instead of 61 KB and 252: four workgroups per CU instead of two, which is what this latency-bound
// kernel lacks (measured with whole images: matrix pipe 42 % busy, 47 % of the wave cycles waiting to issue).
// UACT / VACT >= 0: the activations are compile-time constants (sigmoid image, ReLU input map: the SVHN decoder); the
// run-time codes cost a compare / select chain per element in a kernel that is bound by instruction issue.
// DENSE (whole 16x16 -> 32x32 images, NT = 256): the halo ring of the gradient tile is zero padding that never changes, so it is
// zeroed once and a thread only ever stages INTERIOR elements — CU * 1024 of them = CU float4 per thread, read from the
// contiguous NCHW image with 16-byte loads, channel = float4 index (no per-element geometry, masks or guards: the generic
// path spends 14 4-byte loads x 2 tensors and ~25 compare / select / exec-mask instructions per element on them).
template <int CU, int CV, int NT, int PU, int OCC, int UACT, int VACT, bool DENSE>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void small_up_bwd_kernel(const float* __restrict__ dU, const float* __restrict__ Uout,
                                                           int u_act_rt, const float* __restrict__ V, int v_act_rt,
                                                           const float* __restrict__ Wref, float* __restrict__ dV,
                                                           float* __restrict__ partial, int n, int h, int w, int units,
                                                           mvk_prof_slot* prof) {
  mvk_prof_begin(prof);
  using C = SmallCfg<CU, CV>;
  const int u_act = UACT >= 0 ? UACT : u_act_rt, v_act = VACT >= 0 ? VACT : v_act_rt;
#ifdef MVK_ABLATE  // tools/smallup_ablate.sh: which part of the kernel bounds it (wrong results by construction)
  const int abl = (units >> 8) & 0xff;
  const int dephase = units >> 16;  // experiment: the second half of the grid starts `dephase` x ~0.4 us late
  units &= 0xff;
  if (blockIdx.x >= gridDim.x / 2)
    for (int q = 0; q < dephase; ++q) __builtin_amdgcn_s_sleep(100);
#else
  constexpr int abl = 0;
#endif
  constexpr int NW = NT / 64;    // waves per workgroup
  constexpr int WP = PU / NW;    // positions per wave
  constexpr int MT = WP / 16;    // 16-row MFMA tiles per wave
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int hu = h / units;      // input rows per unit
  const int P = hu * w, DH = 2 * hu + 2, DW = 2 * w + 2;
  float* Wt = smem;                       // [NC][WT]: Wt[k=(cu,tap)][cv] = Wref[cv][k]
  float* Ds = Wt + C::NC * C::WT;         // [CU][DH][DW] pre-activation gradient with halo
  float* Vs = Ds + ((CU * DH * DW + 3) & ~3);  // [P][VS]
  int* posoff = reinterpret_cast<int*>(Vs + P * C::VS);  // [P]: (2i)*DW + 2j
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  for (int i = tid; i < CV * C::NC; i += NT) {
    const int cv = i / C::NC, k = i - cv * C::NC;
    Wt[k * C::WT + cv] = Wref[i];
  }
  for (int p = tid; p < P; p += NT) posoff[p] = (2 * (p / w)) * DW + 2 * (p % w);
  const bool active = wave * WP < P;
  if (DENSE)
    for (int i = tid; i < CU * DH * DW; i += NT) Ds[i] = 0.f;  // the halo stays zero for the whole launch
  f32x4 qdu[CU], quo[CU];
  float dblq[CU];
#pragma unroll
  for (int c = 0; c < CU; ++c) dblq[c] = 0.f;
  const int dbase = ((tid >> 3) + 1) * 34 + (tid & 7) * 4 + 1;  // DENSE: tile offset of this thread's 4 pixels (row tid/8)

  f32x4 accw[C::NTV][CU];
#pragma unroll
  for (int a = 0; a < C::NTV; ++a)
#pragma unroll
    for (int b = 0; b < CU; ++b) accw[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 dbv[C::NTV];  // column sums of dV for this lane's channels cv = b*16 + 4*lq + r (positions: this lane's l15 group)
#pragma unroll
  for (int b = 0; b < C::NTV; ++b) dbv[b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int H2 = 2 * h, W2 = 2 * w;
  // register prefetch of the NEXT unit's three tiles (dU, Uout with halo indexing; V as float4)
  constexpr int ND = (CU * (PU / 8 + 2) * 34 + NT - 1) / NT;  // halo-tile elements per thread (w <= 16, hu <= PU / 16)
  constexpr int NV = PU * CV / 4 / NT;
  const int nd = CU * DH * DW, n4 = P * CV / 4;
  float pdu[ND], puo[ND];
  f32x4 pv[NV];
  float dslot[ND];  // bias-gradient partial of tile slot u (a slot always holds the same channel)
#pragma unroll
  for (int u = 0; u < ND; ++u) dslot[u] = 0.f;
  // halo-tile geometry is unit independent up to the row offset of the unit: offset of tile element (cu, y, x) for a
  // unit starting at output row 0 (may be negative), its tile row y, channel, and whether its column is inside the image
  int tinfo[ND];  // bits 0..7 y, 8..9 channel, 10 column inside the image and idx < nd, 12.. offset + W2 (>= 0)
#pragma unroll
  for (int u = 0; u < ND; ++u) {
    const int idx = tid + u * NT;
    const int idc = idx < nd ? idx : nd - 1;
    const int cu = idc / (DH * DW);
    const int rem = idc - cu * (DH * DW);
    const int y = rem / DW, x = rem - y * DW;
    const int ow = x - 1;
    const bool colin = idx < nd && ow >= 0 && ow < W2;
    tinfo[u] = y | (cu << 8) | ((int)colin << 10) | (((cu * H2 + y) * W2 + (colin ? ow : 0)) << 12);
  }
  const long long nunits = (long long)n * units;
  auto prefetch = [&](long long unit) __attribute__((always_inline)) {
    const long long img = unit >> (units - 1);           // units is 1 or 2
    const int r2 = 2 * (int)(unit & (units - 1)) * hu;  // first output row of the unit
    const float* du = dU + img * CU * H2 * W2;
    const float* uo = Uout + img * CU * H2 * W2;
    if (DENSE) {
#pragma unroll
      for (int k = 0; k < CU; ++k) {
        qdu[k] = reinterpret_cast<const f32x4*>(du)[tid + k * NT];
        quo[k] = reinterpret_cast<const f32x4*>(uo)[tid + k * NT];
      }
    } else
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int oh = r2 + (tinfo[u] & 255) - 1;
      const bool in = ((tinfo[u] >> 10) & 1) && oh >= 0 && oh < H2;
      const int off = in ? (tinfo[u] >> 12) + (r2 - 1) * W2 : 0;  // clamped: the load is unconditional, the padding zeroed
      pdu[u] = du[off];  // raw: the padding mask is applied when the tile is staged, so nothing waits on these loads
      puo[u] = uo[off];  // before the MFMA phase of the current unit
    }
    const f32x4* src = reinterpret_cast<const f32x4*>(V + (img * h * w + (long long)(r2 >> 1) * w) * CV);
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int idx = tid + u * NT;
      pv[u] = src[(DENSE || idx < n4) ? idx : n4 - 1];
    }
  };
  if ((long long)blockIdx.x < nunits) prefetch(blockIdx.x);
#ifdef MVK_SUPROF  // tools/smallup_phase.sh: per-wave cycle counters
  unsigned long long su_t[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long su_t0 = __builtin_readcyclecounter();
#define SU_T(i) { const unsigned long long n_ = __builtin_readcyclecounter(); su_t[i] += n_ - su_last; su_last = n_; }
#else
#define SU_T(i)
#endif
  for (long long unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
#ifdef MVK_SUPROF
    unsigned long long su_last = __builtin_readcyclecounter();
#endif
    __syncthreads();  // previous unit's tiles are no longer read
    SU_T(0)
    // --- stage dUpre with halo (sigmoid' applied here), bias-gradient partials, and the V tile
    const int r2s = 2 * (int)(unit & (units - 1)) * hu;  // first output row of the unit being staged
    if (DENSE) {
      if (!(abl & 16)) {
#pragma unroll
        for (int k = 0; k < CU; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = qdu[k][e] * mvk_act_grad_from_out(quo[k][e], u_act);
            Ds[k * 34 * 34 + dbase + e] = v;
            dblq[k] += v;
          }
      }
    } else
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int idx = tid + u * NT;
      if (idx < nd && !(abl & 16)) {
        const int oh = r2s + (tinfo[u] & 255) - 1;
        const bool in = ((tinfo[u] >> 10) & 1) && oh >= 0 && oh < H2;
        const float v = in ? pdu[u] * mvk_act_grad_from_out(puo[u], u_act) : 0.f;  // zero padding outside the image
        Ds[idx] = v;
        const int y = tinfo[u] & 255;
        dslot[u] += (y >= 1 && y <= 2 * hu) ? v : 0.f;  // halo rows belong to the neighbouring unit (or are padding)
      }
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int idx = tid + u * NT;
      if ((DENSE || idx < n4) && !(abl & 32)) {
        const int pos = idx / (CV / 4), q = idx - pos * (CV / 4);
        *reinterpret_cast<f32x4*>(Vs + pos * C::VS + 4 * q) = pv[u];
      }
    }
    SU_T(1)
    __syncthreads();
    SU_T(0)
    if (unit + gridDim.x < nunits && !(abl & 8)) prefetch(unit + gridDim.x);
    SU_T(2)
    if (!DENSE && !active) continue;
    // --- backward data: dV[pos][cv] = sum_{k=(cu,kh,kw)} dUpre[cu][2i-1+kh][2j-1+kw] * W[cv][k]
    f32x4 acc[MT][C::NTV];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < C::NTV; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    int po[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a) po[a] = posoff[wave * WP + a * 16 + l15];
#pragma unroll 2
    for (int ks = 0; ks < ((abl & 1) ? 1 : C::NC / 4); ++ks) {
      const int k = ks * 4 + lq;
      const int koff = (k >> 4) * DH * DW + ((k >> 2) & 3) * DW + (k & 3);
      float av[MT], bv[C::NTV];
#pragma unroll
      for (int a = 0; a < MT; ++a) av[a] = Ds[koff + po[a]];
#pragma unroll
      for (int b = 0; b < C::NTV; ++b) bv[b] = Wt[k * C::WT + b * 16 + l15];
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < C::NTV; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[b], av[a], acc[a][b], 0, 0, 0);  // transposed tile
    }
    SU_T(3)
    if (!(abl & 64)) {
      // operands swapped: this lane holds dV[pos = a*16 + l15][cv = b*16 + 4*lq .. +3] -> 16-byte mask reads and stores
      const long long img = unit >> (units - 1);
      const int r0 = (int)(unit & (units - 1)) * hu;
      float* dv = dV + (img * h * w + (long long)r0 * w) * CV;
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < C::NTV; ++b) {
          const int pos = wave * WP + a * 16 + l15, cv = b * 16 + lq * 4;
          const f32x4 vin = *reinterpret_cast<const f32x4*>(Vs + pos * C::VS + cv);
          f32x4 g;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            g[r] = acc[a][b][r] * mvk_act_grad_from_out(vin[r], v_act);
            dbv[b][r] += g[r];
          }
          if (!(abl & 4)) *reinterpret_cast<f32x4*>(dv + pos * CV + cv) = g;
        }
    }
    SU_T(4)
    // --- backward weight: dW[cv][k] += sum_pos V[pos][cv] * dUpre(gathered)[pos][k]; this wave's WP positions
#pragma unroll 2
    for (int ks = 0; ks < ((abl & 2) ? 1 : WP / 4); ++ks) {
      const int kpos = wave * WP + ks * 4 + lq;
      const int pbase = posoff[kpos];
      float av[C::NTV], bv[CU];
#pragma unroll
      for (int a = 0; a < C::NTV; ++a) av[a] = Vs[kpos * C::VS + a * 16 + l15];
#pragma unroll
      for (int b = 0; b < CU; ++b) bv[b] = Ds[b * DH * DW + (l15 >> 2) * DW + (l15 & 3) + pbase];
#pragma unroll
      for (int a = 0; a < C::NTV; ++a)
#pragma unroll
        for (int b = 0; b < CU; ++b)
          accw[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], accw[a][b], 0, 0, 0);
    }
    SU_T(5)
  }
#ifdef MVK_SUPROF
  if (g_su_dbg && lane == 0) {
    unsigned long long* o = g_su_dbg + (blockIdx.x * NW + wave) * 8;
    o[0] = __builtin_readcyclecounter() - su_t0;
#pragma unroll
    for (int i = 0; i < 6; ++i) o[1 + i] = su_t[i];
  }
#endif
  // --- cross-wave reduction of the weight / bias partials, one slab per workgroup
  __syncthreads();
  float* red = smem;  // [4][CV*NC] — reuses Wt/Ds/Vs (needs 4*CV*NC floats); waves 4.. add into the slab of wave-4
#pragma unroll 1
  for (int pass = 0; pass < NW / 4; ++pass) {
    if ((wave >> 2) == pass) {
#pragma unroll
      for (int a = 0; a < C::NTV; ++a)
#pragma unroll
        for (int b = 0; b < CU; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int cv = a * 16 + lq * 4 + r, col = b * 16 + l15;
            float* dst = red + (wave & 3) * CV * C::NC + cv * C::NC + col;
            *dst = pass == 0 ? accw[a][b][r] : *dst + accw[a][b][r];
          }
    }
    __syncthreads();
  }
  float* slab = partial + (long long)blockIdx.x * (CV * C::NC + CU + CV);
  for (int i = tid; i < CV * C::NC; i += NT)
    slab[i] = red[i] + red[CV * C::NC + i] + red[2 * CV * C::NC + i] + red[3 * CV * C::NC + i];
  __syncthreads();
  // bias partials
  float dbl[CU];
#pragma unroll
  for (int c = 0; c < CU; ++c) dbl[c] = 0.f;
  if (DENSE) {
#pragma unroll
    for (int c = 0; c < CU; ++c) dbl[c] = dblq[c];
  } else
#pragma unroll
  for (int u = 0; u < ND; ++u)
#pragma unroll
    for (int c = 0; c < CU; ++c)
      if (c == ((tinfo[u] >> 8) & 3)) dbl[c] += dslot[u];
  float* bred = smem;
#pragma unroll
  for (int c = 0; c < CU; ++c) {
    const float s = wave_sum(dbl[c]);
    if (lane == 0) bred[c * NW + wave] = s;
  }
  __syncthreads();
  if (tid < CU) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < NW; ++q) t += bred[tid * NW + q];
    slab[CV * C::NC + tid] = t;
  }
  __syncthreads();
  // column sums of dV: combine the 16 position groups of each wave, then the waves (fixed order)
  float* vred = smem;  // [CV][NW*16]
#pragma unroll
  for (int b = 0; b < C::NTV; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) vred[(b * 16 + lq * 4 + r) * (NW * 16) + wave * 16 + l15] = dbv[b][r];
  __syncthreads();
  if (tid < CV) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < NW * 16; ++q) t += vred[tid * (NW * 16) + q];
    slab[CV * C::NC + CU + tid] = t;
  }
  mvk_prof_end(prof);
}

// ---------------------------------------------------------------------------------------------------------
typedef __bf16 su_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mvk::bf16x8 su_tr_pair(const char* p0, const char* p1) {
  typedef __attribute__((address_space(3))) su_bf16x4* lp;
  const su_bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(p0));
  const su_bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(p1));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// The backward for the SVHN decoder's shape (16x16 inputs, Cv = 32, sigmoid image, ReLU input map) on the bf16 matrix cores.
// The fp32 kernel above spends 55 % of a wave's time in its two v_mfma_f32_16x16x4_f32 loops; here both GEMMs take 6 products of
// order <= 2 of bf16 pieces (x = x0 + x1 + x2, bf3.hpp) on v_mfma_f32_16x16x32_bf16: 192 MFMAs of 16 cycles per wave and image
// instead of 192 of 32, fp32-level error.  The weight gradient reduces over positions while both of its operands are stored
// [position][channel]: gfx950's transposing LDS read (ds_read_b64_tr_b16) delivers them k-contiguous, as in imgwgrad_kernel.
// NO im2col: the gradient image is staged pixel-major into four PARITY planes
// Dp[y' & 1][x' & 1][y' >> 1][x' >> 1][4 channels (3 + a zero)] (y' = y + 1, x' = x + 1: 17 x 17 entries of 8 bytes per plane, zero
// halo), so the 3 channels of a tap are one 8-byte piece and K is ordered k' = 4 tap + channel (64 with the zero channel):
//   backward data    the k-octet (taps 2 o, 2 o + 1) of a position is two 8-byte reads from two parity planes;
//   backward weight  the transposing read takes per-lane addresses: lane (row, piece) reads the piece of tap 4 b + piece at
//                    its position's (parity, shifted) entry — the im2col matrix is never written.
// 76 KB of LDS, 256 threads: two workgroups per CU like the fp32 kernel.
// PRE (round 3, the fused tail of small_up_fwd_bf_kernel): dU already is the gradient with respect to the layer's pre-activation
// per unit of its image's score; it is multiplied by rowscale[img] (the gradient of the loss with respect to that score) while it is
// staged, and the layer's output is not read at all.
template <int CU, bool PRE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void small_up_bwd_bf_kernel(
    const float* __restrict__ dU, const float* __restrict__ Uout, const float* __restrict__ V, const float* __restrict__ Wref,
    float* __restrict__ dV, float* __restrict__ partial, int n, mvk_prof_slot* prof, const float* __restrict__ rowscale = nullptr,
    float* dv_amax = nullptr) {
  mvk_prof_begin(prof);
  float amax_l = 0.f;  // max |dV| of this thread's stores (published at the end when dv_amax is given)
  using mvk::bf16x8;
  using mvk::u32x2;
  using mvk::u32x4;
  static_assert(CU == 3, "three image channels + one zero channel per 8-byte piece");
  constexpr int CV = 32, NC = 16 * CU, P = 256, NT = 256, NW = 4, MT = 4;
  constexpr int PL1 = 289 * 8;        // bytes of one parity plane (17 x 17 pieces)
  constexpr int PLD = 4 * PL1;        // bytes of one bf16 piece plane of the gradient
  constexpr int VPL = P * 64;         // bytes of one piece plane of V [pos][32 cv]
  constexpr int OFF_V = 3 * PLD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* Dp = reinterpret_cast<char*>(smem);
  char* Vp = Dp + OFF_V;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  for (int i = tid; i < OFF_V / 4; i += NT) smem[i] = 0.f;  // the halo entries stay zero for the whole launch
  // weight pieces of the backward-data GEMM: lane (cv = 16 b + l15, k'-octet o = 4 s + lq = taps 2 o, 2 o + 1 x 4 channels)
  bf16x8 wf[2][2][3];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int o = 4 * s2 + lq;
      const float* wr = Wref + (b * 16 + l15) * NC + 2 * o;  // + 16 ch + tt
      unsigned pc[3][4];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        mvk::bf3_split(wr[tt], wr[16 + tt], pc[0][2 * tt], pc[1][2 * tt], pc[2][2 * tt]);
        mvk::bf3_split(wr[32 + tt], 0.f, pc[0][2 * tt + 1], pc[1][2 * tt + 1], pc[2][2 * tt + 1]);
      }
#pragma unroll
      for (int p = 0; p < 3; ++p) wf[b][s2][p] = __builtin_bit_cast(bf16x8, u32x4{pc[p][0], pc[p][1], pc[p][2], pc[p][3]});
    }
  f32x4 accw[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) accw[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 dbv[MT][2];
#pragma unroll
  for (int a = 0; a < MT; ++a) dbv[a][0] = dbv[a][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dblq[CU] = {0.f, 0.f, 0.f};
  // staging: this thread's 4 pixels (row tid >> 3, columns 4 (tid & 7) ..) of every channel -> their parity-plane pieces
  int dst[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int yp = (tid >> 3) + 1, xp = (tid & 7) * 4 + e + 1;
    dst[e] = (((yp & 1) * 2 + (xp & 1)) * 289 + (yp >> 1) * 17 + (xp >> 1)) * 8;
  }
  f32x4 qdu[CU], quo[CU], pv[8];
  float rs_next = 1.f;
  // (the 14 loads of the next image go out in ONE burst right behind the staging: spread over the backward-data tiles they
  // arrive too late — staging then waits 38 % of the wave's cycles for them, 136 us instead of 110)
  auto prefetch_part = [&](long long img, int part) __attribute__((always_inline)) {
    if (part < CU) {
      qdu[part] = reinterpret_cast<const f32x4*>(dU + img * (CU * 1024))[tid + part * NT];
      if (!PRE) quo[part] = reinterpret_cast<const f32x4*>(Uout + img * (CU * 1024))[tid + part * NT];
    }
    if (PRE && part == 0) rs_next = rowscale ? rowscale[img] : 1.f;
    const f32x4* src = reinterpret_cast<const f32x4*>(V + img * (P * CV));
    pv[2 * part] = src[tid + (2 * part) * NT];
    pv[2 * part + 1] = src[tid + (2 * part + 1) * NT];
  };
  auto prefetch = [&](long long img) __attribute__((always_inline)) {
#pragma unroll
    for (int part = 0; part < 4; ++part) prefetch_part(img, part);
  };
  // entry of tap (kh, kw) for input position (i, j): parity (kh & 1, kw & 1), entry (i + (kh >> 1), j + (kw >> 1))
  auto dentry = [](int pos, int tap) {
    const int i = pos >> 4, j = pos & 15, kh = tap >> 2, kw = tap & 3;
    return ((((kh & 1) * 2 + (kw & 1)) * 289) + (i + (kh >> 1)) * 17 + j + (kw >> 1)) * 8;
  };
  int daddr[MT][2];  // backward data: first tap of this lane's octet (the second: the next parity plane, + PL1)
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) daddr[a][s2] = dentry((wave * MT + a) * 16 + l15, 2 * (4 * s2 + lq));
  int vaddr[2], waddr[2][4];  // weight gradient: k-steps 2 wave, 2 wave + 1; rows 8 lq + (l15 >> 2) (+ 4), piece l15 & 3
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int pos0 = 32 * (2 * wave + ks) + 8 * lq + (l15 >> 2);
    vaddr[ks] = pos0 * 64 + (4 * (l15 & 3)) * 2;
#pragma unroll
    for (int b = 0; b < 4; ++b) waddr[ks][b] = dentry(pos0, 4 * b + (l15 & 3));
  }
  constexpr int PA[6] = {0, 1, 2, 0, 1, 0};  // smallest terms first
  constexpr int PB[6] = {2, 1, 0, 1, 0, 0};
  long long img = blockIdx.x;
  if (img < n) prefetch(img);
#ifdef MVK_SUPROF  // tools/smallup_phase.sh: per-wave cycle counters
  unsigned long long su_t[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long su_t0 = __builtin_readcyclecounter();
#endif
  for (; img < n; img += gridDim.x) {
#ifdef MVK_SUPROF
    unsigned long long su_last = __builtin_readcyclecounter();
#endif
    __syncthreads();  // the previous image's planes are no longer read
    SU_T(0)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v[CU];
#pragma unroll
      for (int c = 0; c < CU; ++c) {
        v[c] = PRE ? qdu[c][e] * rs_next : qdu[c][e] * (quo[c][e] * (1.f - quo[c][e]));  // sigmoid'
        dblq[c] += v[c];
      }
      unsigned a0, a1, a2, b0, b1, b2;
      mvk::bf3_split(v[0], v[1], a0, a1, a2);
      mvk::bf3_split(v[2], 0.f, b0, b1, b2);
      *reinterpret_cast<u32x2*>(Dp + dst[e]) = u32x2{a0, b0};
      *reinterpret_cast<u32x2*>(Dp + PLD + dst[e]) = u32x2{a1, b1};
      *reinterpret_cast<u32x2*>(Dp + 2 * PLD + dst[e]) = u32x2{a2, b2};
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = tid + u * NT;
      unsigned a0, a1, a2, b0, b1, b2;
      mvk::bf3_split(pv[u][0], pv[u][1], a0, a1, a2);
      mvk::bf3_split(pv[u][2], pv[u][3], b0, b1, b2);
      const int off = (idx >> 3) * 64 + (idx & 7) * 8;
      *reinterpret_cast<u32x2*>(Vp + off) = u32x2{a0, b0};
      *reinterpret_cast<u32x2*>(Vp + VPL + off) = u32x2{a1, b1};
      *reinterpret_cast<u32x2*>(Vp + 2 * VPL + off) = u32x2{a2, b2};
    }
    SU_T(1)
    __syncthreads();
    SU_T(0)
    if (img + gridDim.x < n) prefetch(img + gridDim.x);
    SU_T(2)
    // --- backward data (transposed tile: rows = channels, columns = positions) + ReLU mask + channel sums
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      f32x4 c2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 cf[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const u32x2 t0 = *reinterpret_cast<const u32x2*>(Dp + p * PLD + daddr[a][s2]);
          const u32x2 t1 = *reinterpret_cast<const u32x2*>(Dp + p * PLD + daddr[a][s2] + PL1);
          cf[p] = __builtin_bit_cast(bf16x8, u32x4{t0[0], t0[1], t1[0], t1[1]});
        }
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int t = 0; t < 6; ++t) c2[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b][s2][PA[t]], cf[PB[t]], c2[b], 0, 0, 0);
      }
      const int dpos = (wave * MT + a) * 16 + l15;
      float* dv = dV + img * (P * CV) + dpos * CV;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int cv = b * 16 + lq * 4;
        const u32x2 vb = *reinterpret_cast<const u32x2*>(Vp + dpos * 64 + cv * 2);  // leading pieces of V: same sign, zero iff V = 0
        const unsigned hv[4] = {vb[0] << 16, vb[0] & 0xffff0000u, vb[1] << 16, vb[1] & 0xffff0000u};
        f32x4 gq;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gq[r] = __uint_as_float(hv[r]) > 0.f ? c2[b][r] : 0.f;
          dbv[a][b][r] += gq[r];
        }
        amax_l = fmaxf(fmaxf(amax_l, fmaxf(fabsf(gq[0]), fabsf(gq[1]))), fmaxf(fabsf(gq[2]), fabsf(gq[3])));
        *reinterpret_cast<f32x4*>(dv + cv) = gq;
      }
    }
    SU_T(3)
    // --- backward weight: this wave's two 32-position k-steps, both channel tiles, the four (4 taps x 4 channels) column tiles
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 va[2][3];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int p = 0; p < 3; ++p) va[a][p] = su_tr_pair(Vp + p * VPL + vaddr[ks] + a * 32, Vp + p * VPL + vaddr[ks] + a * 32 + 4 * 64);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        bf16x8 cb[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) cb[p] = su_tr_pair(Dp + p * PLD + waddr[ks][b], Dp + p * PLD + waddr[ks][b] + 4 * 8);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int t = 0; t < 6; ++t)
            accw[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va[a][PA[t]], cb[PB[t]], accw[a][b], 0, 0, 0);
      }
    }
    SU_T(5)
  }
#ifdef MVK_SUPROF
  if (g_su_dbg && lane == 0) {
    unsigned long long* o = g_su_dbg + (blockIdx.x * NW + wave) * 8;
    o[0] = __builtin_readcyclecounter() - su_t0;
#pragma unroll
    for (int i = 0; i < 6; ++i) o[1 + i] = su_t[i];
  }
#endif
  // --- one slab per workgroup: the waves' weight-gradient partials in order, bias partials, channel sums of dV
  __syncthreads();
  float* red = reinterpret_cast<float*>(Vp);  // [NW][CV][64 columns k' = 4 tap + channel]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave * (CV * 64) + (a * 16 + lq * 4 + r) * 64 + b * 16 + l15] = accw[a][b][r];
  __syncthreads();
  float* slab = partial + (long long)blockIdx.x * (CV * NC + CU + CV);
  for (int i = tid; i < CV * NC; i += NT) {
    const int cv = i / NC, k = i - cv * NC, src = cv * 64 + (k & 15) * 4 + (k >> 4);  // k = 16 channel + tap
    slab[i] = ((red[src] + red[CV * 64 + src]) + red[2 * CV * 64 + src]) + red[3 * CV * 64 + src];
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < CU; ++c) {
    const float sdb = wave_sum(dblq[c]);
    if (lane == 0) red[c * NW + wave] = sdb;
  }
  __syncthreads();
  if (tid < CU) slab[CV * NC + tid] = ((red[tid * NW] + red[tid * NW + 1]) + red[tid * NW + 2]) + red[tid * NW + 3];
  __syncthreads();
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(b * 16 + lq * 4 + r) * P + (wave * MT + a) * 16 + l15] = dbv[a][b][r];
  __syncthreads();
  {  // 8 threads per channel sum 32 positions each, then the 8 partial sums in order
    const int cv = tid >> 3, seg = tid & 7;
    float t = 0.f;
#pragma unroll 8
    for (int q = 0; q < 32; ++q) t += red[cv * P + seg * 32 + q];
    red[CV * P + tid] = t;
  }
  __syncthreads();
  if (tid < CV) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[CV * P + tid * 8 + q];
    slab[CV * NC + CU + tid] = t;
  }
  if (dv_amax) mvk::amax_publish(amax_l, dv_amax, red);  // uniform branch; synchronises before it touches red
  mvk_prof_end(prof);
}

// ---------------------------------------------------------------------------------------------------------
// The scaled-fp16 form of small_up_bwd_bf_kernel<CU, PRE = true> (bf3.hpp: x s = hi + lo / 2048, a product = hi hi' + (hi lo' +
// lo hi') / 2048 in a main and a cross accumulator): 96 MFMAs per wave and image instead of 192, two piece planes instead of
// three (51 KB of LDS).  Same data flow, same parity planes, same transposing reads.  The scales:
//   gradient  s_d from du_amax x max |rowscale| — the fused tail publishes a bound of what it stores (amax protocol; from its
//             largest row sum, MVK_SUH_GMAX), the largest row weight is found in the prologue (n floats, L2-resident).  ONE
//             scale for the launch: the weight gradient accumulates over the images of a workgroup in registers;
//   V         s_v from v_amax, the bound the 64 -> 32 launch published; weights: the workgroup computes max |W| itself.
// The ReLU mask of dV comes from the pieces of V: V >= 0 by contract (v_act = ReLU), and V > 0 iff one of its two pieces is
// non-zero down to 2^-49 of the tensor's maximum (hi alone would round 2^-39 of it to zero).
typedef _Float16 su_f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mvk::f16x8 su_tr_pair_h(const char* p0, const char* p1) {
  typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
  typedef __attribute__((address_space(3))) h4* lp;
  const su_f16x4 lo = __builtin_bit_cast(su_f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(p0)));
  const su_f16x4 hi = __builtin_bit_cast(su_f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(p1)));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int CU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void small_up_bwd_h_kernel(
    const float* __restrict__ dU, const float* __restrict__ V, const float* __restrict__ Wref, float* __restrict__ dV,
    float* __restrict__ partial, int n, mvk_prof_slot* prof, const float* __restrict__ rowscale,
    const float* __restrict__ du_amax, const float* __restrict__ v_amax, float* dv_amax) {
  mvk_prof_begin(prof);
  float amax_l = 0.f;  // max |dV| of this thread's stores (published at the end when dv_amax is given)
  using mvk::f16x8;
  using mvk::u32x2;
  using mvk::u32x4;
  static_assert(CU == 3, "three image channels + one zero channel per 8-byte piece");
  constexpr int CV = 32, NC = 16 * CU, P = 256, NT = 256, NW = 4, MT = 4;
  constexpr int PL1 = 289 * 8;        // bytes of one parity plane (17 x 17 pieces)
  constexpr int PLD = 4 * PL1;        // bytes of one fp16 piece plane of the gradient
  constexpr int VPL = P * 64;         // bytes of one piece plane of V [pos][32 cv]
  constexpr int OFF_V = 2 * PLD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* Dp = reinterpret_cast<char*>(smem);
  char* Vp = Dp + OFF_V;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  for (int i = tid; i < OFF_V / 4; i += NT) smem[i] = 0.f;  // the halo entries stay zero for the whole launch
  // ---- scales: max |W| and max |rowscale| of the launch (every workgroup finds the same two numbers)
  float wmax = 0.f, rmax = rowscale ? 0.f : 1.f;
  for (int i = tid; i < CV * NC; i += NT) wmax = fmaxf(wmax, fabsf(Wref[i]));
  if (rowscale)
    for (int i = tid; i < n; i += NT) rmax = fmaxf(rmax, fabsf(rowscale[i]));
  wmax = __uint_as_float(wave_max_dpp_bits(wmax));  // (no lane shuffles: their address registers would stay live across the loop)
  rmax = __uint_as_float(wave_max_dpp_bits(rmax));
  {
    float* sc = reinterpret_cast<float*>(Vp);  // read by everyone before the first staging barrier below
    if (lane == 0) sc[2 * wave] = wmax, sc[2 * wave + 1] = rmax;
    __syncthreads();
    wmax = fmaxf(fmaxf(sc[0], sc[2]), fmaxf(sc[4], sc[6]));
    rmax = fmaxf(fmaxf(sc[1], sc[3]), fmaxf(sc[5], sc[7]));
  }
  const float sw = mvk::f16_scale_of(wmax), sv = mvk::f16_scale_of(*v_amax), sd = mvk::f16_scale_of(*du_amax * rmax);
  const float inv_dv = mvk::f16_inv_scale(sw) * mvk::f16_inv_scale(sd), inv_dw = mvk::f16_inv_scale(sv) * mvk::f16_inv_scale(sd);
  // weight pieces of the backward-data GEMM: lane (cv = 16 b + l15, k'-octet o = 4 s + lq = taps 2 o, 2 o + 1 x 4 channels)
  f16x8 wf[2][2][2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int o = 4 * s2 + lq;
      const float* wr = Wref + (b * 16 + l15) * NC + 2 * o;  // + 16 ch + tt
      unsigned pc[2][4];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        mvk::f16_split(wr[tt] * sw, wr[16 + tt] * sw, pc[0][2 * tt], pc[1][2 * tt]);
        mvk::f16_split(wr[32 + tt] * sw, 0.f, pc[0][2 * tt + 1], pc[1][2 * tt + 1]);
      }
#pragma unroll
      for (int p = 0; p < 2; ++p) wf[b][s2][p] = __builtin_bit_cast(f16x8, u32x4{pc[p][0], pc[p][1], pc[p][2], pc[p][3]});
    }
  f32x4 accw[2][4], accx[2][4];  // weight gradient: main and cross terms
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) accw[a][b] = accx[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 dbv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};  // channel sums of dV: this lane's 4 + 4 channels
  float dblq[CU] = {0.f, 0.f, 0.f};
  // staging: this thread's 4 pixels (row tid >> 3, columns 4 (tid & 7) ..) of every channel -> their parity-plane pieces
  int dst[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int yp = (tid >> 3) + 1, xp = (tid & 7) * 4 + e + 1;
    dst[e] = (((yp & 1) * 2 + (xp & 1)) * 289 + (yp >> 1) * 17 + (xp >> 1)) * 8;
  }
  f32x4 qdu[CU], pv[8];
  float rs_next = 1.f;
  auto prefetch = [&](long long img) __attribute__((always_inline)) {  // one burst right behind the staging (see the bf16 form)
#pragma unroll
    for (int c = 0; c < CU; ++c) qdu[c] = reinterpret_cast<const f32x4*>(dU + img * (CU * 1024))[tid + c * NT];
    rs_next = rowscale ? rowscale[img] : 1.f;
    const f32x4* src = reinterpret_cast<const f32x4*>(V + img * (P * CV));
#pragma unroll
    for (int u = 0; u < 8; ++u) pv[u] = src[tid + u * NT];
  };
  // entry of tap (kh, kw) for input position (i, j): parity (kh & 1, kw & 1), entry (i + (kh >> 1), j + (kw >> 1))
  auto dentry = [](int pos, int tap) {
    const int i = pos >> 4, j = pos & 15, kh = tap >> 2, kw = tap & 3;
    return ((((kh & 1) * 2 + (kw & 1)) * 289) + (i + (kh >> 1)) * 17 + j + (kw >> 1)) * 8;
  };
  int daddr[MT][2];  // backward data: first tap of this lane's octet (the second: the next parity plane, + PL1)
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) daddr[a][s2] = dentry((wave * MT + a) * 16 + l15, 2 * (4 * s2 + lq));
  int vaddr[2], waddr[2][4];  // weight gradient: k-steps 2 wave, 2 wave + 1; rows 8 lq + (l15 >> 2) (+ 4), piece l15 & 3
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int pos0 = 32 * (2 * wave + ks) + 8 * lq + (l15 >> 2);
    vaddr[ks] = pos0 * 64 + (4 * (l15 & 3)) * 2;
#pragma unroll
    for (int b = 0; b < 4; ++b) waddr[ks][b] = dentry(pos0, 4 * b + (l15 & 3));
  }
  long long img = blockIdx.x;
  if (img < n) prefetch(img);
  for (; img < n; img += gridDim.x) {
    __syncthreads();  // the previous image's planes (first pass: the scale words) are no longer read
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v[CU];
#pragma unroll
      for (int c = 0; c < CU; ++c) {
        v[c] = qdu[c][e] * rs_next;
        dblq[c] += v[c];
      }
      unsigned a0, a1, b0, b1;
      mvk::f16_split_su(v[0], v[1], sd, a0, a1);
      mvk::f16_split_su(v[2], 0.f, sd, b0, b1);
      *reinterpret_cast<u32x2*>(Dp + dst[e]) = u32x2{a0, b0};
      *reinterpret_cast<u32x2*>(Dp + PLD + dst[e]) = u32x2{a1, b1};
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = tid + u * NT;
      unsigned a0, a1, b0, b1;
      mvk::f16_split_su(pv[u][0], pv[u][1], sv, a0, a1);
      mvk::f16_split_su(pv[u][2], pv[u][3], sv, b0, b1);
      const int off = (idx >> 3) * 64 + (idx & 7) * 8;
      *reinterpret_cast<u32x2*>(Vp + off) = u32x2{a0, b0};
      *reinterpret_cast<u32x2*>(Vp + VPL + off) = u32x2{a1, b1};
    }
    __syncthreads();
    if (img + gridDim.x < n) prefetch(img + gridDim.x);
    // --- backward data (transposed tile: rows = channels, columns = positions) + ReLU mask + channel sums
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      f32x4 cm[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, cx[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        f16x8 cf[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const u32x2 t0 = *reinterpret_cast<const u32x2*>(Dp + p * PLD + daddr[a][s2]);
          const u32x2 t1 = *reinterpret_cast<const u32x2*>(Dp + p * PLD + daddr[a][s2] + PL1);
          cf[p] = __builtin_bit_cast(f16x8, u32x4{t0[0], t0[1], t1[0], t1[1]});
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          cx[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[b][s2][0], cf[1], cx[b], 0, 0, 0);
          cm[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[b][s2][0], cf[0], cm[b], 0, 0, 0);
          cx[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[b][s2][1], cf[0], cx[b], 0, 0, 0);
        }
      }
      const int dpos = (wave * MT + a) * 16 + l15;
      float* dv = dV + img * (P * CV) + dpos * CV;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int cv = b * 16 + lq * 4;
        const u32x2 vh = *reinterpret_cast<const u32x2*>(Vp + dpos * 64 + cv * 2);
        const u32x2 vl = *reinterpret_cast<const u32x2*>(Vp + VPL + dpos * 64 + cv * 2);
        const unsigned m0 = vh[0] | vl[0], m1 = vh[1] | vl[1];
        const bool on[4] = {(m0 & 0x7fffu) != 0u, (m0 & 0x7fff0000u) != 0u, (m1 & 0x7fffu) != 0u, (m1 & 0x7fff0000u) != 0u};
        f32x4 gq;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gq[r] = on[r] ? fmaf(cx[b][r], 1.f / 2048.f, cm[b][r]) * inv_dv : 0.f;
          dbv[b][r] += gq[r];
        }
        amax_l = fmaxf(fmaxf(amax_l, fmaxf(fabsf(gq[0]), fabsf(gq[1]))), fmaxf(fabsf(gq[2]), fabsf(gq[3])));
        *reinterpret_cast<f32x4*>(dv + cv) = gq;
      }
    }
    // --- backward weight: this wave's two 32-position k-steps, both channel tiles, the four (4 taps x 4 channels) column tiles
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8 va[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int p = 0; p < 2; ++p) va[a][p] = su_tr_pair_h(Vp + p * VPL + vaddr[ks] + a * 32, Vp + p * VPL + vaddr[ks] + a * 32 + 4 * 64);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        f16x8 cb[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) cb[p] = su_tr_pair_h(Dp + p * PLD + waddr[ks][b], Dp + p * PLD + waddr[ks][b] + 4 * 8);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          accx[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va[a][0], cb[1], accx[a][b], 0, 0, 0);
          accw[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va[a][0], cb[0], accw[a][b], 0, 0, 0);
          accx[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va[a][1], cb[0], accx[a][b], 0, 0, 0);
        }
      }
    }
  }
  // --- one slab per workgroup: the waves' weight-gradient partials in order, bias partials, channel sums of dV
  __syncthreads();
  float* red = reinterpret_cast<float*>(Vp);  // [NW][CV][64 columns k' = 4 tap + channel] = the two piece planes of V
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        red[wave * (CV * 64) + (a * 16 + lq * 4 + r) * 64 + b * 16 + l15] = fmaf(accx[a][b][r], 1.f / 2048.f, accw[a][b][r]) * inv_dw;
  __syncthreads();
  float* slab = partial + (long long)blockIdx.x * (CV * NC + CU + CV);
  for (int i = tid; i < CV * NC; i += NT) {
    const int cv = i / NC, k = i - cv * NC, src = cv * 64 + (k & 15) * 4 + (k >> 4);  // k = 16 channel + tap
    slab[i] = ((red[src] + red[CV * 64 + src]) + red[2 * CV * 64 + src]) + red[3 * CV * 64 + src];
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < CU; ++c) {
    const float sdb = wave_sum(dblq[c]);
    if (lane == 0) red[c * NW + wave] = sdb;
  }
  __syncthreads();
  if (tid < CU) slab[CV * NC + tid] = ((red[tid * NW] + red[tid * NW + 1]) + red[tid * NW + 2]) + red[tid * NW + 3];
  __syncthreads();
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(b * 16 + lq * 4 + r) * 64 + wave * 16 + l15] = dbv[b][r];
  __syncthreads();
  {  // 8 threads per channel sum 8 entries each, then the 8 partial sums in order
    const int cv = tid >> 3, seg = tid & 7;
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[cv * 64 + seg * 8 + q];
    red[CV * 64 + tid] = t;
  }
  __syncthreads();
  if (tid < CV) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[CV * 64 + tid * 8 + q];
    slab[CV * NC + CU + tid] = t;
  }
  if (dv_amax) mvk::amax_publish(amax_l, dv_amax, red);  // uniform branch; synchronises before it touches red
  mvk_prof_end(prof);
}

// ---------------------------------------------------------------------------------------------------------
// forward of the image-CONSUMING layer: V[n,h,w,Cv] = act(conv4s2(U[n,Cu,2h,2w]) + b), Cu <= 4 (svhn.py:13-15 first
// Conv2d).  Same tile algebra as the backward-data part above (the gradient of the image-producing ConvTranspose IS this
// convolution): the NCHW image with a zero halo is staged in LDS with coalesced loads (persistent workgroups, next
// image prefetched in registers), the 16 Cu window of every position is gathered from LDS into the A operand of
// v_mfma_f32_16x16x4_f32 (exact fp32), the packed weight [16 Cu][Cv] sits in LDS for the whole launch.
// ---------------------------------------------------------------------------------------------------------
template <int CU, int CV, int NT>
__global__ __launch_bounds__(NT) void small_down_fwd_kernel(const float* __restrict__ U, const float* __restrict__ Wdown,
                                                            const float* __restrict__ bias, float* __restrict__ V, int n,
                                                            int h, int w, int act, int wref) {
  // wref: Wdown is the layer's REFERENCE weight [Cv][Cu][4][4] (1536 floats at 3 -> 32 channels): the first kernel of the
  // encoder then does not wait for the step's weight-pack launch, which runs beside it
  using C = SmallCfg<CU, CV>;
  constexpr int NW = NT / 64, WP = 256 / NW, MT = WP / 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int P = h * w, DH = 2 * h + 2, DW = 2 * w + 2;
  float* Wt = smem;                                       // [NC][WT]: Wt[k = tap*CU + cu][cv] (the packed down layout)
  float* Ds = Wt + C::NC * C::WT;                         // [CU][DH][DW] image with zero halo
  int* posoff = reinterpret_cast<int*>(Ds + ((CU * DH * DW + 3) & ~3));  // [P]: (2i)*DW + 2j
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  for (int i = tid; i < C::NC * CV; i += NT) {  // i = k * CV + cv, k = tap * CU + cu
    const int k = i / CV, cv = i % CV;
    Wt[k * C::WT + cv] = wref ? Wdown[(cv * CU + k % CU) * 16 + k / CU] : Wdown[i];
  }
  for (int p = tid; p < P; p += NT) posoff[p] = (2 * (p / w)) * DW + 2 * (p % w);
  const bool active = wave * WP < P;
  float bv0[C::NTV];
#pragma unroll
  for (int b = 0; b < C::NTV; ++b) bv0[b] = bias ? bias[b * 16 + l15] : 0.f;

  const int H2 = 2 * h, W2 = 2 * w;
  constexpr int ND = (CU * 34 * 34 + NT - 1) / NT;  // halo-tile elements per thread (h, w <= 16)
  const int nd = CU * DH * DW;
  float pu[ND];
  int hoff[ND];  // bits 0..27 offset into the image (clamped), 30 inside the image
#pragma unroll
  for (int u = 0; u < ND; ++u) {
    const int idx = tid + u * NT;
    const int idc = idx < nd ? idx : nd - 1;
    const int cu = idc / (DH * DW);
    const int rem = idc - cu * (DH * DW);
    const int y = rem / DW, x = rem - y * DW;
    const int oh = y - 1, ow = x - 1;
    const bool in = idx < nd && oh >= 0 && oh < H2 && ow >= 0 && ow < W2;
    const int ohc = oh < 0 ? 0 : (oh >= H2 ? H2 - 1 : oh), owc = ow < 0 ? 0 : (ow >= W2 ? W2 - 1 : ow);
    hoff[u] = ((cu * H2 + ohc) * W2 + owc) | ((int)in << 30);
  }
  auto prefetch = [&](long long img) __attribute__((always_inline)) {
    const float* src = U + img * CU * H2 * W2;
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const float a = src[hoff[u] & 0x0fffffff];  // unconditional clamped load, the halo is zeroed
      pu[u] = ((hoff[u] >> 30) & 1) ? a : 0.f;
    }
  };
  if ((long long)blockIdx.x < n) prefetch(blockIdx.x);
  for (long long img = blockIdx.x; img < n; img += gridDim.x) {
    __syncthreads();  // the previous image's tile is no longer read
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int idx = tid + u * NT;
      if (idx < nd) Ds[idx] = pu[u];
    }
    __syncthreads();
    if (img + gridDim.x < n) prefetch(img + gridDim.x);
    if (!active) continue;
    f32x4 acc[MT][C::NTV];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < C::NTV; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    int po[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a) po[a] = posoff[wave * WP + a * 16 + l15];
#pragma unroll 2
    for (int ks = 0; ks < C::NC / 4; ++ks) {
      const int k = ks * 4 + lq;
      const int tap = k / CU, cu = k - tap * CU;
      const int koff = cu * DH * DW + (tap >> 2) * DW + (tap & 3);
      float av[MT], bv[C::NTV];
#pragma unroll
      for (int a = 0; a < MT; ++a) av[a] = Ds[koff + po[a]];
#pragma unroll
      for (int b = 0; b < C::NTV; ++b) bv[b] = Wt[k * C::WT + b * 16 + l15];
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < C::NTV; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
    float* out = V + img * P * CV;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < C::NTV; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pos = wave * WP + a * 16 + lq * 4 + r, cv = b * 16 + l15;
          out[pos * CV + cv] = mvk_act(acc[a][b][r] + bv0[b], act);
        }
  }
}

// dWref += sum_b partial[b][0:CV*NC];  db += sum_b partial[b][CV*NC + cu];  db_v += sum_b partial[b][CV*NC+CU+cv]
__global__ __launch_bounds__(256) void small_up_bwd_reduce_kernel(const float* __restrict__ partial, int nblocks, int nw,
                                                                  int ncu, int ncv, float* __restrict__ dWref,
                                                                  float* __restrict__ db, float* __restrict__ db_v) {
  const int e = threadIdx.x & 31, zl = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + e;
  const int total = nw + ncu + ncv;
  float s = 0.f;
  if (i < total) {
#pragma unroll 4
    for (int b = zl; b < nblocks; b += 8) s += partial[(long long)b * total + i];
  }
  __shared__ float red[8][33];
  red[zl][e] = s;
  __syncthreads();
  if (zl != 0 || i >= total) return;
  s = ((red[0][e] + red[1][e]) + (red[2][e] + red[3][e])) + ((red[4][e] + red[5][e]) + (red[6][e] + red[7][e]));
  if (i < nw)
    dWref[i] += s;
  else if (i < nw + ncu) {
    if (db) db[i - nw] += s;
  } else if (db_v) {
    db_v[i - nw - ncu] += s;
  }
}

template <int CU, int CV>
static size_t fwd_lds(int P) {
  using C = SmallCfg<CU, CV>;
  const size_t tile = (size_t)P * (C::VS > C::CS ? C::VS : C::CS);
  return (CV * C::NC + tile + 4) * sizeof(float);  // + the zero word of the output gather
}
template <int CU, int CV>
static size_t bwd_lds(int h, int w) {
  using C = SmallCfg<CU, CV>;
  const int P = h * w, DH = 2 * h + 2, DW = 2 * w + 2;
  size_t fl = (size_t)C::NC * C::WT + ((CU * DH * DW + 3) & ~3) + (size_t)P * C::VS + P;
  const size_t red = (size_t)4 * CV * C::NC;
  if (fl < red) fl = red;
  return fl * sizeof(float);
}

static bool supported(int h, int w, int Cu, int Cv) {
  const int P = h * w;
  return Cu >= 1 && Cu <= 4 && (Cv == 16 || Cv == 32 || Cv == 64) && P % 64 == 0 && P <= 256;
}

template <int CU, int CV>
static int launch_fwd(const float* V, const float* Wref, const float* bias, float* U, int n, int h, int w, int act,
                      hipStream_t s, const float* X = nullptr, int xrows = 1, float scale = 1.f, float* rows = nullptr,
                      float grad_weight = 1.f, const float* v_amax = nullptr, float* du_amax = nullptr) {
  const size_t lds = fwd_lds<CU, CV>(h * w);
  constexpr int NT = MVK_SMALL_FWD_THREADS;
  if (lds > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(small_up_fwd_kernel<CU, CV, NT, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(small_up_fwd_kernel<CU, CV, NT, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const int grid = n < 512 ? n : 512;  // persistent: 2 workgroups per CU, each loops over images with prefetch
  // algorithmic bytes: the input map read once, the image written once
  // (+ the target images of the fused tail, read once per data point)
  mvk_prof_slot* prof = mvk::prof_next(5, 4.0 * n * h * w * (CV + 4.0 * CU) + (X ? 16.0 * xrows * h * w * CU : 0.0));
  // dense: 256 positions = one 16-row tile per wave, CV / 4 * 256 float4 and CU * 1024 outputs divide the workgroup evenly
  const bool dense = h == 16 && w == 16 && NT == 1024 && (256 * CV / 4) % NT == 0 && (CU * 1024) % NT == 0;
  // MVK_SMALL_FWD_BF=0: the exact-fp32 matrix instructions for every shape; =512: the split kernel with 512-thread workgroups
  static const int bf = mvk_tune("MVK_SMALL_FWD_BF") ? atoi(mvk_tune("MVK_SMALL_FWD_BF")) : 1024;
  if constexpr (CV == 32 && CU == 3) {
    if (h == 16 && w == 16 && bf > 0) {
      const size_t blds = 3 * (16 * CU) * 64 + (256 * (16 * CU + 1) + 24) * sizeof(float);  // + zero word + 16 wave partials
      const float inv_s2 = 1.f / (scale * scale), lconst = (float)(CU * 1024) * (logf(scale) + 0.91893853320467274178f);
      if (v_amax) {  // scaled fp16 pairs, weights as the A operand (small_up_fwd_h_kernel): 512-thread workgroups, 2 per CU
        const size_t hlds = 2 * (16 * CU) * 64 + (256 * (16 * CU + 4) + 48) * sizeof(float);  // + zero words (zidx, +16, +32), wave partials
        if (X)
          hipLaunchKernelGGL((small_up_fwd_h_kernel<CU, 512, true>), dim3(grid), dim3(512), hlds, s, V, Wref, bias, U, n, act, prof,
                             v_amax, X, xrows, inv_s2, lconst, rows, inv_s2 * grad_weight, du_amax);
        else
          hipLaunchKernelGGL((small_up_fwd_h_kernel<CU, 512, false>), dim3(grid), dim3(512), hlds, s, V, Wref, bias, U, n, act, prof,
                             v_amax);
        MVK_CHECK_LAUNCH();
        mvk::prof_fold(prof, s);
        return MVK_OK;
      }
      // fused tail: 512-thread workgroups (4 waves per SIMD, 128 registers) — at 1024 threads it spills; MVK_SMALL_NLL_NT=1024 for A/B
      static const int nll_nt = mvk_tune("MVK_SMALL_NLL_NT") ? atoi(mvk_tune("MVK_SMALL_NLL_NT")) : 512;
      if (X && nll_nt == 1024)
        hipLaunchKernelGGL((small_up_fwd_bf_kernel<CU, 1024, true>), dim3(grid), dim3(1024), blds, s, V, Wref, bias, U, n, act,
                           prof, X, xrows, inv_s2, lconst, rows, inv_s2 * grad_weight);
      else if (X)
        hipLaunchKernelGGL((small_up_fwd_bf_kernel<CU, 512, true>), dim3(grid), dim3(512), blds, s, V, Wref, bias, U, n, act,
                           prof, X, xrows, inv_s2, lconst, rows, inv_s2 * grad_weight);
      else if (bf == 512)
        hipLaunchKernelGGL((small_up_fwd_bf_kernel<CU, 512>), dim3(grid), dim3(512), blds, s, V, Wref, bias, U, n, act, prof);
      else
        hipLaunchKernelGGL((small_up_fwd_bf_kernel<CU, 1024>), dim3(grid), dim3(1024), blds, s, V, Wref, bias, U, n, act, prof);
      MVK_CHECK_LAUNCH();
      mvk::prof_fold(prof, s);
      return MVK_OK;
    }
  }
  if (X || v_amax || du_amax) return MVK_EINVAL;  // the fused tail / the scaled form: 16x16x32 -> 3 channels only (mvk_conv4s2_small_up_nll_supported)
  if (dense)
    hipLaunchKernelGGL((small_up_fwd_kernel<CU, CV, NT, true>), dim3(grid), dim3(NT), lds, s, V, Wref, bias, U, n, h, w, act, prof);
  else
    hipLaunchKernelGGL((small_up_fwd_kernel<CU, CV, NT, false>), dim3(grid), dim3(NT), lds, s, V, Wref, bias, U, n, h, w, act, prof);
  MVK_CHECK_LAUNCH();
  mvk::prof_fold(prof, s);
  return MVK_OK;
}

template <int CU, int CV>
static int launch_bwd(const float* dU, const float* Uout, int u_act, const float* V, int v_act, const float* Wref,
                      float* dV, float* dWref, float* db, float* db_v, float* ws, int64_t ws_floats, int n, int h,
                      int w, hipStream_t s, bool pre = false, const float* rowscale = nullptr, float* dv_amax = nullptr,
                      const float* du_amax = nullptr, const float* v_amax = nullptr) {
  using C = SmallCfg<CU, CV>;
  const int slab = CV * C::NC + CU + CV;
  // MVK_SMALL_BWD_UNITS=2: half-image work units for 16x16 inputs (3 workgroups per CU instead of 2).  Measured at
  // n = 5120: 142.7 us vs 146.0 us alone, 162-166 us vs 137-139 us inside the MoPoE step (same step time): off by default.
  static const int units_env = mvk_tune("MVK_SMALL_BWD_UNITS") ? atoi(mvk_tune("MVK_SMALL_BWD_UNITS")) : 1;
  const int units = (h == 16 && w == 16 && units_env == 2) ? 2 : 1;
  const long long nunits = (long long)n * units;
  // MVK_SMALL_BWD_BF=0 (read per call: tests switch it): the exact-fp32 kernel for every shape.  Default: the split-bf16 kernel at
  // the SVHN decoder's shape — 128 -> 110 us alone at n = 5120, -13 us per training step (three A/B pairs on one box).
  const char* bf_str = mvk_tune("MVK_SMALL_BWD_BF");
  const bool bf = (!bf_str || atoi(bf_str) != 0) && CU == 3 && CV == 32 && h == 16 && w == 16 && units_env != 2 &&
                  (pre || u_act == MVK_ACT_SIGMOID) && v_act == MVK_ACT_RELU && mvk_aligned16(dU) &&
                  (pre || mvk_aligned16(Uout)) && mvk_aligned16(V) && mvk_aligned16(dV) && mvk_aligned16(Wref);
  if ((pre || dv_amax) && !bf) return MVK_EINVAL;  // the pre-activation form and the published maximum: split-bf16 kernel only
  if ((du_amax || v_amax) && !(pre && du_amax && v_amax)) return MVK_EINVAL;  // the scaled-fp16 form: both bounds, pre-activation form
  const int gmax = units == 2 ? 1024 : 512;
  int grid = nunits < gmax ? (int)nunits : gmax;
  float* dslab = (mvk::defer_free(db) && mvk::defer_free(db_v)) ? mvk::defer_scratch(dWref, (long long)grid * slab, s) : nullptr;
  if (dslab) ws = dslab;
  else if ((int64_t)grid * slab > ws_floats) grid = (int)(ws_floats / slab);
  if (grid < 1) return MVK_EINVAL;
  if constexpr (CU == 3 && CV == 32) {
    if (bf) {
      mvk_prof_slot* prof = mvk::prof_next(6, 4.0 * n * h * w * (2.0 * CV + 8.0 * CU));
      constexpr int BLDS = 3 * 4 * 289 * 8 + 3 * 256 * 64;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(small_up_bwd_bf_kernel<CU, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, BLDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(small_up_bwd_bf_kernel<CU, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, BLDS);
      if (du_amax) {
        constexpr int HLDS = 2 * 4 * 289 * 8 + 2 * 256 * 64;
        hipLaunchKernelGGL((small_up_bwd_h_kernel<CU>), dim3(grid), dim3(256), HLDS, s, dU, V, Wref, dV, ws, n, prof, rowscale,
                           du_amax, v_amax, dv_amax);
      } else if (pre)
        hipLaunchKernelGGL((small_up_bwd_bf_kernel<CU, true>), dim3(grid), dim3(256), BLDS, s, dU, Uout, V, Wref, dV, ws, n, prof,
                           rowscale, dv_amax);
      else
        hipLaunchKernelGGL((small_up_bwd_bf_kernel<CU, false>), dim3(grid), dim3(256), BLDS, s, dU, Uout, V, Wref, dV, ws, n, prof,
                           rowscale, dv_amax);
      MVK_CHECK_LAUNCH();
      mvk::prof_fold(prof, s);
      const int total = CV * C::NC + CU + CV;
      if (dslab) {
        int rc = mvk::defer_push_plain(dWref, dslab, CV * C::NC, grid, total, s);
        if (rc == MVK_OK && db) rc = mvk::defer_push_plain(db, dslab + CV * C::NC, CU, grid, total, s);
        if (rc == MVK_OK && db_v) rc = mvk::defer_push_plain(db_v, dslab + CV * C::NC + CU, CV, grid, total, s);
        return rc;
      }
      hipLaunchKernelGGL(small_up_bwd_reduce_kernel, dim3((total + 31) / 32), dim3(256), 0, s, ws, grid, CV * C::NC, CU, CV,
                         dWref, db, db_v);
      MVK_CHECK_LAUNCH();
      return MVK_OK;
    }
  }
  const size_t lds = bwd_lds<CU, CV>(h / units, w);
  constexpr int NT = MVK_SMALL_BWD_THREADS;
  if (lds > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(small_up_bwd_kernel<CU, CV, NT, 256, 2, -1, -1, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(
        reinterpret_cast<const void*>(small_up_bwd_kernel<CU, CV, NT, 256, 2, MVK_ACT_SIGMOID, MVK_ACT_RELU, false>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(small_up_bwd_kernel<CU, CV, NT, 256, 2, -1, -1, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(
        reinterpret_cast<const void*>(small_up_bwd_kernel<CU, CV, NT, 256, 2, MVK_ACT_SIGMOID, MVK_ACT_RELU, true>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  // algorithmic bytes: image gradient + image read once, the saved input map read once, its gradient written once
  mvk_prof_slot* prof = mvk::prof_next(6, 4.0 * n * h * w * (2.0 * CV + 8.0 * CU));
  static const int occ_env = mvk_tune("MVK_SMALL_BWD_OCC") ? atoi(mvk_tune("MVK_SMALL_BWD_OCC")) : 3;
  const bool spec = u_act == MVK_ACT_SIGMOID && v_act == MVK_ACT_RELU;
#ifdef MVK_ABLATE
  const int abl_bits = ((mvk_tune("MVK_ABLATE") ? atoi(mvk_tune("MVK_ABLATE")) : 0) << 8) |
                       ((mvk_tune("MVK_DEPHASE") ? atoi(mvk_tune("MVK_DEPHASE")) : 0) << 16);
#else
  constexpr int abl_bits = 0;
#endif
#define MVK_SUB_LAUNCH(PU_, OCC_, UA_, VA_, UNITS_, DENSE_)                                                               \
  hipLaunchKernelGGL((small_up_bwd_kernel<CU, CV, NT, PU_, OCC_, UA_, VA_, DENSE_>), dim3(grid), dim3(NT), lds, s, dU, Uout, \
                     u_act, V, v_act, Wref, dV, ws, n, h, w, (UNITS_) | abl_bits, prof)
  static const int dense_env = mvk_tune("MVK_SMALL_BWD_DENSE") ? atoi(mvk_tune("MVK_SMALL_BWD_DENSE")) : 1;  // A/B switch
  const bool dense = dense_env && h == 16 && w == 16 && NT == 256 && mvk_aligned16(dU) && mvk_aligned16(Uout);
  if (units == 2 && occ_env == 4) {
    if (spec) MVK_SUB_LAUNCH(128, 4, MVK_ACT_SIGMOID, MVK_ACT_RELU, 2, false);
    else MVK_SUB_LAUNCH(128, 4, -1, -1, 2, false);
  } else if (units == 2) {
    if (spec) MVK_SUB_LAUNCH(128, 3, MVK_ACT_SIGMOID, MVK_ACT_RELU, 2, false);
    else MVK_SUB_LAUNCH(128, 3, -1, -1, 2, false);
  } else if (dense) {
    if (spec) MVK_SUB_LAUNCH(256, 2, MVK_ACT_SIGMOID, MVK_ACT_RELU, 1, true);
    else MVK_SUB_LAUNCH(256, 2, -1, -1, 1, true);
  } else {
    if (spec) MVK_SUB_LAUNCH(256, 2, MVK_ACT_SIGMOID, MVK_ACT_RELU, 1, false);
    else MVK_SUB_LAUNCH(256, 2, -1, -1, 1, false);
  }
#undef MVK_SUB_LAUNCH
  MVK_CHECK_LAUNCH();
  mvk::prof_fold(prof, s);
  const int total = CV * C::NC + CU + CV;
  if (dslab) {  // the three ordered sums over the per-workgroup slabs are queued (mvk_defer_flush)
    int rc = mvk::defer_push_plain(dWref, dslab, CV * C::NC, grid, total, s);
    if (rc == MVK_OK && db) rc = mvk::defer_push_plain(db, dslab + CV * C::NC, CU, grid, total, s);
    if (rc == MVK_OK && db_v) rc = mvk::defer_push_plain(db_v, dslab + CV * C::NC + CU, CV, grid, total, s);
    return rc;
  }
  hipLaunchKernelGGL(small_up_bwd_reduce_kernel, dim3((total + 31) / 32), dim3(256), 0, s, ws, grid, CV * C::NC, CU, CV,
                     dWref, db, db_v);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

template <int CU, int CV>
static int launch_down_fwd(const float* U, const float* Wdown, const float* bias, float* V, int n, int h, int w, int act,
                           hipStream_t s, int wref) {
  using C = SmallCfg<CU, CV>;
  constexpr int NT = MVK_SMALL_DOWN_THREADS;
  const int P = h * w, DH = 2 * h + 2, DW = 2 * w + 2;
  const size_t lds = ((size_t)C::NC * C::WT + ((CU * DH * DW + 3) & ~3) + P) * sizeof(float);
  const int grid = n < 1024 ? n : 1024;  // persistent: up to 4 workgroups per CU, each loops over images with prefetch
  hipLaunchKernelGGL((small_down_fwd_kernel<CU, CV, NT>), dim3(grid), dim3(NT), lds, s, U, Wdown, bias, V, n, h, w, act, wref);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

#define MVK_SMALL_DISPATCH(FN, ...)                                  \
  switch (Cu * 100 + Cv) {                                           \
    case 116: return FN<1, 16>(__VA_ARGS__);                         \
    case 132: return FN<1, 32>(__VA_ARGS__);                         \
    case 164: return FN<1, 64>(__VA_ARGS__);                         \
    case 216: return FN<2, 16>(__VA_ARGS__);                         \
    case 232: return FN<2, 32>(__VA_ARGS__);                         \
    case 264: return FN<2, 64>(__VA_ARGS__);                         \
    case 316: return FN<3, 16>(__VA_ARGS__);                         \
    case 332: return FN<3, 32>(__VA_ARGS__);                         \
    case 364: return FN<3, 64>(__VA_ARGS__);                         \
    case 416: return FN<4, 16>(__VA_ARGS__);                         \
    case 432: return FN<4, 32>(__VA_ARGS__);                         \
    case 464: return FN<4, 64>(__VA_ARGS__);                         \
    default: return MVK_EINVAL;                                      \
  }

}  // namespace

extern "C" {

int mvk_conv4s2_small_up_supported(int h, int w, int Cu, int Cv) { return supported(h, w, Cu, Cv) ? 1 : 0; }

int mvk_conv4s2_small_up_fwd(const float* V, const float* Wref, const float* bias, float* U, int n, int h, int w,
                             int Cu, int Cv, int act, void* stream) {
  if (!V || !Wref || !U || n < 0 || !supported(h, w, Cu, Cv) || !mvk_aligned16(V)) return MVK_EINVAL;
  if (n == 0) return MVK_OK;
  hipStream_t s = mvk_stream(stream);
  MVK_SMALL_DISPATCH(launch_fwd, V, Wref, bias, U, n, h, w, act, s)
}

/* The fused decoder tail (round 3): the last layer of Decoder_VAE_SVHN (models/nn/svhn.py:52-70) scored against the data by a
 * Normal(scale) likelihood in its own epilogue.  rows[n] = -log p(x | image) summed over the image (the row sums
 * mvk_recon_nll_fwd produces), dpre = d rows / d pre-activation, stored where the image would be.  X: [xrows][Cu * 4 h w]; image i is
 * scored against X[i % xrows].  Only where mvk_conv4s2_small_up_nll_supported says so (16x16 x 32 channels -> 3 x 32x32). */
int mvk_conv4s2_small_up_nll_supported(int h, int w, int Cu, int Cv) { return h == 16 && w == 16 && Cu == 3 && Cv == 32 ? 1 : 0; }

int mvk_conv4s2_small_up_fwd_nll_w(const float* V, const float* Wref, const float* bias, const float* X, int xrows, float scale,
                                   float grad_weight, float* dpre, float* rows, int n, int h, int w, int Cu, int Cv, int act,
                                   void* stream) {
  if (!V || !Wref || !X || !dpre || !rows || n < 0 || xrows <= 0 || !(scale > 0.f) || !mvk_aligned16(V) ||
      !mvk_conv4s2_small_up_nll_supported(h, w, Cu, Cv))
    return MVK_EINVAL;
  if (n == 0) return MVK_OK;
  return launch_fwd<3, 32>(V, Wref, bias, dpre, n, h, w, act, mvk_stream(stream), X, xrows, scale, rows, grad_weight);
}

/* The same two launches on scaled fp16 pairs (3 MFMAs per product, bf3.hpp): v_amax = device scalar bounding max |V| (published by
 * the launch that produced V: amax protocol); where mvk_conv4s2_small_up_nll_supported says so. */
int mvk_conv4s2_small_up_fwd_nll_s(const float* V, const float* Wref, const float* bias, const float* X, int xrows, float scale,
                                   float grad_weight, float* dpre, float* rows, int n, int h, int w, int Cu, int Cv, int act,
                                   const float* v_amax, void* stream) {
  if (!V || !Wref || !X || !dpre || !rows || !v_amax || n < 0 || xrows <= 0 || !(scale > 0.f) || !mvk_aligned16(V) ||
      !mvk_conv4s2_small_up_nll_supported(h, w, Cu, Cv))
    return MVK_EINVAL;
  if (n == 0) return MVK_OK;
  return launch_fwd<3, 32>(V, Wref, bias, dpre, n, h, w, act, mvk_stream(stream), X, xrows, scale, rows, grad_weight, v_amax);
}

/* mvk_conv4s2_small_up_fwd_nll_s that also publishes an upper bound of max |dpre| (amax protocol: *dpre_amax must hold 0 before
 * the launch): what mvk_conv4s2_small_up_bwd_pre_s scales the gradient image by. */
int mvk_conv4s2_small_up_fwd_nll_sy(const float* V, const float* Wref, const float* bias, const float* X, int xrows, float scale,
                                    float grad_weight, float* dpre, float* rows, int n, int h, int w, int Cu, int Cv, int act,
                                    const float* v_amax, float* dpre_amax, void* stream) {
  if (!V || !Wref || !X || !dpre || !rows || !v_amax || !dpre_amax || n < 0 || xrows <= 0 || !(scale > 0.f) || !mvk_aligned16(V) ||
      !mvk_conv4s2_small_up_nll_supported(h, w, Cu, Cv))
    return MVK_EINVAL;
  if (n == 0) return MVK_OK;
  return launch_fwd<3, 32>(V, Wref, bias, dpre, n, h, w, act, mvk_stream(stream), X, xrows, scale, rows, grad_weight, v_amax,
                           dpre_amax);
}

/* mvk_conv4s2_small_up_bwd_pre_y on scaled fp16 pairs (small_up_bwd_h_kernel): dpre_amax bounds max |dpre| (published by
 * mvk_conv4s2_small_up_fwd_nll_sy), v_amax bounds max |V| (published by the launch that produced V); dv_amax may be NULL. */
int mvk_conv4s2_small_up_bwd_pre_s(const float* dpre, const float* rowscale, const float* V, int v_act, const float* Wref,
                                   float* dV, float* dWref, float* db, float* db_v, float* ws, int64_t ws_floats, int n, int h,
                                   int w, int Cu, int Cv, const float* dpre_amax, const float* v_amax, float* dv_amax,
                                   void* stream) {
  if (!dpre || !V || !Wref || !dV || !dWref || !ws || !dpre_amax || !v_amax || n < 0 || !mvk_aligned16(V) ||
      !mvk_conv4s2_small_up_nll_supported(h, w, Cu, Cv))
    return MVK_EINVAL;
  if (n == 0) return MVK_OK;
  return launch_bwd<3, 32>(dpre, nullptr, MVK_ACT_NONE, V, v_act, Wref, dV, dWref, db, db_v, ws, ws_floats, n, h, w,
                           mvk_stream(stream), true, rowscale, dv_amax, dpre_amax, v_amax);
}

int mvk_conv4s2_small_up_fwd_s(const float* V, const float* Wref, const float* bias, float* U, int n, int h, int w, int Cu, int Cv,
                               int act, const float* v_amax, void* stream) {
  if (!V || !Wref || !U || !v_amax || n < 0 || !mvk_aligned16(V) || !mvk_conv4s2_small_up_nll_supported(h, w, Cu, Cv))
    return MVK_EINVAL;
  if (n == 0) return MVK_OK;
  return launch_fwd<3, 32>(V, Wref, bias, U, n, h, w, act, mvk_stream(stream), nullptr, 1, 1.f, nullptr, 1.f, v_amax);
}

int mvk_conv4s2_small_up_fwd_nll(const float* V, const float* Wref, const float* bias, const float* X, int xrows, float scale,
                                 float* dpre, float* rows, int n, int h, int w, int Cu, int Cv, int act, void* stream) {
  return mvk_conv4s2_small_up_fwd_nll_w(V, Wref, bias, X, xrows, scale, 1.f, dpre, rows, n, h, w, Cu, Cv, act, stream);
}

/* Its backward: dpre (from mvk_conv4s2_small_up_fwd_nll) times rowscale[n] (d loss / d rows; NULL = 1) through the layer. */
int mvk_conv4s2_small_up_bwd_pre(const float* dpre, const float* rowscale, const float* V, int v_act, const float* Wref, float* dV,
                                 float* dWref, float* db, float* db_v, float* ws, int64_t ws_floats, int n, int h, int w, int Cu,
                                 int Cv, void* stream) {
  if (!dpre || !V || !Wref || !dV || !dWref || !ws || n < 0 || !mvk_aligned16(V) ||
      !mvk_conv4s2_small_up_nll_supported(h, w, Cu, Cv))
    return MVK_EINVAL;
  if (n == 0) return MVK_OK;
  return launch_bwd<3, 32>(dpre, nullptr, MVK_ACT_NONE, V, v_act, Wref, dV, dWref, db, db_v, ws, ws_floats, n, h, w,
                           mvk_stream(stream), true, rowscale);
}

// mvk_conv4s2_small_up_bwd_pre with the published maximum of dV (amax protocol: dv_amax must hold 0 before the launch)
int mvk_conv4s2_small_up_bwd_pre_y(const float* dpre, const float* rowscale, const float* V, int v_act, const float* Wref,
                                   float* dV, float* dWref, float* db, float* db_v, float* ws, int64_t ws_floats, int n, int h,
                                   int w, int Cu, int Cv, float* dv_amax, void* stream) {
  if (!dpre || !V || !Wref || !dV || !dWref || !ws || !dv_amax || n < 0 || !mvk_aligned16(V) ||
      !mvk_conv4s2_small_up_nll_supported(h, w, Cu, Cv))
    return MVK_EINVAL;
  if (n == 0) return MVK_OK;
  return launch_bwd<3, 32>(dpre, nullptr, MVK_ACT_NONE, V, v_act, Wref, dV, dWref, db, db_v, ws, ws_floats, n, h, w,
                           mvk_stream(stream), true, rowscale, dv_amax);
}

int mvk_conv4s2_small_down_fwd(const float* U, const float* Wdown, const float* bias, float* V, int n, int h, int w,
                               int Cu, int Cv, int act, void* stream) {
  if (n == 0) return MVK_OK;
  if (!U || !Wdown || !V || n < 0 || !supported(h, w, Cu, Cv)) return MVK_EINVAL;
  hipStream_t s = mvk_stream(stream);
  MVK_SMALL_DISPATCH(launch_down_fwd, U, Wdown, bias, V, n, h, w, act, s, 0)
}

/* the same layer from the REFERENCE weight layout Wref[Cv][Cu][4][4] (no weight pack in front of the encoder's first kernel) */
int mvk_conv4s2_small_down_fwd_wref(const float* U, const float* Wref, const float* bias, float* V, int n, int h, int w,
                                    int Cu, int Cv, int act, void* stream) {
  if (n == 0) return MVK_OK;
  if (!U || !Wref || !V || n < 0 || !supported(h, w, Cu, Cv)) return MVK_EINVAL;
  hipStream_t s = mvk_stream(stream);
  MVK_SMALL_DISPATCH(launch_down_fwd, U, Wref, bias, V, n, h, w, act, s, 1)
}

int mvk_conv4s2_small_up_bwd(const float* dU, const float* Uout, int u_act, const float* V, int v_act,
                             const float* Wref, float* dV, float* dWref, float* db, float* db_v, float* ws,
                             int64_t ws_floats, int n, int h, int w, int Cu, int Cv, void* stream) {
  if (!dU || !Uout || !V || !Wref || !dV || !dWref || !ws || n < 0 || !supported(h, w, Cu, Cv) || !mvk_aligned16(V))
    return MVK_EINVAL;
  if (n == 0) return MVK_OK;
  hipStream_t s = mvk_stream(stream);
  MVK_SMALL_DISPATCH(launch_bwd, dU, Uout, u_act, V, v_act, Wref, dV, dWref, db, db_v, ws, ws_floats, n, h, w, s)
}

}  // extern "C"
