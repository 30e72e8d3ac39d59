// Fused ELBO kernels: posterior aggregation (PoE / MoPoE) + reparameterisation + Gaussian KL, and the
// HBM-bound reconstruction-NLL kernel over the K-sample axis with the gradient emitted in the same pass.
#include "common.hpp"
#include <cstdlib>

namespace {

constexpr int MAXM = MVK_MAX_MODALITIES;
constexpr float POE_EPS = 1e-8f;  // models/base/base_utils.py:122 default eps

struct PtrTable {
  const float* mu[MAXM];
  const float* lv[MAXM];
  const uint8_t* mask[MAXM];
};
struct OutPtrTable {
  float* dmu[MAXM];
  float* dlv[MAXM];
};

// ---------------------------------------------------------------------------------------------------------
// MoPoE posterior.  One wave per batch row; lanes stride over the latent dimension.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mopoe_posterior_fwd_kernel(
    const PtrTable pt, int M, const int32_t* __restrict__ subset_masks, int S, const int32_t* __restrict__ sel,
    const float* __restrict__ weights, const float* __restrict__ eps, int K, int B, int L, float* __restrict__ z,
    float* __restrict__ kld_rows, float* __restrict__ mus_out, float* __restrict__ lvs_out,
    float* __restrict__ joint_mu, float* __restrict__ joint_lv, mvk_prof_slot* prof) {
  mvk_prof_begin(prof);
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int my_sel = sel[b];
  const uint32_t full = (M >= 32) ? 0xffffffffu : ((1u << M) - 1u);
  float kld_acc = 0.f;
  for (int l = lane; l < L; l += 64) {
    const long long o = (long long)b * L + l;
    float T[MAXM], muT[MAXM];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      if (m < M) {
        float mu = pt.mu[m][o];
        float var = expf(pt.lv[m][o]) + POE_EPS;
        T[m] = 1.0f / var;
        muT[m] = mu * T[m];
      } else {
        T[m] = 0.f;
        muT[m] = 0.f;
      }
    }
    for (int s = 0; s < S; ++s) {
      const uint32_t bits = (uint32_t)subset_masks[s];
      float D = 0.f, N = 0.f;
      bool first = true;
#pragma unroll
      for (int m = 0; m < MAXM; ++m) {
        if (m < M && ((bits >> m) & 1u)) {
          // sequential sum over the stacked experts, as torch.sum(dim=0) does for a short leading dim
          D = first ? T[m] : D + T[m];
          N = first ? muT[m] : N + muT[m];
          first = false;
        }
      }
      if (bits == full) {  // N(0, I) prior expert: var = exp(0) + 1e-8 == 1.0f in fp32
        D += 1.0f / (1.0f + POE_EPS);
        N += 0.0f;
      }
      const float mu_s = N / D;
      const float var_s = 1.0f / D;
      const float lv_s = logf(var_s);
      const float w = weights ? weights[(long long)s * B + b] : 1.0f / (float)S;
      const float kl = -0.5f * (1.0f - expf(lv_s) - mu_s * mu_s + lv_s);
      kld_acc += w * kl;
      if (mus_out) {
        mus_out[((long long)s * B + b) * L + l] = mu_s;
        lvs_out[((long long)s * B + b) * L + l] = lv_s;
      }
      if (s == my_sel) {
        const float sd = expf(0.5f * lv_s);
        for (int k = 0; k < K; ++k) {
          const long long zo = ((long long)k * B + b) * L + l;
          z[zo] = mu_s + sd * eps[zo];
        }
        if (joint_mu) {
          joint_mu[o] = mu_s;
          joint_lv[o] = lv_s;
        }
      }
    }
  }
  kld_acc = wave_sum(kld_acc);
  if (lane == 0) kld_rows[b] = kld_acc;
  mvk_prof_end_wave(prof);
}

__global__ __launch_bounds__(256) void mopoe_posterior_bwd_kernel(
    const PtrTable pt, const OutPtrTable ot, int M, const int32_t* __restrict__ subset_masks, int S,
    const int32_t* __restrict__ sel, const float* __restrict__ weights, const float* __restrict__ eps,
    const float* __restrict__ dz, int K, int B, int L, const float* __restrict__ gkld_rows, mvk_prof_slot* prof) {
  mvk_prof_begin(prof);
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int my_sel = sel[b];
  const float g = gkld_rows ? gkld_rows[b] : 0.0f;
  const uint32_t full = (M >= 32) ? 0xffffffffu : ((1u << M) - 1u);
  for (int l = lane; l < L; l += 64) {
    const long long o = (long long)b * L + l;
    float T[MAXM], mu[MAXM], ev[MAXM], dmu[MAXM], dlv[MAXM];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      dmu[m] = 0.f;
      dlv[m] = 0.f;
      if (m < M) {
        mu[m] = pt.mu[m][o];
        ev[m] = expf(pt.lv[m][o]);
        T[m] = 1.0f / (ev[m] + POE_EPS);
      } else {
        mu[m] = 0.f;
        ev[m] = 0.f;
        T[m] = 0.f;
      }
    }
    // sum over k of dz and dz*eps for the selected subset
    float sdz = 0.f, sdze = 0.f;
    {  // five samples' loads in flight at a time (one at a time = K dependent memory latencies at the head of the encoders'
       // backward chain); the sums run in k order as before: bit-identical
      const long long ks = (long long)B * L;
      long long zo = (long long)b * L + l;
      int k = 0;
      for (; k + 5 <= K; k += 5, zo += 5 * ks) {
        float d[5], e[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          d[j] = dz[zo + j * ks];
          e[j] = eps[zo + j * ks];
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          sdz += d[j];
          sdze += d[j] * e[j];
        }
      }
      for (; k < K; ++k, zo += ks) {
        const float d = dz[zo];
        sdz += d;
        sdze += d * eps[zo];
      }
    }
    for (int s = 0; s < S; ++s) {
      const uint32_t bits = (uint32_t)subset_masks[s];
      float D = 0.f, N = 0.f;
#pragma unroll
      for (int m = 0; m < MAXM; ++m) {
        if (m < M && ((bits >> m) & 1u)) {
          D += T[m];
          N += mu[m] * T[m];
        }
      }
      if (bits == full) D += 1.0f / (1.0f + POE_EPS);
      const float mu_s = N / D;
      const float lv_s = logf(1.0f / D);
      const float w = weights ? weights[(long long)s * B + b] : 1.0f / (float)S;
      const float c = g * w;
      float g_mu = c * mu_s;
      float g_lv = c * 0.5f * (expf(lv_s) - 1.0f);
      if (s == my_sel) {
        g_mu += sdz;
        g_lv += 0.5f * expf(0.5f * lv_s) * sdze;
      }
      // mu_s = N/D ; lv_s = -log D
      const float dN = g_mu / D;
      const float dD = -(g_mu * mu_s + g_lv) / D;
#pragma unroll
      for (int m = 0; m < MAXM; ++m) {
        if (m < M && ((bits >> m) & 1u)) {
          dmu[m] += dN * T[m];
          const float dT = dN * mu[m] + dD;
          dlv[m] += dT * (-T[m] * T[m] * ev[m]);
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      if (m < M) {
        ot.dmu[m][o] = dmu[m];
        ot.dlv[m][o] = dlv[m];
      }
    }
  }
  mvk_prof_end_wave(prof);
}

// ---------------------------------------------------------------------------------------------------------
// MVTCAE posterior.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mvtcae_posterior_fwd_kernel(const PtrTable pt, int M,
                                                                   const float* __restrict__ eps, int K, int B,
                                                                   int L, float* __restrict__ z,
                                                                   float* __restrict__ joint_kl_rows,
                                                                   float* __restrict__ cond_kl_rows,
                                                                   float* __restrict__ joint_mu,
                                                                   float* __restrict__ joint_lv) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  bool avail[MAXM];
#pragma unroll
  for (int m = 0; m < MAXM; ++m) avail[m] = (m < M) && (pt.mask[m] ? pt.mask[m][b] != 0 : true);
  float jkl = 0.f;
  float ckl[MAXM];
#pragma unroll
  for (int m = 0; m < MAXM; ++m) ckl[m] = 0.f;
  for (int l = lane; l < L; l += 64) {
    const long long o = (long long)b * L + l;
    float mu[MAXM], lv[MAXM];
    float D = 0.f, N = 0.f;
    bool first = true;
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      mu[m] = 0.f;
      lv[m] = 0.f;
      if (m < M) {
        mu[m] = pt.mu[m][o];
        lv[m] = pt.lv[m][o];
        // missing rows: log-variance = +inf -> T = 0 (mvtcae_model.py:128-129)
        const float T = avail[m] ? 1.0f / (expf(lv[m]) + POE_EPS) : 0.0f;
        const float mt = mu[m] * T;
        D = first ? T : D + T;
        N = first ? mt : N + mt;
        first = false;
      }
    }
    const float jmu = N / D;
    const float jlv = logf(1.0f / D);
    const float ejl = expf(jlv);
    jkl += -0.5f * (1.0f - ejl - jmu * jmu + jlv);
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      if (m < M && avail[m]) {
        const float em = expf(lv[m]);
        const float dm = jmu - mu[m];
        ckl[m] += -0.5f * (1.0f - ejl / em - dm * dm / em + jlv - lv[m]);
      }
    }
    const float sd = expf(0.5f * jlv);
    for (int k = 0; k < K; ++k) {
      const long long zo = ((long long)k * B + b) * L + l;
      z[zo] = jmu + sd * eps[zo];
    }
    if (joint_mu) {
      joint_mu[o] = jmu;
      joint_lv[o] = jlv;
    }
  }
  jkl = wave_sum(jkl);
  if (lane == 0) joint_kl_rows[b] = jkl;
#pragma unroll
  for (int m = 0; m < MAXM; ++m) {
    if (m < M) {
      float v = wave_sum(ckl[m]);
      if (lane == 0) cond_kl_rows[(long long)m * B + b] = v;
    }
  }
}

__global__ __launch_bounds__(256) void mvtcae_posterior_bwd_kernel(const PtrTable pt, const OutPtrTable ot, int M,
                                                                   const float* __restrict__ eps,
                                                                   const float* __restrict__ dz, int K, int B, int L,
                                                                   const float* __restrict__ gjoint_rows,
                                                                   const float* __restrict__ gcond_rows) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float jc = gjoint_rows ? gjoint_rows[b] : 0.0f;
  bool avail[MAXM];
#pragma unroll
  for (int m = 0; m < MAXM; ++m) avail[m] = (m < M) && (pt.mask[m] ? pt.mask[m][b] != 0 : true);
  for (int l = lane; l < L; l += 64) {
    const long long o = (long long)b * L + l;
    float mu[MAXM], lv[MAXM], T[MAXM], ev[MAXM];
    float D = 0.f, N = 0.f;
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      mu[m] = lv[m] = T[m] = ev[m] = 0.f;
      if (m < M) {
        mu[m] = pt.mu[m][o];
        lv[m] = pt.lv[m][o];
        ev[m] = expf(lv[m]);
        T[m] = avail[m] ? 1.0f / (ev[m] + POE_EPS) : 0.0f;
        D += T[m];
        N += mu[m] * T[m];
      }
    }
    const float jmu = N / D;
    const float jlv = logf(1.0f / D);
    const float ejl = expf(jlv);
    float sdz = 0.f, sdze = 0.f;
    for (int k = 0; k < K; ++k) {
      const long long zo = ((long long)k * B + b) * L + l;
      const float d = dz[zo];
      sdz += d;
      sdze += d * eps[zo];
    }
    float g_mu = sdz + jc * jmu;
    float g_lv = 0.5f * expf(0.5f * jlv) * sdze + jc * 0.5f * (ejl - 1.0f);
    float dmu[MAXM], dlv[MAXM];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      dmu[m] = dlv[m] = 0.f;
      if (m < M && avail[m]) {
        const float cc = gcond_rows ? gcond_rows[(long long)m * B + b] : 0.0f;
        const float dm = jmu - mu[m];
        const float inv = 1.0f / ev[m];
        g_mu += cc * dm * inv;
        g_lv += cc * 0.5f * (ejl * inv - 1.0f);
        dmu[m] = -cc * dm * inv;
        dlv[m] = -cc * 0.5f * (ejl * inv + dm * dm * inv - 1.0f);
      }
    }
    const float dN = g_mu / D;
    const float dD = -(g_mu * jmu + g_lv) / D;
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      if (m < M) {
        if (avail[m]) {
          dmu[m] += dN * T[m];
          const float dT = dN * mu[m] + dD;
          dlv[m] += dT * (-T[m] * T[m] * ev[m]);
        }
        ot.dmu[m][o] = dmu[m];
        ot.dlv[m][o] = dlv[m];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Reconstruction NLL over [K,B,D] with the gradient emitted in the same pass.  HBM-bound:
// algorithmic bytes = 4*B*D*(2K+1) per modality (read recon, write d_recon, read x once).
// One block = one batch row b and a chunk of KC samples; x[b,:] lives in registers across the chunk.
// ---------------------------------------------------------------------------------------------------------
constexpr int KC = 16;         // max samples per block (LDS partial sums)
constexpr int NLL_THREADS = 256;
constexpr int NLL_NV = 3;      // float4 per thread and column tile on the long-row vector path (SVHN: 3072 = 3 * 1024)

struct ReconTable {
  mvk_recon_desc d[MAXM];
  int block_start[MAXM + 1];  // prefix sum of blocks per modality
  int kchunks[MAXM];          // sample chunks per batch row (large rows are cut finer: load balance, see launch_recon)
  int n;
  mvk_prof_slot* prof;        // device-timestamp record of this launch (null: profiler off)
};

__device__ __forceinline__ void nll_elem(int dist, float inv_s, float inv_s2, float r, float x, float& nll,
                                         float& dr) {
  if (dist == MVK_DIST_NORMAL) {
    const float d = r - x;
    nll = 0.5f * d * d * inv_s2;  // + log(scale) + 0.5 log(2 pi), added per row
    dr = d * inv_s2;
  } else if (dist == MVK_DIST_LAPLACE) {
    const float d = r - x;
    nll = fabsf(d) * inv_s;  // + log(2 scale) per row
    dr = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * inv_s;
  } else {  // Bernoulli with logits r: BCE-with-logits = max(r,0) - r x + log1p(exp(-|r|))
    const float e = expf(-fabsf(r));
    nll = fmaxf(r, 0.f) - r * x + log1pf(e);
    const float sig = r >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
    dr = sig - x;
  }
}

__device__ __forceinline__ float nll_row_const(int dist, float scale, long long D) {
  if (dist == MVK_DIST_NORMAL) return (float)D * (logf(scale) + 0.918938533204672742f);
  if (dist == MVK_DIST_LAPLACE) return (float)D * logf(2.0f * scale);
  return 0.f;
}

// Vector path of one block: row b, samples k0 .. k0+kn-1.  The row is walked in column tiles of NV * 256 float4; the
// x tile stays in registers while the tile of every sample streams past it, and the 16-byte loads of sample k+1 are
// issued before the arithmetic of sample k (two register stages).  NV is a compile-time constant per launch group
// (1: rows up to 1024 values, 3: longer rows -- SVHN's 3072 values are exactly one tile), so there are no dummy loads
// and the register count stays low enough for 8 waves per SIMD; partial sums go to LDS per (sample, wave), so the
// sample loop is a plain loop.  Loads are unconditional (lanes past the row end re-read element 0 and are masked in
// the arithmetic): a predicated HIP float4 load is scalarised into four branchy dword loads.
typedef float nll_f32x4 __attribute__((ext_vector_type(4)));

template <int DIST, int NV>
__device__ __forceinline__ void recon_vec_body(const mvk_recon_desc& d, int B, int b, int k0, int kn, float gbase,
                                               float inv_s, float inv_s2, const float* __restrict__ xrow,
                                               float (*red)[NLL_THREADS / 64]) {
  const long long D = d.D;
  const int nv = (int)(D >> 2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t0 = 0; t0 < nv; t0 += NV * NLL_THREADS) {
    nll_f32x4 xv[NV], cur[NV], nxt[NV];
    bool live[NV];
    int off[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int idx = t0 + threadIdx.x + j * NLL_THREADS;
      live[j] = idx < nv;
      off[j] = live[j] ? idx : 0;
      xv[j] = reinterpret_cast<const nll_f32x4*>(xrow)[off[j]];
    }
    auto load_sample = [&](int k, nll_f32x4 (&dst)[NV]) __attribute__((always_inline)) {
      const nll_f32x4* rp = reinterpret_cast<const nll_f32x4*>(d.recon + ((long long)(k0 + k) * B + b) * D);
#pragma unroll
      for (int j = 0; j < NV; ++j) dst[j] = __builtin_nontemporal_load(rp + off[j]);  // streamed once (plain loads: +1 us in the step)
    };
    load_sample(0, cur);
    for (int k = 0; k < kn; ++k) {
      if (k + 1 < kn) load_sample(k + 1, nxt);
      nll_f32x4* gp = d.drecon ? reinterpret_cast<nll_f32x4*>(d.drecon + ((long long)(k0 + k) * B + b) * D) : nullptr;
      const float gw = gbase * (d.rowcoef ? d.rowcoef[(long long)(k0 + k) * B + b] : 1.0f);
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        nll_f32x4 gr;
        float n0, n1, n2, n3, g0, g1, g2, g3;
        nll_elem(DIST, inv_s, inv_s2, cur[j][0], xv[j][0], n0, g0);
        nll_elem(DIST, inv_s, inv_s2, cur[j][1], xv[j][1], n1, g1);
        nll_elem(DIST, inv_s, inv_s2, cur[j][2], xv[j][2], n2, g2);
        nll_elem(DIST, inv_s, inv_s2, cur[j][3], xv[j][3], n3, g3);
        acc += live[j] ? (n0 + n1) + (n2 + n3) : 0.f;
        gr[0] = g0 * gw;
        gr[1] = g1 * gw;
        gr[2] = g2 * gw;
        gr[3] = g3 * gw;
        if (gp && live[j]) gp[off[j]] = gr;
      }
      acc = wave_sum(acc);
      if (lane == 0) red[k][wave] += acc;
#pragma unroll
      for (int j = 0; j < NV; ++j) cur[j] = nxt[j];
    }
  }
}

// Sample chunks per batch row: one block walks up to KC samples of its row (x is read once).  MVK_RECON_CHUNK caps the
// chunk length (experiment hook).  Measured in the MoPoE step (MnistSvhn, K=10, B=512): 10 samples per block 28.5 us,
// 5: 30.3, 3: 34.0, 2: 31.8 -- the coarsest cut wins.
static int recon_max_chunk() {
  static int v = -1;
  if (v < 0) {
    const char* e = mvk_tune("MVK_RECON_CHUNK");
    v = e ? atoi(e) : KC;
    if (v < 1) v = 1;
    if (v > KC) v = KC;
  }
  return v;
}
static int recon_kchunks(int K, long long D) {
  const int cap = D > 1024 ? recon_max_chunk() : KC;
  return (K + cap - 1) / cap;
}

// MODE 0: scalar path (unaligned rows / row length not a multiple of 4); 1: vector path, Normal / Laplace rows; 2: vector
// path, Bernoulli rows (a separate instantiation: its exp / log1p temporaries cost 24 more registers, i.e. 2 waves per
// SIMD that the streaming Normal / Laplace rows would lose)
template <int MODE, bool FWD>
__global__ __launch_bounds__(NLL_THREADS) void recon_nll_kernel(const ReconTable tb, int K, int B) {
  mvk_prof_begin(tb.prof);
  // locate the modality of this block
  int mi = 0;
#pragma unroll
  for (int i = 1; i < MAXM; ++i)
    if (i < tb.n && (int)blockIdx.x >= tb.block_start[i]) mi = i;
  const mvk_recon_desc& d = tb.d[mi];
  const int local = blockIdx.x - tb.block_start[mi];
  const int kchunks = tb.kchunks[mi];
  const int kper = (K + kchunks - 1) / kchunks;  // balanced chunks (K=10 -> 5+5, not 8+2)
  const int b = local / kchunks;
  const int k0 = (local % kchunks) * kper;
  const int kn = (K - k0) < kper ? (K - k0) : kper;
  const long long D = d.D;
  const float inv_s = 1.0f / d.scale, inv_s2 = inv_s * inv_s;
  const float mk = d.mask ? (d.mask[b] ? 1.0f : 0.0f) : 1.0f;
  const float gbase = d.coef * d.rescale * mk;
  const float* xrow = d.x + (long long)b * D;
  __shared__ float red[KC][NLL_THREADS / 64];  // partial row sums per (sample, wave); only lane 0 of a wave touches it
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < KC; ++k) red[k][wave] = 0.f;
  }

  if (MODE > 0) {
    // distribution and tile width resolved once per block: the element loops below are branch-free
    const bool small = D <= 4 * NLL_THREADS;
#define MVK_RECON_BODY(DIST)                                                                              \
  do {                                                                                                    \
    if (small) recon_vec_body<DIST, 1>(d, B, b, k0, kn, gbase, inv_s, inv_s2, xrow, red);                  \
    else recon_vec_body<DIST, NLL_NV>(d, B, b, k0, kn, gbase, inv_s, inv_s2, xrow, red);                   \
  } while (0)
    if (MODE == 2) MVK_RECON_BODY(MVK_DIST_BERNOULLI);
    else if (d.dist == MVK_DIST_NORMAL) MVK_RECON_BODY(MVK_DIST_NORMAL);
    else MVK_RECON_BODY(MVK_DIST_LAPLACE);
#undef MVK_RECON_BODY
  } else {
    for (int k = 0; k < kn; ++k) {
      const long long ro = ((long long)(k0 + k) * B + b) * D;
      const float gw = gbase * (d.rowcoef ? d.rowcoef[(long long)(k0 + k) * B + b] : 1.0f);
      float acc = 0.f;
      for (long long i = threadIdx.x; i < D; i += NLL_THREADS) {
        float n, g;
        nll_elem(d.dist, inv_s, inv_s2, d.recon[ro + i], xrow[i], n, g);
        acc += n;
        if (d.drecon) d.drecon[ro + i] = g * gw;
      }
      acc = wave_sum(acc);
      if (lane == 0) red[k][wave] += acc;
    }
  }

  if (FWD) {
    __syncthreads();
    if ((int)threadIdx.x < kn) {
      const int k = threadIdx.x;
      float s = red[k][0] + red[k][1] + red[k][2] + red[k][3];
      s = (s + nll_row_const(d.dist, d.scale, D)) * d.rescale;
      d.rows[(long long)(k0 + k) * B + b] = s;
    }
  }
  mvk_prof_end(tb.prof);
}

// ---------------------------------------------------------------------------------------------------------
// Categorical decoder distribution (base_utils.py:28-57 `cross_entropy`): log p = x * log_softmax(r + 1e-6) over the
// LAST dimension (n_classes), summed over the positions of the sample.  One block per (k, b); every wave walks class
// rows (positions) of length C: max, sum-exp, then the gradient  d(-log p)/dr_c = softmax_c * sum_c' x_c' - x_c.
// ---------------------------------------------------------------------------------------------------------
template <bool FWD>
__global__ __launch_bounds__(NLL_THREADS) void recon_categorical_kernel(const mvk_recon_desc d, int K, int B) {
  const int kb = blockIdx.x;
  const int b = kb % B;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = d.n_classes;
  const long long P = d.D / C;
  const float mk = d.mask ? (d.mask[b] ? 1.0f : 0.0f) : 1.0f;
  const float gw = d.coef * d.rescale * mk * (d.rowcoef ? d.rowcoef[kb] : 1.0f);
  const float* r0 = d.recon + (long long)kb * d.D;
  const float* x0 = d.x + (long long)b * d.D;
  float* g0 = d.drecon ? d.drecon + (long long)kb * d.D : nullptr;
  float nll = 0.f;
  for (long long p = wave; p < P; p += NLL_THREADS / 64) {
    const float* r = r0 + p * C;
    const float* x = x0 + p * C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, r[c] + 1e-6f);
    mx = wave_max(mx);
    float se = 0.f, sxr = 0.f, sx = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float v = r[c] + 1e-6f;
      se += expf(v - mx);
      sxr += x[c] * v;
      sx += x[c];
    }
    se = wave_sum(se);
    sxr = wave_sum(sxr);
    sx = wave_sum(sx);
    const float lse = mx + logf(se);
    nll += -(sxr - lse * sx);
    if (g0) {
      float* g = g0 + p * C;
      for (int c = lane; c < C; c += 64) g[c] = gw * (expf(r[c] + 1e-6f - lse) * sx - x[c]);
    }
  }
  if (FWD) {
    __shared__ float red[NLL_THREADS / 64];
    if (lane == 0) red[wave] = nll;  // every lane of a wave holds the wave's total
    __syncthreads();
    if (threadIdx.x == 0) d.rows[kb] = (red[0] + red[1] + red[2] + red[3]) * d.rescale;
  }
}

static int launch_recon(const mvk_recon_desc* descs, int n_mod, int K, int B, bool fwd, hipStream_t s) {
  if (!descs || n_mod < 1 || n_mod > MAXM || K < 1 || B < 0) return MVK_EINVAL;
  if (B == 0) return MVK_OK;
  for (int i = 0; i < n_mod; ++i) {
    const mvk_recon_desc& d = descs[i];
    if (!d.recon || !d.x || d.D <= 0 || (fwd && !d.rows) || (!fwd && !d.drecon)) return MVK_EINVAL;
    if (d.dist < 0 || d.dist > MVK_DIST_CATEGORICAL) return MVK_EINVAL;
    if (d.dist == MVK_DIST_CATEGORICAL && (d.n_classes < 1 || d.D % d.n_classes)) return MVK_EINVAL;
  }
  for (int i = 0; i < n_mod; ++i) {  // categorical terms: class-row kernel, one launch per term
    if (descs[i].dist != MVK_DIST_CATEGORICAL) continue;
    if (fwd)
      hipLaunchKernelGGL((recon_categorical_kernel<true>), dim3(K * B), dim3(NLL_THREADS), 0, s, descs[i], K, B);
    else
      hipLaunchKernelGGL((recon_categorical_kernel<false>), dim3(K * B), dim3(NLL_THREADS), 0, s, descs[i], K, B);
    MVK_CHECK_LAUNCH();
  }
  // launch per group (scalar rows, Normal / Laplace vector rows, Bernoulli vector rows) so that an odd-sized or
  // register-hungry modality does not slow the others down
  for (int pass = 0; pass < 3; ++pass) {
    ReconTable tb;
    tb.n = 0;
    int blocks = 0;
    for (int i = 0; i < n_mod; ++i) {
      const mvk_recon_desc& d = descs[i];
      if (d.dist == MVK_DIST_CATEGORICAL) continue;
      const bool v = !((d.D & 3) || !mvk_aligned16(d.recon) || !mvk_aligned16(d.x) || (d.drecon && !mvk_aligned16(d.drecon)));
      const int group = !v ? 0 : (d.dist == MVK_DIST_BERNOULLI ? 2 : 1);
      if (group != pass) continue;
      tb.d[tb.n] = d;
      tb.block_start[tb.n] = blocks;
      tb.kchunks[tb.n] = recon_kchunks(K, d.D);
      blocks += B * tb.kchunks[tb.n];
      tb.n++;
    }
    if (tb.n == 0) continue;
    tb.block_start[tb.n] = blocks;
    double bytes = 0.0;  // algorithmic HBM bytes: reconstructions read once, gradients written once, x once
    for (int i = 0; i < tb.n; ++i)
      bytes += 4.0 * B * (double)tb.d[i].D * (K + (tb.d[i].drecon ? K : 0) + 1);
    tb.prof = (fwd && pass == 1) ? mvk::prof_next(1, bytes) : nullptr;
    const dim3 g(blocks), t(NLL_THREADS);
    if (pass == 0) {
      if (fwd) hipLaunchKernelGGL((recon_nll_kernel<0, true>), g, t, 0, s, tb, K, B);
      else hipLaunchKernelGGL((recon_nll_kernel<0, false>), g, t, 0, s, tb, K, B);
    } else if (pass == 1) {
      if (fwd) hipLaunchKernelGGL((recon_nll_kernel<1, true>), g, t, 0, s, tb, K, B);
      else hipLaunchKernelGGL((recon_nll_kernel<1, false>), g, t, 0, s, tb, K, B);
    } else {
      if (fwd) hipLaunchKernelGGL((recon_nll_kernel<2, true>), g, t, 0, s, tb, K, B);
      else hipLaunchKernelGGL((recon_nll_kernel<2, false>), g, t, 0, s, tb, K, B);
    }
    MVK_CHECK_LAUNCH();
    mvk::prof_fold(tb.prof, s);
  }
  return MVK_OK;
}


// ---------------------------------------------------------------------------------------------------------
// JMVAE posterior (jmvae_model.py:133-174): z = mu + exp(lv/2) eps of the JOINT encoder, KL(q(z|X) || N(0,I)) rows
// and the rows of LJM = sum_m KL(q(z|X) || q(z|x_m)) = sum_m 1/2 (lv_m - lv + (e^lv + (mu - mu_m)^2) / e^lv_m - 1).
// pt.mu / pt.lv hold the unimodal encoders' outputs.  One wave per batch row.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void jmvae_posterior_fwd_kernel(const float* __restrict__ jmu,
                                                                  const float* __restrict__ jlv, const PtrTable pt,
                                                                  int M, const float* __restrict__ eps, int K, int B,
                                                                  int L, float* __restrict__ z,
                                                                  float* __restrict__ kld_rows,
                                                                  float* __restrict__ ljm_rows) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float kld = 0.f, ljm = 0.f;
  for (int l = lane; l < L; l += 64) {
    const long long o = (long long)b * L + l;
    const float mu = jmu[o], lv = jlv[o];
    const float ev = expf(lv);
    kld += -0.5f * (1.0f + lv - mu * mu - ev);
    float acc = 0.f;  // summed over the modalities per element first, like the reference (:160-172)
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      if (m < M) {
        const float um = pt.mu[m][o], ul = pt.lv[m][o];
        const float d = mu - um;
        acc += 0.5f * (ul - lv + (ev + d * d) / expf(ul) - 1.0f);
      }
    }
    ljm += acc;
    const float sd = expf(0.5f * lv);
    for (int k = 0; k < K; ++k) {
      const long long zo = ((long long)k * B + b) * L + l;
      z[zo] = mu + sd * eps[zo];
    }
  }
  kld = wave_sum(kld);
  ljm = wave_sum(ljm);
  if (lane == 0) {
    kld_rows[b] = kld;
    ljm_rows[b] = ljm;
  }
}

__global__ __launch_bounds__(256) void jmvae_posterior_bwd_kernel(const float* __restrict__ jmu,
                                                                  const float* __restrict__ jlv, const PtrTable pt,
                                                                  const OutPtrTable ot, int M,
                                                                  const float* __restrict__ eps,
                                                                  const float* __restrict__ dz, int K, int B, int L,
                                                                  const float* __restrict__ gkld_rows,
                                                                  const float* __restrict__ gljm_rows,
                                                                  float* __restrict__ djmu, float* __restrict__ djlv) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * L) return;
  const int b = (int)(i / L);
  const float mu = jmu[i], lv = jlv[i];
  const float ev = expf(lv), sd = expf(0.5f * lv);
  const float gk = gkld_rows ? gkld_rows[b] : 0.f, gj = gljm_rows ? gljm_rows[b] : 0.f;
  float gmu = gk * mu, glv = gk * (-0.5f) * (1.0f - ev);
  if (dz) {
    for (int k = 0; k < K; ++k) {
      const long long zo = (long long)k * B * L + i;
      const float g = dz[zo];
      gmu += g;
      glv += g * 0.5f * sd * eps[zo];
    }
  }
#pragma unroll
  for (int m = 0; m < MAXM; ++m) {
    if (m < M) {
      const float um = pt.mu[m][i], ul = pt.lv[m][i];
      const float iv = 1.0f / expf(ul);
      const float d = mu - um;
      gmu += gj * d * iv;
      glv += gj * 0.5f * (ev * iv - 1.0f);
      ot.dmu[m][i] = -gj * d * iv;
      ot.dlv[m][i] = gj * 0.5f * (1.0f - (ev + d * d) * iv);
    }
  }
  djmu[i] = gmu;
  djlv[i] = glv;
}

// ---------------------------------------------------------------------------------------------------------
// MVAE posterior (models/mvae/mvae_model.py:56-118): for every subset s of the objective, the product of the available
// experts of s AND the N(0,I) prior, in the log-sum-exp form of `stable_poe` (base_utils.py:133-147), one
// reparameterised sample per subset and row, and the KL to the prior.  One wave per batch row, lanes over the latent.
// The sample of subset s is written once per modality of s, into that modality's decoder input zm[m][slot(s,m)]
// ([K_m, B, L], K_m = number of subsets holding m), so every decoder runs once over all its subsets.
// ---------------------------------------------------------------------------------------------------------
constexpr int MAXS = MVK_MVAE_MAX_SUBSETS;
struct MvaeTable {
  uint32_t bits[MAXS];
  signed char slot[MAXS][MAXM];  // slab of zm[m] that holds subset s (-1: m not in s)
  float* zm[MAXM];
  const float* dzm[MAXM];
};

__global__ __launch_bounds__(256) void mvae_posterior_fwd_kernel(const PtrTable pt, const MvaeTable tb, int M, int S,
                                                                 const float* __restrict__ eps, int B, int L,
                                                                 float* __restrict__ kld_rows,
                                                                 float* __restrict__ sub_mu,
                                                                 float* __restrict__ sub_lv) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  bool avail[MAXM];
#pragma unroll
  for (int m = 0; m < MAXM; ++m) avail[m] = (m < M) && (pt.mask[m] ? pt.mask[m][b] != 0 : true);
  for (int s = 0; s < S; ++s) {
    const uint32_t bits = tb.bits[s];
    float kld = 0.f;
    for (int l = lane; l < L; l += 64) {
      const long long o = (long long)b * L + l;
      float a[MAXM], mu[MAXM];
      float amax = 0.f;  // the prior expert: ln(1/var) = 0
#pragma unroll
      for (int m = 0; m < MAXM; ++m) {
        a[m] = -INFINITY;  // missing rows: log-variance = +inf (mvae_model.py:71-75)
        mu[m] = 0.f;
        if (m < M && ((bits >> m) & 1u) && avail[m]) {
          a[m] = -pt.lv[m][o];
          mu[m] = pt.mu[m][o];
          amax = fmaxf(amax, a[m]);
        }
      }
      float se = expf(-amax), num = 0.f;  // prior: exp(0 - amax), mean 0
#pragma unroll
      for (int m = 0; m < MAXM; ++m) {
        if (a[m] > -INFINITY) {
          se += expf(a[m] - amax);
          num += expf(a[m]) * mu[m];  // (exp(ln_inv_vars) * mus).sum(0)
        }
      }
      const float lnv = -(amax + logf(se));
      const float mus = num * expf(lnv);
      kld += -0.5f * (1.0f + lnv - mus * mus - expf(lnv));
      const long long so = ((long long)s * B + b) * L + l;
      const float zz = mus + expf(0.5f * lnv) * eps[so];
#pragma unroll
      for (int m = 0; m < MAXM; ++m) {
        if (m < M && tb.slot[s][m] >= 0) tb.zm[m][((long long)tb.slot[s][m] * B + b) * L + l] = zz;
      }
      if (sub_mu) {
        sub_mu[so] = mus;
        sub_lv[so] = lnv;
      }
    }
    kld = wave_sum(kld);
    if (lane == 0) kld_rows[(long long)s * B + b] = kld;
  }
}

// d mu_e = g_mu w_e, d lv_e = w_e (g_lnv - g_mu (mu_e - mu_s)) with w_e = exp(-lv_e) / sum_e' exp(-lv_e') the
// precision weights (softmax of -lv over the experts and the prior): no division by a possibly tiny variance.
__global__ __launch_bounds__(256) void mvae_posterior_bwd_kernel(const PtrTable pt, const OutPtrTable ot,
                                                                 const MvaeTable tb, int M, int S,
                                                                 const float* __restrict__ eps, int B, int L,
                                                                 const float* __restrict__ gkld_rows) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  bool avail[MAXM];
#pragma unroll
  for (int m = 0; m < MAXM; ++m) avail[m] = (m < M) && (pt.mask[m] ? pt.mask[m][b] != 0 : true);
  for (int l = lane; l < L; l += 64) {
    const long long o = (long long)b * L + l;
    float dmu[MAXM], dlv[MAXM], mu[MAXM], nlv[MAXM];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      dmu[m] = dlv[m] = mu[m] = nlv[m] = 0.f;
      if (m < M) {
        mu[m] = pt.mu[m][o];
        nlv[m] = -pt.lv[m][o];
      }
    }
    for (int s = 0; s < S; ++s) {
      const uint32_t bits = tb.bits[s];
      float amax = 0.f;
      bool in[MAXM];
#pragma unroll
      for (int m = 0; m < MAXM; ++m) {
        in[m] = m < M && ((bits >> m) & 1u) && avail[m];
        if (in[m]) amax = fmaxf(amax, nlv[m]);
      }
      float se = expf(-amax), num = 0.f;
#pragma unroll
      for (int m = 0; m < MAXM; ++m) {
        if (in[m]) {
          se += expf(nlv[m] - amax);
          num += expf(nlv[m]) * mu[m];
        }
      }
      const float lnv = -(amax + logf(se));
      const float mus = num * expf(lnv);
      const long long so = ((long long)s * B + b) * L + l;
      float dz = 0.f;
#pragma unroll
      for (int m = 0; m < MAXM; ++m) {
        if (m < M && tb.slot[s][m] >= 0 && tb.dzm[m]) dz += tb.dzm[m][((long long)tb.slot[s][m] * B + b) * L + l];
      }
      const float gk = gkld_rows ? gkld_rows[(long long)s * B + b] : 0.f;
      const float g_mu = dz + gk * mus;
      const float g_lnv = dz * 0.5f * expf(0.5f * lnv) * eps[so] - 0.5f * gk * (1.0f - expf(lnv));
#pragma unroll
      for (int m = 0; m < MAXM; ++m) {
        if (in[m]) {
          const float w = expf(nlv[m] - amax) / se;
          dmu[m] += g_mu * w;
          dlv[m] += w * (g_lnv - g_mu * (mu[m] - mus));
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      if (m < M) {
        ot.dmu[m][o] = dmu[m];
        ot.dlv[m][o] = dlv[m];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Diagonal Gaussian: K reparameterised samples and the KL to N(0, I) per row (the modality-specific "style" latents of
// MoPoE, mopoe_model.py:171-178 and :212-221).  One wave per row.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gauss_sample_kl_fwd_kernel(const float* __restrict__ mu,
                                                                  const float* __restrict__ lv,
                                                                  const float* __restrict__ eps, int K, int B, int L,
                                                                  float* __restrict__ w, float* __restrict__ kl_rows) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float kl = 0.f;
  for (int l = lane; l < L; l += 64) {
    const long long o = (long long)b * L + l;
    const float m = mu[o], v = lv[o];
    kl += -0.5f * (1.0f - expf(v) - m * m + v);
    const float sd = expf(0.5f * v);
    for (int k = 0; k < K; ++k) {
      const long long zo = ((long long)k * B + b) * L + l;
      w[zo] = m + sd * eps[zo];
    }
  }
  kl = wave_sum(kl);
  if (lane == 0) kl_rows[b] = kl;
}

__global__ __launch_bounds__(256) void gauss_sample_kl_bwd_kernel(const float* __restrict__ mu,
                                                                  const float* __restrict__ lv,
                                                                  const float* __restrict__ eps,
                                                                  const float* __restrict__ dw,
                                                                  const float* __restrict__ gkl, int K, int B, int L,
                                                                  float* __restrict__ dmu, float* __restrict__ dlv) {
  const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
  if (o >= (long long)B * L) return;
  const int b = (int)(o / L);
  const float m = mu[o], v = lv[o];
  const float sd = expf(0.5f * v);
  float sdw = 0.f, sdwe = 0.f;
  if (dw) {
    for (int k = 0; k < K; ++k) {
      const long long zo = (long long)k * B * L + o;
      const float g = dw[zo];
      sdw += g;
      sdwe += g * eps[zo];
    }
  }
  const float gk = gkl ? gkl[b] : 0.f;
  dmu[o] = sdw + gk * m;
  dlv[o] = 0.5f * sd * sdwe + gk * 0.5f * (expf(v) - 1.0f);
}

}  // namespace

extern "C" {

int mvk_mopoe_posterior_fwd(const float* const* mu, const float* const* lv, int M, const int32_t* subset_masks, int S,
                            const int32_t* sel, const float* weights, const float* eps, int K, int B, int L, float* z,
                            float* kld_rows, float* mus_out, float* lvs_out, float* joint_mu, float* joint_lv,
                            void* stream) {
  if (!mu || !lv || M < 1 || M > MAXM || !subset_masks || S < 1 || !sel || !eps || !z || !kld_rows || K < 1 || L < 1)
    return MVK_EINVAL;
  if ((mus_out == nullptr) != (lvs_out == nullptr) || (joint_mu == nullptr) != (joint_lv == nullptr)) return MVK_EINVAL;
  if (B <= 0) return B == 0 ? MVK_OK : MVK_EINVAL;
  PtrTable pt{};
  for (int m = 0; m < M; ++m) {
    if (!mu[m] || !lv[m]) return MVK_EINVAL;
    pt.mu[m] = mu[m];
    pt.lv[m] = lv[m];
  }
  // profiler kind 11: the latency-sized members of the ELBO group (bytes: encoder outputs in, noise in, samples out)
  mvk_prof_slot* prof = mvk::prof_next(11, 4.0 * B * L * (2.0 * M + 2.0 * K) + 4.0 * B);
  hipLaunchKernelGGL(mopoe_posterior_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, mvk_stream(stream), pt, M,
                     subset_masks, S, sel, weights, eps, K, B, L, z, kld_rows, mus_out, lvs_out, joint_mu, joint_lv, prof);
  MVK_CHECK_LAUNCH();
  mvk::prof_fold(prof, mvk_stream(stream));
  return MVK_OK;
}

int mvk_mopoe_posterior_bwd(const float* const* mu, const float* const* lv, int M, const int32_t* subset_masks, int S,
                            const int32_t* sel, const float* weights, const float* eps, const float* dz, int K, int B,
                            int L, const float* gkld_rows, float* const* dmu, float* const* dlv, void* stream) {
  if (!mu || !lv || M < 1 || M > MAXM || !subset_masks || S < 1 || !sel || !eps || !dz || !dmu || !dlv || K < 1 ||
      L < 1)
    return MVK_EINVAL;
  if (B <= 0) return B == 0 ? MVK_OK : MVK_EINVAL;
  PtrTable pt{};
  OutPtrTable ot{};
  for (int m = 0; m < M; ++m) {
    if (!mu[m] || !lv[m] || !dmu[m] || !dlv[m]) return MVK_EINVAL;
    pt.mu[m] = mu[m];
    pt.lv[m] = lv[m];
    ot.dmu[m] = dmu[m];
    ot.dlv[m] = dlv[m];
  }
  mvk_prof_slot* prof = mvk::prof_next(11, 4.0 * B * L * (4.0 * M + 2.0 * K) + 4.0 * B);
  hipLaunchKernelGGL(mopoe_posterior_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, mvk_stream(stream), pt, ot, M,
                     subset_masks, S, sel, weights, eps, dz, K, B, L, gkld_rows, prof);
  MVK_CHECK_LAUNCH();
  mvk::prof_fold(prof, mvk_stream(stream));
  return MVK_OK;
}

int mvk_mvtcae_posterior_fwd(const float* const* mu, const float* const* lv, const uint8_t* const* masks, int M,
                             const float* eps, int K, int B, int L, float* z, float* joint_kl_rows,
                             float* cond_kl_rows, float* joint_mu, float* joint_lv, void* stream) {
  if (!mu || !lv || M < 1 || M > MAXM || !eps || !z || !joint_kl_rows || !cond_kl_rows || K < 1 || L < 1)
    return MVK_EINVAL;
  if ((joint_mu == nullptr) != (joint_lv == nullptr)) return MVK_EINVAL;
  if (B <= 0) return B == 0 ? MVK_OK : MVK_EINVAL;
  PtrTable pt{};
  for (int m = 0; m < M; ++m) {
    if (!mu[m] || !lv[m]) return MVK_EINVAL;
    pt.mu[m] = mu[m];
    pt.lv[m] = lv[m];
    pt.mask[m] = masks ? masks[m] : nullptr;
  }
  hipLaunchKernelGGL(mvtcae_posterior_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, mvk_stream(stream), pt, M, eps, K,
                     B, L, z, joint_kl_rows, cond_kl_rows, joint_mu, joint_lv);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_mvtcae_posterior_bwd(const float* const* mu, const float* const* lv, const uint8_t* const* masks, int M,
                             const float* eps, const float* dz, int K, int B, int L, const float* gjoint_rows,
                             const float* gcond_rows, float* const* dmu, float* const* dlv, void* stream) {
  if (!mu || !lv || M < 1 || M > MAXM || !eps || !dz || !dmu || !dlv || K < 1 || L < 1) return MVK_EINVAL;
  if (B <= 0) return B == 0 ? MVK_OK : MVK_EINVAL;
  PtrTable pt{};
  OutPtrTable ot{};
  for (int m = 0; m < M; ++m) {
    if (!mu[m] || !lv[m] || !dmu[m] || !dlv[m]) return MVK_EINVAL;
    pt.mu[m] = mu[m];
    pt.lv[m] = lv[m];
    pt.mask[m] = masks ? masks[m] : nullptr;
    ot.dmu[m] = dmu[m];
    ot.dlv[m] = dlv[m];
  }
  hipLaunchKernelGGL(mvtcae_posterior_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, mvk_stream(stream), pt, ot, M, eps,
                     dz, K, B, L, gjoint_rows, gcond_rows);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

static int mvae_table(const int32_t* subset_bits, int S, int M, MvaeTable* tb) {
  if (!subset_bits || S < 1 || S > MAXS) return MVK_EINVAL;
  int count[MAXM] = {0};
  const uint32_t full = (M >= 32) ? 0xffffffffu : ((1u << M) - 1u);
  for (int s = 0; s < S; ++s) {
    const uint32_t bits = (uint32_t)subset_bits[s];
    if (bits == 0 || (bits & ~full)) return MVK_EINVAL;
    tb->bits[s] = bits;
    for (int m = 0; m < MAXM; ++m) {
      tb->slot[s][m] = -1;
      if (m < M && ((bits >> m) & 1u)) {
        if (count[m] > 127) return MVK_EINVAL;
        tb->slot[s][m] = (signed char)count[m]++;
      }
    }
  }
  return MVK_OK;
}

int mvk_mvae_posterior_fwd(const float* const* mu, const float* const* lv, const uint8_t* const* masks, int M,
                           const int32_t* subset_bits, int S, const float* eps, int B, int L, float* const* zm,
                           float* kld_rows, float* sub_mu, float* sub_lv, void* stream) {
  if (!mu || !lv || M < 1 || M > MAXM || !eps || !zm || !kld_rows || L < 1) return MVK_EINVAL;
  if ((sub_mu == nullptr) != (sub_lv == nullptr)) return MVK_EINVAL;
  MvaeTable tb{};
  if (mvae_table(subset_bits, S, M, &tb) != MVK_OK) return MVK_EINVAL;
  if (B <= 0) return B == 0 ? MVK_OK : MVK_EINVAL;
  PtrTable pt{};
  for (int m = 0; m < M; ++m) {
    if (!mu[m] || !lv[m]) return MVK_EINVAL;
    pt.mu[m] = mu[m];
    pt.lv[m] = lv[m];
    pt.mask[m] = masks ? masks[m] : nullptr;
    tb.zm[m] = zm[m];
    bool used = false;
    for (int s = 0; s < S; ++s) used = used || tb.slot[s][m] >= 0;
    if (used && !zm[m]) return MVK_EINVAL;
  }
  hipLaunchKernelGGL(mvae_posterior_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, mvk_stream(stream), pt, tb, M, S, eps,
                     B, L, kld_rows, sub_mu, sub_lv);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_mvae_posterior_bwd(const float* const* mu, const float* const* lv, const uint8_t* const* masks, int M,
                           const int32_t* subset_bits, int S, const float* eps, const float* const* dzm, int B, int L,
                           const float* gkld_rows, float* const* dmu, float* const* dlv, void* stream) {
  if (!mu || !lv || M < 1 || M > MAXM || !eps || !dzm || !dmu || !dlv || L < 1) return MVK_EINVAL;
  MvaeTable tb{};
  if (mvae_table(subset_bits, S, M, &tb) != MVK_OK) return MVK_EINVAL;
  if (B <= 0) return B == 0 ? MVK_OK : MVK_EINVAL;
  PtrTable pt{};
  OutPtrTable ot{};
  for (int m = 0; m < M; ++m) {
    if (!mu[m] || !lv[m] || !dmu[m] || !dlv[m]) return MVK_EINVAL;
    pt.mu[m] = mu[m];
    pt.lv[m] = lv[m];
    pt.mask[m] = masks ? masks[m] : nullptr;
    ot.dmu[m] = dmu[m];
    ot.dlv[m] = dlv[m];
    tb.dzm[m] = dzm[m];
  }
  hipLaunchKernelGGL(mvae_posterior_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, mvk_stream(stream), pt, ot, tb, M, S,
                     eps, B, L, gkld_rows);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_gauss_sample_kl_fwd(const float* mu, const float* lv, const float* eps, int K, int B, int L, float* w,
                            float* kl_rows, void* stream) {
  if (B == 0) return MVK_OK;
  if (!mu || !lv || !eps || !w || !kl_rows || K < 1 || L < 1 || B < 0) return MVK_EINVAL;
  hipLaunchKernelGGL(gauss_sample_kl_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, mvk_stream(stream), mu, lv, eps, K, B,
                     L, w, kl_rows);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_gauss_sample_kl_bwd(const float* mu, const float* lv, const float* eps, const float* dw, const float* gkl,
                            int K, int B, int L, float* dmu, float* dlv, void* stream) {
  if (!mu || !lv || !eps || !dmu || !dlv || K < 1 || L < 1) return MVK_EINVAL;
  if (B <= 0) return B == 0 ? MVK_OK : MVK_EINVAL;
  const long long n = (long long)B * L;
  hipLaunchKernelGGL(gauss_sample_kl_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, mvk_stream(stream),
                     mu, lv, eps, dw, gkl, K, B, L, dmu, dlv);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_jmvae_posterior_fwd(const float* joint_mu, const float* joint_lv, const float* const* mu, const float* const* lv,
                            int M, const float* eps, int K, int B, int L, float* z, float* kld_rows, float* ljm_rows,
                            void* stream) {
  if (!joint_mu || !joint_lv || !mu || !lv || M < 1 || M > MAXM || !eps || !z || !kld_rows || !ljm_rows || K < 1 ||
      L < 1)
    return MVK_EINVAL;
  if (B <= 0) return B == 0 ? MVK_OK : MVK_EINVAL;
  PtrTable pt{};
  for (int m = 0; m < M; ++m) {
    if (!mu[m] || !lv[m]) return MVK_EINVAL;
    pt.mu[m] = mu[m];
    pt.lv[m] = lv[m];
  }
  hipLaunchKernelGGL(jmvae_posterior_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, mvk_stream(stream), joint_mu,
                     joint_lv, pt, M, eps, K, B, L, z, kld_rows, ljm_rows);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_jmvae_posterior_bwd(const float* joint_mu, const float* joint_lv, const float* const* mu, const float* const* lv,
                            int M, const float* eps, const float* dz, int K, int B, int L, const float* gkld_rows,
                            const float* gljm_rows, float* djoint_mu, float* djoint_lv, float* const* dmu,
                            float* const* dlv, void* stream) {
  if (!joint_mu || !joint_lv || !mu || !lv || M < 1 || M > MAXM || !eps || !djoint_mu || !djoint_lv || !dmu || !dlv ||
      K < 1 || L < 1)
    return MVK_EINVAL;
  if (B <= 0) return B == 0 ? MVK_OK : MVK_EINVAL;
  PtrTable pt{};
  OutPtrTable ot{};
  for (int m = 0; m < M; ++m) {
    if (!mu[m] || !lv[m] || !dmu[m] || !dlv[m]) return MVK_EINVAL;
    pt.mu[m] = mu[m];
    pt.lv[m] = lv[m];
    ot.dmu[m] = dmu[m];
    ot.dlv[m] = dlv[m];
  }
  const long long n = (long long)B * L;
  hipLaunchKernelGGL(jmvae_posterior_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, mvk_stream(stream),
                     joint_mu, joint_lv, pt, ot, M, eps, dz, K, B, L, gkld_rows, gljm_rows, djoint_mu, djoint_lv);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_recon_nll_fwd(const mvk_recon_desc* descs, int n_mod, int K, int B, void* stream) {
  return launch_recon(descs, n_mod, K, B, true, mvk_stream(stream));
}

int mvk_recon_nll_bwd(const mvk_recon_desc* descs, int n_mod, int K, int B, void* stream) {
  return launch_recon(descs, n_mod, K, B, false, mvk_stream(stream));
}

}  // extern "C"
