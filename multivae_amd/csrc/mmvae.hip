// MMVAE mixture-of-experts importance weights (IWAE / DReG), Normal and Laplace(softmax-scale) families.
// Follows models/mmvae/mmvae_model.py:66-74 (log_var_to_std), :95-158 (forward), :160-236 (compute_k_lws),
// :238-292 (dreg_looser / iwae_looser).
// MMVAE+ (models/mmvaePlus/mmvaePlus_model.py:110-360) runs on the same kernels: the latent of a modality is the
// concatenation [u (shared, Ls dims), w (private)], the mixture density covers the shared dims only, the private dims
// are scored by the modality's own posterior (lqw), the prior terms are weighted by beta, and the private part of a
// cross-modal decoder input is sampled from the target modality's prior (cross_latent kernels).  All tensors are small ([K,B,L], L ~ 20-64): these kernels are
// latency-bound; the design goal is few launches and deterministic reductions.
#include "common.hpp"

namespace {

constexpr int MAXM = MVK_MAX_MODALITIES;
constexpr float HALF_LOG_2PI = 0.918938533204672742f;

struct MmPtrs {
  const float* mu[MAXM];
  const float* sd[MAXM];
  const float* noise[MAXM];
  const uint8_t* mask[MAXM];
  float* z[MAXM];
  float* lpz[MAXM];
  float* lqz[MAXM];
  float* lq_all[MAXM];
  float* lqw[MAXM];  // MMVAE+: log q_c(w_c) rows (NULL for MMVAE)
};

__device__ __forceinline__ float lat_logp(int family, float z, float loc, float sd) {
  if (family == MVK_FAMILY_NORMAL) {
    const float d = z - loc;
    return -(d * d) / (2.0f * sd * sd) - logf(sd) - HALF_LOG_2PI;
  }
  return -logf(2.0f * sd) - fabsf(z - loc) / sd;
}
// d logp / d z  (d/d loc is its negative)
__device__ __forceinline__ float lat_dlogp_dz(int family, float z, float loc, float sd) {
  const float d = z - loc;
  if (family == MVK_FAMILY_NORMAL) return -d / (sd * sd);
  return -(d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / sd;
}
__device__ __forceinline__ float lat_dlogp_dsd(int family, float z, float loc, float sd) {
  const float d = z - loc;
  if (family == MVK_FAMILY_NORMAL) return d * d / (sd * sd * sd) - 1.0f / sd;
  return -1.0f / sd + fabsf(d) / (sd * sd);
}
// z = loc + sd * t(noise)
__device__ __forceinline__ float lat_t(int family, float noise) {
  if (family == MVK_FAMILY_NORMAL) return noise;
  const float sg = noise > 0.f ? 1.f : (noise < 0.f ? -1.f : 0.f);
  return -sg * log1pf(-fabsf(noise));
}

// ---- std from log-variance ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void std_fwd_kernel(const float* __restrict__ lv, int rows, int L, int family,
                                                      float* __restrict__ sd) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* x = lv + (long long)r * L;
  float* y = sd + (long long)r * L;
  if (family == MVK_FAMILY_NORMAL) {
    for (int l = lane; l < L; l += 64) y[l] = expf(0.5f * x[l]);
    return;
  }
  if (family == MVK_FAMILY_NORMAL_SOFTPLUS) {  // F.softplus (threshold 20) + 1e-6, mmvaePlus_model.py:117-118
    for (int l = lane; l < L; l += 64) y[l] = (x[l] > 20.f ? x[l] : log1pf(expf(x[l]))) + 1e-6f;
    return;
  }
  float mx = -INFINITY;
  for (int l = lane; l < L; l += 64) mx = fmaxf(mx, x[l]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int l = lane; l < L; l += 64) s += expf(x[l] - mx);
  s = wave_sum(s);
  for (int l = lane; l < L; l += 64) y[l] = expf(x[l] - mx) / s * (float)L + 1e-6f;
}

__global__ __launch_bounds__(256) void std_bwd_kernel(const float* __restrict__ lv, const float* __restrict__ sd,
                                                      const float* __restrict__ dsd, int rows, int L, int family,
                                                      float* __restrict__ dlv) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const long long o = (long long)r * L;
  if (family == MVK_FAMILY_NORMAL) {
    for (int l = lane; l < L; l += 64) dlv[o + l] = dsd[o + l] * 0.5f * sd[o + l];
    return;
  }
  if (family == MVK_FAMILY_NORMAL_SOFTPLUS) {  // d softplus = sigmoid
    for (int l = lane; l < L; l += 64) {
      const float x = lv[o + l];
      dlv[o + l] = dsd[o + l] * (x > 20.f ? 1.0f : 1.0f / (1.0f + expf(-x)));
    }
    return;
  }
  // sd = softmax * L + 1e-6  ->  d lv_i = L * p_i * (dsd_i - sum_j dsd_j p_j)
  const float invL = 1.0f / (float)L;
  float dot = 0.f;
  for (int l = lane; l < L; l += 64) dot += dsd[o + l] * (sd[o + l] - 1e-6f) * invL;
  dot = wave_sum(dot);
  for (int l = lane; l < L; l += 64) {
    const float p = (sd[o + l] - 1e-6f) * invL;
    dlv[o + l] = (float)L * p * (dsd[o + l] - dot);
  }
  (void)lv;
}

// ---- forward: samples, prior log-density, mixture log-density --------------------------------------------------
// one wave per (c, k, b); lanes over l
__global__ __launch_bounds__(256) void latent_fwd_kernel(const MmPtrs p, const float* __restrict__ prior_mean,
                                                         const float* __restrict__ prior_sd, int M, int K, int B,
                                                         int L, int Ls, int family) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long total = (long long)M * K * B;
  if (row >= total) return;
  const int c = (int)(row / ((long long)K * B));
  const int kb = (int)(row % ((long long)K * B));
  const int b = kb % B;
  float lp = 0.f, lqw = 0.f;
  float lq[MAXM];
#pragma unroll
  for (int m = 0; m < MAXM; ++m) lq[m] = 0.f;
  // static selection of the conditioning modality's pointers
  const float* mu_c = nullptr;
  const float* sd_c = nullptr;
  const float* nz_c = nullptr;
  float* z_c = nullptr;
#pragma unroll
  for (int m = 0; m < MAXM; ++m)
    if (m == c) {
      mu_c = p.mu[m];
      sd_c = p.sd[m];
      nz_c = p.noise[m];
      z_c = p.z[m];
    }
  for (int l = lane; l < L; l += 64) {
    const long long o = (long long)b * L + l;
    const long long zo = (long long)kb * L + l;
    const float z = mu_c[o] + sd_c[o] * lat_t(family, nz_c[zo]);
    z_c[zo] = z;
    lp += lat_logp(family, z, prior_mean[l], prior_sd[l]);
    if (l < Ls) {  // shared dims: every modality's posterior enters the mixture
#pragma unroll
      for (int m = 0; m < MAXM; ++m)
        if (m < M) lq[m] += lat_logp(family, z, p.mu[m][o], p.sd[m][o]);
    } else {  // private dims (MMVAE+): the conditioning modality's own posterior
      lqw += lat_logp(family, z, mu_c[o], sd_c[o]);
    }
  }
  lp = wave_sum(lp);
  lqw = wave_sum(lqw);
  int navail = 0;
  float mx = -INFINITY;
#pragma unroll
  for (int m = 0; m < MAXM; ++m) {
    if (m < M) {
      lq[m] = wave_sum(lq[m]);
      const bool av = p.mask[m] ? p.mask[m][b] != 0 : true;
      if (!av) lq[m] = -INFINITY;  // mmvae_model.py:195
      navail += av ? 1 : 0;
      mx = fmaxf(mx, lq[m]);
    }
  }
  if (lane == 0) {
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < MAXM; ++m)
      if (m < M) s += expf(lq[m] - mx);
    const float lse = mx + logf(s);
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      if (m == c) {
        p.lpz[m][kb] = lp;
        p.lqz[m][kb] = lse - logf((float)navail);
        if (p.lqw[m]) p.lqw[m][kb] = lqw;
        for (int mm = 0; mm < M; ++mm) {
          float v = 0.f;
#pragma unroll
          for (int q = 0; q < MAXM; ++q)
            if (q == mm) v = lq[q];
          p.lq_all[m][(long long)mm * K * B + kb] = v;
        }
      }
    }
  }
}

// ---- objective: lw, softmax weights over K, loss -------------------------------------------------------------
struct ObjPtrs {
  const float* rows[MAXM * MAXM];  // [c][r]: rescaled NLL rows of modality r decoded from z_c
  const float* lpz[MAXM];
  const float* lqz[MAXM];
  const float* lqw[MAXM];  // MMVAE+ (may be NULL)
  const uint8_t* mask[MAXM];
  float* lw[MAXM];
  float* w[MAXM];
  float* rowcoef[MAXM];
};

__global__ __launch_bounds__(1024) void objective_kernel(const ObjPtrs p, int M, int K, int B, int dreg, float beta,
                                                         float* __restrict__ loss) {
  float local = 0.f;
  for (int idx = threadIdx.x; idx < M * B; idx += blockDim.x) {
    const int c = idx / B;
    const int b = idx % B;
    const float* lpz = nullptr;
    const float* lqz = nullptr;
    const uint8_t* mk = nullptr;
    float* lw = p.lw[c];
    float* w = p.w[c];
    float* rc = p.rowcoef[c];
    lpz = p.lpz[c];
    lqz = p.lqz[c];
    mk = p.mask[c];
    int navail = 0;
#pragma unroll
    for (int m = 0; m < MAXM; ++m)
      if (m < M) navail += (p.mask[m] ? (p.mask[m][b] != 0) : 1);
    const float mc = mk ? (mk[b] ? 1.f : 0.f) : 1.f;
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) {
      const long long o = (long long)k * B + b;
      float lpx = 0.f;  // sum_r log p(x_r | z_c) * mask_r  (mmvae_model.py:208-225)
      for (int r = 0; r < M; ++r) {
        const float mr = p.mask[r] ? (p.mask[r][b] ? 1.f : 0.f) : 1.f;
        lpx += -p.rows[c * MAXM + r][o] * mr;
      }
      const float lqw = p.lqw[c] ? p.lqw[c][o] : 0.f;
      const float v = (lpx + beta * (lpz[o] - lqz[o] - lqw)) * mc;  // mmvaePlus_model.py:262
      lw[o] = v;
      mx = fmaxf(mx, v);
    }
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += expf(lw[(long long)k * B + b] - mx);
    const float lse = mx + logf(s);
    float obj = 0.f;
    for (int k = 0; k < K; ++k) {
      const long long o = (long long)k * B + b;
      const float wk = expf(lw[o] - lse);
      w[o] = wk;
      rc[o] = -wk * mc / (float)navail;  // d loss / d lw[c][k,b]
      obj += wk * lw[o];
    }
    if (!dreg) obj = lse - logf((float)K);
    local += obj / (float)navail;
  }
  __shared__ float red[16];
  local = wave_sum(local);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    *loss = -t;
  }
}

// ---- backward on the latent side ------------------------------------------------------------------------------
struct BwdPtrs {
  const float* mu[MAXM];
  const float* sd[MAXM];
  const float* noise[MAXM];
  const float* z[MAXM];
  const uint8_t* mask[MAXM];
  const float* w[MAXM];
  const float* lq_all[MAXM];
  const float* lqz[MAXM];
  const float* dz_dec[MAXM];
  float* dmu[MAXM];
  float* dsd[MAXM];
};

// one wave per batch row b; lanes over l; loops over conditioning modality c and sample k
__global__ __launch_bounds__(256) void latent_bwd_kernel(const BwdPtrs p, const float* __restrict__ prior_mean,
                                                         const float* __restrict__ prior_sd, int M, int K, int B,
                                                         int L, int Ls, float beta, int family, int dreg,
                                                         const float* __restrict__ gscale,
                                                         float* __restrict__ dprior_sd) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float gs = gscale ? *gscale : 1.0f;
  int navail = 0;
  bool avail[MAXM];
#pragma unroll
  for (int m = 0; m < MAXM; ++m) {
    avail[m] = (m < M) && (p.mask[m] ? p.mask[m][b] != 0 : true);
    navail += avail[m] ? 1 : 0;
  }
  const float inv_n = 1.0f / (float)navail;
  for (int l = lane; l < L; l += 64) {
    const long long o = (long long)b * L + l;
    float mu[MAXM], sd[MAXM], dmu[MAXM], dsd[MAXM];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      mu[m] = (m < M) ? p.mu[m][o] : 0.f;
      sd[m] = (m < M) ? p.sd[m][o] : 1.f;
      dmu[m] = 0.f;
      dsd[m] = 0.f;
    }
    const float pm = prior_mean[l], ps = prior_sd[l];
    float dps = 0.f;
#pragma unroll
    for (int c = 0; c < MAXM; ++c) {
      if (c < M && avail[c]) {  // rows with mask_c == 0 have lw == 0 identically: no gradient
        for (int k = 0; k < K; ++k) {
          const long long kb = (long long)k * B + b;
          const long long zo = kb * L + l;
          const float wk = p.w[c][kb];
          const float g = -gs * inv_n * wk * beta;  // dLoss / d lw[c][k,b] times the weight of the latent terms
          const float z = p.z[c][zo];
          // prior
          float gz = g * lat_dlogp_dz(family, z, pm, ps);
          dps += g * lat_dlogp_dsd(family, z, pm, ps);
          // mixture posterior: lqz = LSE_m s_m - log n ; lw -= lqz
          const float lse = p.lqz[c][kb] + logf((float)navail);
#pragma unroll
          for (int m = 0; m < MAXM; ++m) {
            // shared dims: responsibilities of the mixture; private dims: the own posterior with weight 1
            const bool own = (m == c);
            if (m < M && avail[m] && (l < Ls || own)) {
              const float r = (l < Ls) ? expf(p.lq_all[c][(long long)m * K * B + kb] - lse) : 1.0f;
              const float dz_q = lat_dlogp_dz(family, z, mu[m], sd[m]);
              gz -= g * r * dz_q;
              if (!dreg) {  // IWAE differentiates the q parameters directly as well
                dmu[m] += -g * r * (-dz_q);
                dsd[m] += -g * r * lat_dlogp_dsd(family, z, mu[m], sd[m]);
              }
            }
          }
          float gtot = gz + p.dz_dec[c][zo];
          if (dreg) gtot *= wk;  // gradient hook on z (mmvae_model.py:263-266)
          const float t = lat_t(family, p.noise[c][zo]);
          // static accumulation into the conditioning modality's slot
#pragma unroll
          for (int m = 0; m < MAXM; ++m) {
            if (m == c) {
              dmu[m] += gtot;
              dsd[m] += gtot * t;
            }
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      if (m < M) {
        p.dmu[m][o] = dmu[m];
        p.dsd[m][o] = dsd[m];
      }
    }
    if (dprior_sd) dprior_sd[o] = dps;  // per-row term [B, L]: the caller sums the rows in a fixed order
  }
}

// ---- MMVAE+ cross-modal decoder input: [u_c (first Ls dims of z_c), w ~ prior of the target modality] ----------
// mmvaePlus_model.py:152-172.  zc[k,b,:Ls] = z[k,b,:Ls];  zc[k,b,Ls+j] = prior_sd[j] * t(noise[k,b,j])  (prior mean 0).
__global__ __launch_bounds__(256) void cross_latent_fwd_kernel(const float* __restrict__ z,
                                                               const float* __restrict__ prior_sd,
                                                               const float* __restrict__ noise, long long rows, int D,
                                                               int Ls, int family, float* __restrict__ zc) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * D) return;
  const long long r = i / D;
  const int l = (int)(i - r * D);
  zc[i] = l < Ls ? z[i] : prior_sd[l - Ls] * lat_t(family, noise[r * (D - Ls) + (l - Ls)]);
}

// dz[k,b,:Ls] = dzc[k,b,:Ls], 0 elsewhere; dprior_sd[j] = sum_{k,b} dzc[k,b,Ls+j] * t(noise)   (one block per j)
__global__ __launch_bounds__(256) void cross_latent_bwd_kernel(const float* __restrict__ dzc,
                                                               const float* __restrict__ noise, long long rows, int D,
                                                               int Ls, int family, float* __restrict__ dz,
                                                               float* __restrict__ dprior_sd) {
  const int S = D - Ls;
  if ((int)blockIdx.x < S) {
    const int j = blockIdx.x;
    float acc = 0.f;
    for (long long r = threadIdx.x; r < rows; r += 256) acc += dzc[r * D + Ls + j] * lat_t(family, noise[r * S + j]);
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && dprior_sd) dprior_sd[j] = (red[0] + red[1]) + (red[2] + red[3]);
    return;
  }
  const long long i = ((long long)blockIdx.x - S) * 256 + threadIdx.x;
  if (i >= rows * D) return;
  const int l = (int)(i % D);
  dz[i] = l < Ls ? dzc[i] : 0.f;
}

// ---- importance-sampled joint likelihood (compute_joint_nll) -------------------------------------------------
// ln p(x_b) ~= logsumexp_k [ sum_m ln p(x_m | z_kb) + ln p(z_kb) - ln q(z_kb | x_b) ] - ln K with q a uniform mixture
// of E experts (the S subset posteriors of MoPoE, mopoe_model.py:467-594; the M unimodal posteriors of MMVAE,
// mmvae_model.py:365-443; a single joint posterior for MVTCAE, mvtcae_model.py:213-291, and JMVAE,
// joint_model.py:82-154).  The reference walks data points and K-chunks in Python; here the K axis is the leading
// axis of one [K,B] problem (the decoders and mvk_recon_nll_fwd produce the likelihood rows).
constexpr int MAXE = MVK_IWAE_MAX_EXPERTS;
struct IwaePtrs {
  const float* rows[MAXM];
  const float* loc[MAXE];
  const float* sd[MAXE];
};

__global__ __launch_bounds__(256) void iwae_sample_kernel(const float* __restrict__ loc, const float* __restrict__ sd,
                                                          const float* __restrict__ noise, long long n, int BL,
                                                          int family, float* __restrict__ z) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int j = (int)(i % BL);
  z[i] = loc[j] + sd[j] * lat_t(family, noise[i]);
}

// one thread per (k,b): the expert parameters of row b are shared by the K threads of that column (L1 / L2 hits)
__global__ __launch_bounds__(256) void iwae_logw_kernel(IwaePtrs a, const float* __restrict__ z,
                                                        const float* __restrict__ prior_loc,
                                                        const float* __restrict__ prior_sd, int K, int B, int L, int E,
                                                        int R, int family, float* __restrict__ lw) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)K * B) return;
  const int b = (int)(i % B);
  const float* zr = z + i * L;
  float lpz = 0.f;
  for (int l = 0; l < L; ++l)
    lpz += lat_logp(family, zr[l], prior_loc ? prior_loc[l] : 0.f, prior_sd ? prior_sd[l] : 1.f);
  float m = -INFINITY, s = 0.f;  // running logsumexp over the experts
  for (int e = 0; e < E; ++e) {
    const float* lo = a.loc[e] + (long long)b * L;
    const float* sd = a.sd[e] + (long long)b * L;
    float lq = 0.f;
    for (int l = 0; l < L; ++l) lq += lat_logp(family, zr[l], lo[l], sd[l]);
    if (lq > m) {
      s = s * expf(m - lq) + 1.f;
      m = lq;
    } else if (lq > -INFINITY) {
      s += expf(lq - m);
    }
  }
  const float lqz = m + logf(s) - logf((float)E);
  float lpx = 0.f;
  for (int r = 0; r < R; ++r) lpx -= a.rows[r][i];
  lw[i] = lpx + lpz - lqz;
}

// ll[b] = logsumexp over the n arrays x K samples of lw[.][k,b] - ln(n K); one wave per data point
struct IwaeLw {
  const float* lw[MAXM];
};
__global__ __launch_bounds__(256) void iwae_reduce_kernel(IwaeLw a, int n, int K, int B, float* __restrict__ ll) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float m = -INFINITY;
  for (int j = 0; j < n; ++j)
    for (int k = lane; k < K; k += 64) m = fmaxf(m, a.lw[j][(long long)k * B + b]);
  m = wave_max(m);
  float s = 0.f;
  if (m > -INFINITY)
    for (int j = 0; j < n; ++j)
      for (int k = lane; k < K; k += 64) s += expf(a.lw[j][(long long)k * B + b] - m);
  s = wave_sum(s);
  if (lane == 0) ll[b] = m + logf(s) - logf((float)n * (float)K);
}

}  // namespace

extern "C" {

int mvk_mmvaeplus_cross_latent_fwd(const float* z, const float* prior_sd, const float* noise, int64_t rows, int D,
                                   int Ls, int family, float* zc, void* stream) {
  if (!z || !prior_sd || !noise || !zc || rows < 0 || D < 2 || Ls < 1 || Ls >= D || family < 0 || family > 1)
    return MVK_EINVAL;
  if (rows == 0) return MVK_OK;
  hipLaunchKernelGGL(cross_latent_fwd_kernel, dim3((unsigned)((rows * D + 255) / 256)), dim3(256), 0,
                     mvk_stream(stream), z, prior_sd, noise, (long long)rows, D, Ls, family, zc);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_mmvaeplus_cross_latent_bwd(const float* dzc, const float* noise, int64_t rows, int D, int Ls, int family,
                                   float* dz, float* dprior_sd, void* stream) {
  if (!dzc || !noise || !dz || rows < 0 || D < 2 || Ls < 1 || Ls >= D || family < 0 || family > 1) return MVK_EINVAL;
  if (rows == 0) return MVK_OK;
  const unsigned blocks = (unsigned)(D - Ls) + (unsigned)((rows * D + 255) / 256);
  hipLaunchKernelGGL(cross_latent_bwd_kernel, dim3(blocks), dim3(256), 0, mvk_stream(stream), dzc, noise,
                     (long long)rows, D, Ls, family, dz, dprior_sd);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_mmvae_std_fwd(const float* lv, int rows, int L, int family, float* sd, void* stream) {
  if (!lv || !sd || rows < 0 || L < 1 || family < 0 || family > 2) return MVK_EINVAL;
  if (rows == 0) return MVK_OK;
  hipLaunchKernelGGL(std_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, mvk_stream(stream), lv, rows, L, family, sd);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_mmvae_std_bwd(const float* lv, const float* sd, const float* dsd, int rows, int L, int family, float* dlv,
                      void* stream) {
  if (!lv || !sd || !dsd || !dlv || rows < 0 || L < 1 || family < 0 || family > 2) return MVK_EINVAL;
  if (rows == 0) return MVK_OK;
  hipLaunchKernelGGL(std_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, mvk_stream(stream), lv, sd, dsd, rows, L,
                     family, dlv);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_mmvae_latent_fwd(const float* const* mu, const float* const* sd, const float* const* noise,
                         const uint8_t* const* masks, const float* prior_mean, const float* prior_sd, int M, int K,
                         int B, int L, int family, float* const* z, float* const* lpz, float* const* lqz,
                         float* const* lq_all, int shared_dims, float* const* lqw, void* stream) {
  if (!mu || !sd || !noise || !prior_mean || !prior_sd || !z || !lpz || !lqz || !lq_all || M < 1 || M > MAXM ||
      K < 1 || L < 1 || family < 0 || family > 1 || shared_dims < 1 || shared_dims > L || (shared_dims < L && !lqw))
    return MVK_EINVAL;
  if (B <= 0) return B == 0 ? MVK_OK : MVK_EINVAL;
  MmPtrs p{};
  for (int m = 0; m < M; ++m) {
    if (!mu[m] || !sd[m] || !noise[m] || !z[m] || !lpz[m] || !lqz[m] || !lq_all[m]) return MVK_EINVAL;
    p.mu[m] = mu[m];
    p.sd[m] = sd[m];
    p.noise[m] = noise[m];
    p.mask[m] = masks ? masks[m] : nullptr;
    p.z[m] = z[m];
    p.lpz[m] = lpz[m];
    p.lqz[m] = lqz[m];
    p.lq_all[m] = lq_all[m];
    p.lqw[m] = lqw ? lqw[m] : nullptr;
  }
  long long rows = (long long)M * K * B;
  hipLaunchKernelGGL(latent_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, mvk_stream(stream), p,
                     prior_mean, prior_sd, M, K, B, L, shared_dims, family);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_mmvae_objective_fwd(const float* const* rows, const float* const* lpz, const float* const* lqz,
                            const uint8_t* const* masks, int M, int K, int B, int dreg, float* const* lw_out,
                            float* const* w, float* const* rowcoef, float* loss, const float* const* lqw, float beta,
                            void* stream) {
  if (!rows || !lpz || !lqz || !lw_out || !w || !rowcoef || !loss || M < 1 || M > MAXM || K < 1 || B < 1) return MVK_EINVAL;
  ObjPtrs p{};
  for (int m = 0; m < M; ++m) {
    if (!lpz[m] || !lqz[m] || !lw_out[m] || !w[m] || !rowcoef[m]) return MVK_EINVAL;
    p.rowcoef[m] = rowcoef[m];
    for (int r = 0; r < M; ++r) {
      if (!rows[m * M + r]) return MVK_EINVAL;
      p.rows[m * MAXM + r] = rows[m * M + r];
    }
    p.lpz[m] = lpz[m];
    p.lqz[m] = lqz[m];
    p.lqw[m] = lqw ? lqw[m] : nullptr;
    p.mask[m] = masks ? masks[m] : nullptr;
    p.lw[m] = lw_out[m];
    p.w[m] = w[m];
  }
  hipLaunchKernelGGL(objective_kernel, dim3(1), dim3(1024), 0, mvk_stream(stream), p, M, K, B, dreg, beta, loss);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_mmvae_latent_bwd(const float* const* mu, const float* const* sd, const float* const* noise,
                         const float* const* z, const uint8_t* const* masks, const float* prior_mean,
                         const float* prior_sd, const float* const* w, const float* const* lq_all,
                         const float* const* lqz, const float* const* dz_dec, int M, int K, int B, int L, int family,
                         int dreg, const float* gscale, float* const* dmu, float* const* dsd, float* dprior_sd,
                         int shared_dims, float beta, void* stream) {
  if (!mu || !sd || !noise || !z || !prior_mean || !prior_sd || !w || !lq_all || !lqz || !dz_dec || !dmu || !dsd ||
      M < 1 || M > MAXM || K < 1 || L < 1 || family < 0 || family > 1 || shared_dims < 1 || shared_dims > L)
    return MVK_EINVAL;
  if (B <= 0) return B == 0 ? MVK_OK : MVK_EINVAL;
  BwdPtrs p{};
  for (int m = 0; m < M; ++m) {
    if (!mu[m] || !sd[m] || !noise[m] || !z[m] || !w[m] || !lq_all[m] || !lqz[m] || !dz_dec[m] || !dmu[m] || !dsd[m])
      return MVK_EINVAL;
    p.mu[m] = mu[m];
    p.sd[m] = sd[m];
    p.noise[m] = noise[m];
    p.z[m] = z[m];
    p.mask[m] = masks ? masks[m] : nullptr;
    p.w[m] = w[m];
    p.lq_all[m] = lq_all[m];
    p.lqz[m] = lqz[m];
    p.dz_dec[m] = dz_dec[m];
    p.dmu[m] = dmu[m];
    p.dsd[m] = dsd[m];
  }
  hipStream_t s = mvk_stream(stream);
  hipLaunchKernelGGL(latent_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, s, p, prior_mean, prior_sd, M, K, B, L,
                     shared_dims, beta, family, dreg, gscale, dprior_sd);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_iwae_sample(const float* loc, const float* sd, const float* noise, int K, int B, int L, int family, float* z,
                    void* stream) {
  if (K == 0 || B == 0) return MVK_OK;
  if (!loc || !sd || !noise || !z || K < 0 || B < 0 || L < 1 || family < 0 || family > 2) return MVK_EINVAL;
  const long long n = (long long)K * B * L;
  hipLaunchKernelGGL(iwae_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, mvk_stream(stream), loc, sd,
                     noise, n, B * L, family == MVK_FAMILY_LAPLACE_SOFTMAX ? MVK_FAMILY_LAPLACE_SOFTMAX : MVK_FAMILY_NORMAL, z);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_iwae_logw(const float* z, const float* const* rows, int n_rows, const float* const* loc,
                  const float* const* sd, int E, const float* prior_loc, const float* prior_sd, int K, int B, int L,
                  int family, float* lw, void* stream) {
  if (!z || !loc || !sd || !lw || (n_rows > 0 && !rows) || n_rows < 0 || n_rows > MAXM || E < 1 || E > MAXE || K < 0 ||
      B < 0 || L < 1 || family < 0 || family > 2)
    return MVK_EINVAL;
  IwaePtrs a{};
  for (int r = 0; r < n_rows; ++r) {
    if (!rows[r]) return MVK_EINVAL;
    a.rows[r] = rows[r];
  }
  for (int e = 0; e < E; ++e) {
    if (!loc[e] || !sd[e]) return MVK_EINVAL;
    a.loc[e] = loc[e];
    a.sd[e] = sd[e];
  }
  const long long n = (long long)K * B;
  if (n == 0) return MVK_OK;
  hipLaunchKernelGGL(iwae_logw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, mvk_stream(stream), a, z,
                     prior_loc, prior_sd, K, B, L, E, n_rows,
                     family == MVK_FAMILY_LAPLACE_SOFTMAX ? MVK_FAMILY_LAPLACE_SOFTMAX : MVK_FAMILY_NORMAL, lw);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_iwae_reduce(const float* const* lw, int n, int K, int B, float* ll, void* stream) {
  if (!lw || !ll || n < 1 || n > MAXM || K < 1 || B < 0) return MVK_EINVAL;
  IwaeLw a{};
  for (int j = 0; j < n; ++j) {
    if (!lw[j]) return MVK_EINVAL;
    a.lw[j] = lw[j];
  }
  if (B == 0) return MVK_OK;
  hipLaunchKernelGGL(iwae_reduce_kernel, dim3((B + 3) / 4), dim3(256), 0, mvk_stream(stream), a, n, K, B, ll);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

}  // extern "C"
