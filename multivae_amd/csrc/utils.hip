// The reference's public helper functions (multivae/models/base/base_utils.py:28-172) as stand-alone HIP kernels:
// poe / stable_poe, kl_divergence, the per-element decoder log-probabilities of set_decoder_dist and cross_entropy.
// The training path uses the fused forms of elbo.hip; these entry points serve user code written against the helpers.
// All tensors fp32, contiguous; element-wise kernels, one thread per output element (latency-bound sizes).
#include "common.hpp"

namespace {

// ---- poe (base_utils.py:122-130) and stable_poe (:133-147) over E experts ---------------------------------------------
// mode 0: T = 1 / (exp(lv) + eps), mu = sum(mu T) / sum T, lv = log(1 / sum T)
// mode 1: log-sum-exp form without eps (an expert with lv = +inf has weight exactly 0)
__global__ void poe_fwd_kernel(const float* __restrict__ mus, const float* __restrict__ lvs, int E, long long n, float eps,
                               int mode, float* __restrict__ mu, float* __restrict__ lv) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (mode == 0) {
    float st = 0.f, sm = 0.f;
    for (int e = 0; e < E; ++e) {
      const float T = 1.0f / (expf(lvs[e * n + i]) + eps);
      st += T;
      sm += mus[e * n + i] * T;
    }
    mu[i] = sm / st;
    lv[i] = logf(1.0f / st);
  } else {
    float m = -INFINITY;
    for (int e = 0; e < E; ++e) m = fmaxf(m, -lvs[e * n + i]);
    float s = 0.f;
    for (int e = 0; e < E; ++e) s += expf(-lvs[e * n + i] - m);
    const float lnv = -(m + logf(s));
    float acc = 0.f;
    for (int e = 0; e < E; ++e) acc += expf(-lvs[e * n + i]) * mus[e * n + i];
    mu[i] = E == 1 ? mus[i] : acc * expf(lnv);
    lv[i] = E == 1 ? lvs[i] : lnv;
  }
}

// d mu_e = g_mu T_e / S;  d T_e = g_mu (mu_e - mu) / S - g_lv / S;  d lv_e = d T_e * dT/dlv = -d T_e exp(lv_e) T_e^2
__global__ void poe_bwd_kernel(const float* __restrict__ mus, const float* __restrict__ lvs, int E, long long n, float eps,
                               int mode, const float* __restrict__ gmu, const float* __restrict__ glv,
                               float* __restrict__ dmus, float* __restrict__ dlvs) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gm = gmu ? gmu[i] : 0.f, gl = glv ? glv[i] : 0.f;
  if (mode == 1 && E == 1) {
    dmus[i] = gm;
    dlvs[i] = gl;
    return;
  }
  // weights w_e = T_e / S in a form that is safe for lv = +inf (mode 1) and for the eps form (mode 0)
  float m = -INFINITY;
  for (int e = 0; e < E; ++e) {
    const float l = lvs[e * n + i];
    m = fmaxf(m, mode == 0 ? -logf(expf(l) + eps) : -l);
  }
  float s = 0.f, mu = 0.f;
  for (int e = 0; e < E; ++e) {
    const float l = lvs[e * n + i];
    const float w = expf((mode == 0 ? -logf(expf(l) + eps) : -l) - m);
    s += w;
    mu += w * mus[e * n + i];
  }
  mu /= s;
  for (int e = 0; e < E; ++e) {
    const float l = lvs[e * n + i];
    const float lnT = mode == 0 ? -logf(expf(l) + eps) : -l;
    const float w = expf(lnT - m) / s;  // T_e / S
    dmus[e * n + i] = gm * w;
    // dT_e / S-normalised: (g_mu (mu_e - mu) - g_lv) * w; times dlnT/dlv = -exp(l) T (mode 0) or -1 (mode 1)
    const float dlnT = mode == 0 ? -expf(l) * expf(lnT) : -1.0f;
    const float v = (gm * (mus[e * n + i] - mu) - gl) * w * dlnT;
    dlvs[e * n + i] = w == 0.f ? 0.f : v;
  }
}

// ---- kl_divergence (base_utils.py:90-119): rows of 1/2 (plv - lv + exp(lv - plv) + (mean - pmean)^2 / exp(plv) - 1) -----
// every operand is indexed modulo its own element count (trailing-dimension broadcasting)
__global__ void kl_fwd_kernel(const float* __restrict__ mean, long long nm, const float* __restrict__ lv, long long nl,
                              const float* __restrict__ pm, long long npm, const float* __restrict__ plv, long long npl,
                              long long rows, int L, float* __restrict__ kl) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float s = 0.f;
  for (int l = 0; l < L; ++l) {
    const long long i = r * L + l;
    const float a = mean[i % nm], b = lv[i % nl], c = pm[i % npm], d = plv[i % npl];
    s += 0.5f * (d - b + expf(b - d) + ((a - c) * (a - c)) / expf(d) - 1.0f);
  }
  kl[r] = s;
}
// full-shape partial derivatives [rows, L] (broadcast operands are column-summed by the caller)
__global__ void kl_bwd_kernel(const float* __restrict__ mean, long long nm, const float* __restrict__ lv, long long nl,
                              const float* __restrict__ pm, long long npm, const float* __restrict__ plv, long long npl,
                              long long rows, int L, const float* __restrict__ g, float* __restrict__ dmean,
                              float* __restrict__ dlv, float* __restrict__ dpm, float* __restrict__ dplv) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * L) return;
  const float a = mean[i % nm], b = lv[i % nl], c = pm[i % npm], d = plv[i % npl];
  const float gr = g[i / L], inv = 1.0f / expf(d), e = expf(b - d), diff = a - c;
  if (dmean) dmean[i] = gr * diff * inv;
  if (dlv) dlv[i] = gr * 0.5f * (e - 1.0f);
  if (dpm) dpm[i] = -gr * diff * inv;
  if (dplv) dplv[i] = gr * 0.5f * (1.0f - e - diff * diff * inv);
}

// ---- set_decoder_dist log-probabilities (base_utils.py:62-87), element-wise; target broadcast over leading dims -------
__device__ __forceinline__ float log_sigmoid(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }
__global__ void logprob_fwd_kernel(const float* __restrict__ r, const float* __restrict__ x, long long n, long long nx,
                                   int dist, float scale, float* __restrict__ lp) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = r[i], t = x[i % nx];
  float v;
  if (dist == MVK_DIST_NORMAL) {
    const float d = t - a;
    v = -(d * d) / (2.0f * scale * scale) - logf(scale) - 0.91893853320467274178f;
  } else if (dist == MVK_DIST_LAPLACE) {
    v = -logf(2.0f * scale) - fabsf(t - a) / scale;
  } else {  // Bernoulli(logits = r)
    v = t * log_sigmoid(a) + (1.0f - t) * log_sigmoid(-a);
  }
  lp[i] = v;
}
__global__ void logprob_bwd_kernel(const float* __restrict__ r, const float* __restrict__ x, long long n, long long nx,
                                   int dist, float scale, const float* __restrict__ g, float* __restrict__ dr) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = r[i], t = x[i % nx];
  float v;
  if (dist == MVK_DIST_NORMAL) v = (t - a) / (scale * scale);
  else if (dist == MVK_DIST_LAPLACE) v = (t > a ? 1.0f : (t < a ? -1.0f : 0.f)) / scale;
  else v = t - 1.0f / (1.0f + expf(-a));
  dr[i] = g[i] * v;
}
// cross_entropy (base_utils.py:28-57): x * log_softmax(r + eps) over the last dimension C; one wave per class row
__global__ __launch_bounds__(256) void xent_kernel(const float* __restrict__ r, const float* __restrict__ x, long long rows,
                                                   long long xrows, int C, float eps, const float* __restrict__ g,
                                                   float* __restrict__ out) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* rr = r + row * C;
  const float* xx = x + (row % xrows) * C;
  float m = -INFINITY;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, rr[c] + eps);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += expf(rr[c] + eps - m);
  s = wave_sum(s);
  const float lse = m + logf(s);
  if (!g) {  // forward: element-wise x * log_softmax
    for (int c = lane; c < C; c += 64) out[row * C + c] = xx[c] * (rr[c] + eps - lse);
  } else {  // backward: d r_c = g_c x_c - softmax_c sum_c' g_c' x_c'
    float gx = 0.f;
    for (int c = lane; c < C; c += 64) gx += g[row * C + c] * xx[c];
    gx = wave_sum(gx);
    for (int c = lane; c < C; c += 64) out[row * C + c] = g[row * C + c] * xx[c] - expf(rr[c] + eps - lse) * gx;
  }
}

inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" {

int mvk_poe_fwd(const float* mus, const float* lvs, int E, int64_t n, float eps, int stable, float* mu, float* lv,
                void* stream) {
  if (n == 0) return MVK_OK;
  if (!mus || !lvs || !mu || !lv || E < 1 || n < 0) return MVK_EINVAL;
  hipLaunchKernelGGL(poe_fwd_kernel, dim3(blocks_for(n)), dim3(256), 0, mvk_stream(stream), mus, lvs, E, (long long)n, eps,
                     stable ? 1 : 0, mu, lv);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_poe_bwd(const float* mus, const float* lvs, int E, int64_t n, float eps, int stable, const float* gmu,
                const float* glv, float* dmus, float* dlvs, void* stream) {
  if (n == 0) return MVK_OK;
  if (!mus || !lvs || !dmus || !dlvs || E < 1 || n < 0) return MVK_EINVAL;
  hipLaunchKernelGGL(poe_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, mvk_stream(stream), mus, lvs, E, (long long)n, eps,
                     stable ? 1 : 0, gmu, glv, dmus, dlvs);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_kl_gauss_fwd(const float* mean, int64_t n_mean, const float* lv, int64_t n_lv, const float* pmean, int64_t n_pmean,
                     const float* plv, int64_t n_plv, int64_t rows, int L, float* kl, void* stream) {
  if (rows == 0) return MVK_OK;
  if (!mean || !lv || !pmean || !plv || !kl || rows < 0 || L < 1 || n_mean < 1 || n_lv < 1 || n_pmean < 1 || n_plv < 1)
    return MVK_EINVAL;
  hipLaunchKernelGGL(kl_fwd_kernel, dim3(blocks_for(rows)), dim3(256), 0, mvk_stream(stream), mean, (long long)n_mean, lv,
                     (long long)n_lv, pmean, (long long)n_pmean, plv, (long long)n_plv, (long long)rows, L, kl);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_kl_gauss_bwd(const float* mean, int64_t n_mean, const float* lv, int64_t n_lv, const float* pmean, int64_t n_pmean,
                     const float* plv, int64_t n_plv, int64_t rows, int L, const float* g, float* dmean, float* dlv,
                     float* dpmean, float* dplv, void* stream) {
  if (rows == 0) return MVK_OK;
  if (!mean || !lv || !pmean || !plv || !g || rows < 0 || L < 1) return MVK_EINVAL;
  hipLaunchKernelGGL(kl_bwd_kernel, dim3(blocks_for(rows * L)), dim3(256), 0, mvk_stream(stream), mean, (long long)n_mean,
                     lv, (long long)n_lv, pmean, (long long)n_pmean, plv, (long long)n_plv, (long long)rows, L, g, dmean,
                     dlv, dpmean, dplv);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_logprob_fwd(const float* recon, const float* target, int64_t n, int64_t n_target, int dist, float scale, int C,
                    float eps, float* lp, void* stream) {
  if (n == 0) return MVK_OK;
  if (!recon || !target || !lp || n < 0 || n_target < 1 || dist < 0 || dist > MVK_DIST_CATEGORICAL) return MVK_EINVAL;
  if (dist == MVK_DIST_CATEGORICAL) {
    if (C < 1 || n % C || n_target % C) return MVK_EINVAL;
    hipLaunchKernelGGL(xent_kernel, dim3((unsigned)((n / C + 3) / 4)), dim3(256), 0, mvk_stream(stream), recon, target,
                       (long long)(n / C), (long long)(n_target / C), C, eps, (const float*)nullptr, lp);
  } else {
    hipLaunchKernelGGL(logprob_fwd_kernel, dim3(blocks_for(n)), dim3(256), 0, mvk_stream(stream), recon, target,
                       (long long)n, (long long)n_target, dist, scale, lp);
  }
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_logprob_bwd(const float* recon, const float* target, int64_t n, int64_t n_target, int dist, float scale, int C,
                    float eps, const float* g, float* drecon, void* stream) {
  if (n == 0) return MVK_OK;
  if (!recon || !target || !g || !drecon || n < 0 || n_target < 1 || dist < 0 || dist > MVK_DIST_CATEGORICAL)
    return MVK_EINVAL;
  if (dist == MVK_DIST_CATEGORICAL) {
    if (C < 1 || n % C || n_target % C) return MVK_EINVAL;
    hipLaunchKernelGGL(xent_kernel, dim3((unsigned)((n / C + 3) / 4)), dim3(256), 0, mvk_stream(stream), recon, target,
                       (long long)(n / C), (long long)(n_target / C), C, eps, g, drecon);
  } else {
    hipLaunchKernelGGL(logprob_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, mvk_stream(stream), recon, target,
                       (long long)n, (long long)n_target, dist, scale, g, drecon);
  }
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

}  // extern "C"
