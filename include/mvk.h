/*
 * mvk.h — C ABI of libmvk.so: hand-written HIP (gfx950 / CDNA4) kernels for the multimodal-VAE
 * training hot path (MMVAE / MoPoE / MVTCAE forward + ELBO + backward + Adam).
 *
 * The reference (AgatheSenellart/MultiVae) has no FFI: its "operator API" is Python.
 * Each entry point below therefore names the reference Python function(s) it replaces (file:line,
 * relative to /root/reference/src/multivae).  Host code (multivae_amd/, Python on PyTorch-ROCm) binds
 * these through ctypes; INTEGRATION.md shows the binding a MultiVae maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless stated otherwise; the caller owns all
 *     memory; the library never allocates, frees, retains a pointer past the call, or synchronises;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it; calls are re-entrant per stream;
 *   - return value: 0 = MVK_OK, <0 = error (MVK_EINVAL bad argument, MVK_ELAUNCH launch failure);
 *   - noise is always an input buffer: results never depend on an RNG stream;
 *   - gradient outputs documented "+=" are ACCUMULATED into the caller's buffer: zero it first if needed.  Every
 *     reduction is deterministic: split-K slices, column-sum partials and per-workgroup weight-gradient slabs are
 *     written to caller-owned scratch and added in a fixed order (an entry point that is given no scratch falls back
 *     to fp32 atomicAdd, the only non-reproducible path);
 *   - there is no hidden global state, with TWO documented exceptions, both explicit begin / end pairs that the host
 *     owns and that are per device: (1) mvk_defer_begin / mvk_defer_flush / mvk_defer_end — between begin and end the
 *     ordered finishes of leaf gradients that target the registered flat gradient buffer are queued in a caller-owned
 *     arena and run as one launch at flush; (2) mvk_prof_enable / mvk_prof_disable — device-timestamp records of
 *     the launches in between are written to a caller-owned buffer.  Neither survives its end call; with neither
 *     active every entry point is a pure function of its arguments.
 */
#ifndef MVK_H
#define MVK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVK_OK 0
#define MVK_EINVAL (-1)
#define MVK_ELAUNCH (-2)

#define MVK_MAX_MODALITIES 8
#define MVK_MVAE_MAX_SUBSETS 32  /* subsets of one MVAE objective (joint + unimodal + k random ones) */
#define MVK_IWAE_MAX_EXPERTS 32 /* experts of the mixture posterior in mvk_iwae_logw (MoPoE: 2^5 - 1 subsets) */

/* decoder output distributions — models/base/base_utils.py:62-87 (set_decoder_dist) */
#define MVK_DIST_NORMAL 0
#define MVK_DIST_LAPLACE 1
#define MVK_DIST_BERNOULLI 2
#define MVK_DIST_CATEGORICAL 3 /* x * log_softmax(recon + 1e-6) over the last dimension (base_utils.py:28-57) */

/* activations fused into GEMM/conv epilogues */
#define MVK_ACT_NONE 0
#define MVK_ACT_RELU 1
#define MVK_ACT_SIGMOID 2
#define MVK_ACT_LEAKY02 3 /* LeakyReLU(0.2) (models/nn/mmnist.py:222-246, cub.py:250-293) */

/* latent families of MMVAE — models/mmvae/mmvae_model.py:44-49,66-74 */
#define MVK_FAMILY_NORMAL 0
#define MVK_FAMILY_LAPLACE_SOFTMAX 1
#define MVK_FAMILY_NORMAL_SOFTPLUS 2 /* std kernels only (MMVAE+): std = softplus(lv) + 1e-6, density = Normal */

int mvk_version(void);

/* ------------------------------------------------------------------------------------------------
 * Fused posterior aggregation + reparameterisation + Gaussian KL
 * ------------------------------------------------------------------------------------------------ */

/* MoPoE: per-subset PoE (+N(0,I) expert on the full subset), subset selection, z = mu + exp(lv/2) eps,
 * all-subset weighted KL.  Replaces models/base/base_utils.py:122-130 (poe), :150-172
 * (rsample_from_gaussian), models/mopoe/mopoe_model.py:274-350 (inference), :249-262 (_poe_fusion),
 * :435-465 / :417-433 (mixture component selection, given as `sel`), :108-145 (calc_joint_divergence).
 *   mu, lv        HOST arrays of M device pointers, each [B,L], modalities in PoE summation order
 *                 (the reference stacks a subset's experts in sorted-name order, mopoe_model.py:88-101)
 *   subset_masks  device int32 [S]: bit i set = modality i (position in mu/lv) belongs to the subset;
 *                 subsets in the reference's enumeration order (mopoe_model.py:71-80)
 *   sel           device int32 [B]: subset index whose posterior row b samples from
 *   weights       device [S,B] or NULL (= 1/S)
 *   eps           device [K,B,L]
 * outputs
 *   z [K,B,L]; kld_rows [B] = sum_s w[s,b] KL_s[b]; mus_out, lvs_out [S,B,L] (both or neither);
 *   joint_mu, joint_lv [B,L] (both or neither)
 */
int mvk_mopoe_posterior_fwd(const float* const* mu, const float* const* lv, int M, const int32_t* subset_masks,
                            int S, const int32_t* sel, const float* weights, const float* eps, int K, int B,
                            int L, float* z, float* kld_rows, float* mus_out, float* lvs_out, float* joint_mu,
                            float* joint_lv, void* stream);

/* backward of the above.  dz [K,B,L]; gkld_rows [B] = upstream gradient of kld_rows (NULL = 0).
 * dmu, dlv: HOST arrays of M device pointers [B,L], overwritten. */
int mvk_mopoe_posterior_bwd(const float* const* mu, const float* const* lv, int M, const int32_t* subset_masks,
                            int S, const int32_t* sel, const float* weights, const float* eps, const float* dz,
                            int K, int B, int L, const float* gkld_rows, float* const* dmu, float* const* dlv,
                            void* stream);

/* MVTCAE: PoE over all modalities (no prior expert, eps 1e-8), joint KL and per-modality conditional
 * KLs.  Replaces models/mvtcae/mvtcae_model.py:110-169 (_modality_encode masks, _inference) and the KL
 * arithmetic of :42-108.  masks: HOST array of M device uint8 [B] pointers (entries may be NULL), or NULL.
 * outputs: z [K,B,L]; joint_kl_rows [B]; cond_kl_rows [M,B]; joint_mu, joint_lv [B,L] (both or neither). */
int mvk_mvtcae_posterior_fwd(const float* const* mu, const float* const* lv, const uint8_t* const* masks,
                             int M, const float* eps, int K, int B, int L, float* z, float* joint_kl_rows,
                             float* cond_kl_rows, float* joint_mu, float* joint_lv, void* stream);

/* backward: gjoint_rows [B], gcond_rows [M,B] = upstream gradients of the two KL outputs (NULL = 0). */
int mvk_mvtcae_posterior_bwd(const float* const* mu, const float* const* lv, const uint8_t* const* masks,
                             int M, const float* eps, const float* dz, int K, int B, int L,
                             const float* gjoint_rows, const float* gcond_rows, float* const* dmu,
                             float* const* dlv, void* stream);

/* Diagonal Gaussian q = N(mu, exp(lv)): w[k,b,:] = mu[b,:] + exp(lv[b,:] / 2) eps[k,b,:] and kl_rows[b] =
 * -1/2 sum_l (1 - exp(lv) - mu^2 + lv) = KL(q || N(0,I)): the modality-specific ("style") latents of MoPoE
 * (mopoe_model.py:171-178 rsample, :212-221 style_kld).  mu, lv [B,L]; eps, w [K,B,L].  bwd: dw [K,B,L] (nullable),
 * gkl [B] (nullable) -> dmu, dlv [B,L] (overwritten). */
int mvk_gauss_sample_kl_fwd(const float* mu, const float* lv, const float* eps, int K, int B, int L, float* w,
                            float* kl_rows, void* stream);
int mvk_gauss_sample_kl_bwd(const float* mu, const float* lv, const float* eps, const float* dw, const float* gkl,
                            int K, int B, int L, float* dmu, float* dlv, void* stream);

/* MVAE (models/mvae/mvae_model.py:56-118): for each of the S subsets of the objective (subset_bits: HOST array, bit m
 * = modality m), the product of the AVAILABLE experts of the subset and the N(0,I) prior in the log-sum-exp form of
 * `stable_poe` (base_utils.py:133-147; a missing modality has log-variance +inf, mvae_model.py:71-75), one sample
 * z_s = mu_s + exp(lv_s / 2) eps[s] per row and kld_rows[s,b] = -1/2 sum_l (1 + lv_s - mu_s^2 - exp(lv_s)) (:100).
 * z_s is written into the decoder input of every modality of s: zm[m] is [K_m, B, L] with K_m = number of subsets
 * holding m, slab index = rank of s among them (HOST array of M device pointers; so each decoder runs once for all
 * its subsets).  eps [S,B,L]; sub_mu, sub_lv [S,B,L] optional outputs (both or neither).  Rows with no available
 * modality in s give the prior (kld 0, no gradient).  bwd: dzm[m] [K_m,B,L] (entries may be NULL), gkld_rows [S,B]
 * (nullable) -> dmu[m], dlv[m] [B,L] (overwritten). */
int mvk_mvae_posterior_fwd(const float* const* mu, const float* const* lv, const uint8_t* const* masks, int M,
                           const int32_t* subset_bits, int S, const float* eps, int B, int L, float* const* zm,
                           float* kld_rows, float* sub_mu, float* sub_lv, void* stream);
int mvk_mvae_posterior_bwd(const float* const* mu, const float* const* lv, const uint8_t* const* masks, int M,
                           const int32_t* subset_bits, int S, const float* eps, const float* const* dzm, int B, int L,
                           const float* gkld_rows, float* const* dmu, float* const* dlv, void* stream);

/* JMVAE (models/jmvae/jmvae_model.py:133-174): reparameterised sample(s) of the joint encoder's posterior
 * z[k] = joint_mu + exp(joint_lv/2) * eps[k], kld_rows[b] = KL(q(z|X) || N(0,I)) summed over L, and
 * ljm_rows[b] = sum_m KL(q(z|X) || q(z|x_m)) with (mu[m], lv[m]) the unimodal encoders' outputs.
 * bwd: gradients w.r.t. the joint and the unimodal parameters given dz [K,B,L] (may be NULL) and the per-row
 * gradients of the two KL terms (may be NULL = 0). */
int mvk_jmvae_posterior_fwd(const float* joint_mu, const float* joint_lv, const float* const* mu,
                            const float* const* lv, int M, const float* eps, int K, int B, int L, float* z,
                            float* kld_rows, float* ljm_rows, void* stream);
int mvk_jmvae_posterior_bwd(const float* joint_mu, const float* joint_lv, const float* const* mu,
                            const float* const* lv, int M, const float* eps, const float* dz, int K, int B, int L,
                            const float* gkld_rows, const float* gljm_rows, float* djoint_mu, float* djoint_lv,
                            float* const* dmu, float* const* dlv, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused reconstruction NLL over the K-sample axis
 * ------------------------------------------------------------------------------------------------ */
typedef struct mvk_recon_desc {
  const float* recon;   /* [K,B,D] decoder output (logits for bernoulli) */
  const float* x;       /* [B,D] target, broadcast over K */
  const uint8_t* mask;  /* [B] availability or NULL */
  float* rows;          /* out [K,B]: rescale * sum_d -log p(x|recon); NOT masked (fwd only) */
  float* drecon;        /* out [K,B,D] or NULL: coef * mask[b] * rowcoef[k,b] * rescale * d(-log p)/d recon */
  const float* rowcoef; /* [K,B] per-row gradient weight or NULL (= 1) */
  int64_t D;
  int32_t dist;         /* MVK_DIST_* */
  float scale;          /* normal / laplace scale */
  float rescale;        /* likelihood rescaling factor (base_ae_model.py:127-152) */
  float coef;           /* constant gradient weight (e.g. 1/(B*K)) */
  int32_t n_classes;    /* MVK_DIST_CATEGORICAL: size of the last (class) dimension, D % n_classes == 0; else ignored */
} mvk_recon_desc;

/* Per-(k,b) row NLL for up to MVK_MAX_MODALITIES modalities in ONE launch, optionally emitting
 * d(loss)/d(recon) in the same pass.  Replaces models/base/base_utils.py:62-87 (recon_log_probs) and the
 * `.view(B,-1).sum(-1)` row reductions of mopoe_model.py:192-199, mvtcae_model.py:60-68,
 * mmvae_model.py:208-214.  descs: HOST array. */
int mvk_recon_nll_fwd(const mvk_recon_desc* descs, int n_mod, int K, int B, void* stream);

/* Second pass for estimators whose row weights depend on a K-reduction (IWAE/DReG): writes drecon only. */
int mvk_recon_nll_bwd(const mvk_recon_desc* descs, int n_mod, int K, int B, void* stream);

/* Scalar assembly: out[i] = coef[i] * sum_j v_i[j] * (mask_i ? mask_i[j % period_i] : 1) for i < n_terms;
 * out[n_terms] = sum_i lossw[i] * out[i]; out[n_terms+1] = out[n_terms] * loss_sum_scale; *loss_out (nullable)
 * = out[n_terms].  n_terms <= MVK_MAX_TERMS.
 * Replaces the `.mean()` / `.sum()` / `loss = ...` lines of mopoe_model.py:200-227, mvtcae_model.py:96-108. */
#define MVK_MAX_TERMS 64
typedef struct mvk_term_desc {
  const float* v;
  const uint8_t* mask;
  int64_t n;
  int64_t period;
  float coef;
  float lossw;
  float* gfill; /* optional: gfill[0..n) = coef * lossw, the gradient of the loss w.r.t. this term's rows for an upstream
                 * gradient of 1 (terms that enter the loss as a plain weighted row sum: the KL rows) */
} mvk_term_desc;
int mvk_reduce_terms(const mvk_term_desc* terms, int n_terms, float loss_sum_scale, float* out, float* loss_out,
                     void* stream);
/* The same assembly on up to 32 workgroups of 256 threads (the one-workgroup form is latency-bound: 19 us alone for the 170 KB of row sums of
 * the headline step).  ws: caller-owned scratch of >= MVK_REDUCE_TERMS_WS_FLOATS floats whose FIRST word is an arrival counter
 * that must be 0 before the first launch and is 0 again after every launch; not shared between streams.  Every workgroup sums a
 * fixed slice of every term, the last one to arrive adds the partials in workgroup order: deterministic.  ws == NULL (or too
 * small, or no term longer than 4096 entries): exactly mvk_reduce_terms. */
#define MVK_REDUCE_TERMS_WS_FLOATS (1 + 32 * MVK_MAX_TERMS)
int mvk_reduce_terms_ws(const mvk_term_desc* terms, int n_terms, float loss_sum_scale, float* out, float* loss_out, float* ws,
                        int64_t ws_floats, void* stream);

/* buf[i] *= *gscale unless *gscale == 1 (no memory traffic in that case). */
int mvk_scale_by_device_scalar(float* buf, int64_t n, const float* gscale, void* stream);
/* Backward of the loss assembly in one launch (what autograd does for `loss = sum_i w_i * term_i` given d loss):
 * buffers with fill == 0 (the d loss / d recon tensors written by mvk_recon_nll_fwd for an upstream gradient of 1) are
 * multiplied in place by *gscale — skipped on the device when it is 1; buffers with fill != 0 (gradients of KL row
 * tensors) are set to *gscale * coef. */
#define MVK_SEED_MAX 12
typedef struct mvk_seed_desc {
  float* buf;
  int64_t n;
  float coef;
  int32_t fill;
} mvk_seed_desc;
int mvk_loss_backward_seed(const mvk_seed_desc* jobs, int n, const float* gscale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * MMVAE: mixture-of-experts importance weights (IWAE / DReG)
 * ------------------------------------------------------------------------------------------------ */

/* std from log-variance: exp(lv/2), softmax(lv)*L + 1e-6 (models/mmvae/mmvae_model.py:66-74) or
 * softplus(lv) + 1e-6 (models/mmvaePlus/mmvaePlus_model.py:110-120) and its backward.  lv, std: [rows, L]. */
int mvk_mmvae_std_fwd(const float* lv, int rows, int L, int family, float* std, void* stream);
int mvk_mmvae_std_bwd(const float* lv, const float* std, const float* dstd, int rows, int L, int family,
                      float* dlv, void* stream);

/* For every conditioning modality c: z[c] = loc + std * t(noise) (Normal: t = n; Laplace: t = -sign(u) log1p(-|u|),
 * u ~ U(eps-1,1), torch.distributions.Laplace.rsample), lpz[c][k,b] = sum_l log p(z), lq_all[c][m,k,b] =
 * sum_l log q_m(z) (-inf where mask_m[b] == 0) and lqz[c][k,b] = logsumexp_m lq_all - log n_avail[b]
 * (mmvae_model.py:111-123, :160-206).  mu, std [B,L]; noise, z [K,B,L]; lpz, lqz [K,B]; lq_all [M,K,B]:
 * HOST arrays of M device pointers.  prior_mean, prior_std: device [L].  masks: HOST array of M uint8 [B]
 * device pointers (entries may be NULL) or NULL.
 *
 * MMVAE+ (mmvaePlus_model.py:122-262): the latent of a modality is [u (shared_dims), w (L - shared_dims)]; only the
 * first shared_dims dimensions enter the mixture (lq_all, lqz), the rest is scored by the conditioning modality's own
 * posterior: lqw[c][k,b] = sum_{l >= shared_dims} log q_c(z).  MMVAE: shared_dims = L, lqw = NULL. */
int mvk_mmvae_latent_fwd(const float* const* mu, const float* const* std, const float* const* noise,
                         const uint8_t* const* masks, const float* prior_mean, const float* prior_std, int M,
                         int K, int B, int L, int family, float* const* z, float* const* lpz,
                         float* const* lqz, float* const* lq_all, int shared_dims, float* const* lqw, void* stream);

/* lw[c] = (sum_r -rows[c][r] * mask_r + lpz[c] - lqz[c]) * mask_c, softmax weights w[c] over k, and
 * loss = -sum_b (1/n_avail[b]) sum_c obj_c[b] with obj = logsumexp_k lw - log K (IWAE) or sum_k w lw (DReG)
 * (mmvae_model.py:208-292).  rows: HOST array of M*M device pointers, rows[c*M + r] = [K,B] rescaled NLL rows
 * of modality r reconstructed from z[c] (mvk_recon_nll_fwd).  rowcoef[c] [K,B] = d loss / d lw[c] =
 * -w[c] mask_c / n_avail (the per-row weight of mvk_recon_nll_bwd).  loss: device scalar, overwritten. */
/* MMVAE+: lw[c] = (sum_r ... + beta * (lpz[c] - lqz[c] - lqw[c])) * mask_c (mmvaePlus_model.py:262); MMVAE passes
 * lqw = NULL, beta = 1. */
int mvk_mmvae_objective_fwd(const float* const* rows, const float* const* lpz, const float* const* lqz,
                            const uint8_t* const* masks, int M, int K, int B, int dreg, float* const* lw_out,
                            float* const* w, float* const* rowcoef, float* loss, const float* const* lqw, float beta,
                            void* stream);

/* Latent-side backward.  dz_dec[c] [K,B,L]: gradient of the loss w.r.t. z[c] through the decoders (from
 * mvk_recon_nll_bwd with rowcoef = -w[c]/n_avail and the decoders' own backward).  Produces dmu[c], dstd[c]
 * [B,L] and dprior_std [B,L] (nullable): the per-row terms of d loss / d prior_std, which the caller sums over the rows
 * (mvk_colsum_acc: fixed order, bit-reproducible).  DReG: q parameters are detached inside log q and the total gradient
 * reaching z is scaled by w once more (the hook of mmvae_model.py:263-266). */
int mvk_mmvae_latent_bwd(const float* const* mu, const float* const* std, const float* const* noise,
                         const float* const* z, const uint8_t* const* masks, const float* prior_mean,
                         const float* prior_std, const float* const* w, const float* const* lq_all,
                         const float* const* lqz, const float* const* dz_dec, int M, int K, int B, int L,
                         int family, int dreg, const float* gscale, float* const* dmu, float* const* dstd,
                         float* dprior_std, int shared_dims, float beta,
                         void* stream);

/* MMVAE+ cross-modal decoder input (mmvaePlus_model.py:152-172): zc[r, :Ls] = z[r, :Ls] (the shared latent of the
 * conditioning modality) and zc[r, Ls + j] = prior_std[j] * t(noise[r, j]) (a reparameterised sample of the TARGET
 * modality's private prior, mean 0).  rows = K*B, z / zc [rows, D], noise [rows, D - Ls], prior_std [D - Ls].
 * bwd: dz [rows, D] (zero in the private dims) and dprior_std [D - Ls] (nullable), overwritten. */
int mvk_mmvaeplus_cross_latent_fwd(const float* z, const float* prior_std, const float* noise, int64_t rows, int D,
                                   int Ls, int family, float* zc, void* stream);
int mvk_mmvaeplus_cross_latent_bwd(const float* dzc, const float* noise, int64_t rows, int D, int Ls, int family,
                                   float* dz, float* dprior_std, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Importance-sampled joint likelihood (compute_joint_nll; metrics/likelihoods/likelihoods.py:39-56 is the caller)
 * ------------------------------------------------------------------------------------------------ */

/* z[k,b,:] = loc[b,:] + sd[b,:] * t(noise[k,b,:]) (t as in mvk_mmvae_latent_fwd): the K importance samples of every
 * data point, `rsample_from_gaussian(mu, lv, N=K)` (base_utils.py:150-172) / `post_dist(mu, sd).rsample([K])`.
 * loc, sd [B,L]; noise, z [K,B,L]. */
int mvk_iwae_sample(const float* loc, const float* sd, const float* noise, int K, int B, int L, int family, float* z,
                    void* stream);

/* lw[k,b] = -sum_r rows[r][k,b] + sum_l log p(z[k,b,l]) - (logsumexp_e sum_l log q_e(z[k,b,l]) - log E)
 * with q_e = family(loc[e][b,:], sd[e][b,:]) and the prior family(prior_loc, prior_sd) (NULL: 0 / 1).  Replaces the
 * per-data-point, per-chunk bodies of mopoe_model.py:524-588, mmvae_model.py:400-437, mvtcae_model.py:250-284 and
 * joint_model.py:113-148.  rows: HOST array of n_rows device pointers to [K,B] UNRESCALED NLL rows
 * (mvk_recon_nll_fwd with rescale = 1); loc, sd: HOST arrays of E <= MVK_IWAE_MAX_EXPERTS device pointers [B,L];
 * z [K,B,L]; lw [K,B]. */
int mvk_iwae_logw(const float* z, const float* const* rows, int n_rows, const float* const* loc,
                  const float* const* sd, int E, const float* prior_loc, const float* prior_sd, int K, int B, int L,
                  int family, float* lw, void* stream);

/* ll[b] = logsumexp over the n arrays lw[j][k,b] (k < K) - log(n K): the log-mean-exp of the importance weights
 * (the two-level logsumexp over K-chunks of the reference is the same number; MMVAE+ concatenates the weights of its
 * M conditioning modalities, mmvaePlus_model.py:521-525).  lw: HOST array of n <= MVK_MAX_MODALITIES device
 * pointers [K,B]; ll [B]. */
int mvk_iwae_reduce(const float* const* lw, int n, int K, int B, float* ll, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Encoder / decoder layers: implicit GEMM with fp32 operands and results.  Default engine: split-bf16 MFMA
 * (3 bf16 pieces per operand, 6 piece products, fp32 accumulate: fp32-level error); MVK_ENGINE=f32 selects
 * v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains), which also serves small / irregular shapes.
 * ------------------------------------------------------------------------------------------------ */

/* Y[M,N] = act(X[M,K] W[N,K]^T + b[N]) — nn.Linear + activation
 * (models/nn/default_architectures.py:21-72 Encoder_VAE_MLP, :225-258 Decoder_AE_MLP). */
int mvk_linear_fwd(const float* X, const float* W, const float* b, float* Y, int M, int N, int K, int act,
                   float* ws, int64_t ws_floats, void* stream);
/* dX[M,K] = dYpre[M,N] W[N,K] (* act'(prev_out[m,k]) if prev_out != NULL, so the result is directly the
 * previous layer's pre-activation gradient).  dYpre = dY * act'(y_out) when y_out != NULL, else dY.
 * accumulate != 0: dX += (atomic).
 * colsum_acc (may be NULL; requires accumulate == 0): colsum_acc[k] += sum_m dX[m,k] — the previous layer's bias
 * gradient, fused into this launch's epilogue. */
int mvk_linear_bwd_data(const float* dY, const float* W, float* dX, int M, int N, int K, const float* y_out,
                        int y_act, const float* prev_out, int prev_act, int accumulate, float* colsum_acc,
                        float* ws, int64_t ws_floats, void* stream);
/* dW[N,K] += dYpre[M,N]^T X[M,K];  db[N] += colsum(dYpre) (db nullable).
 * Split-K workspace convention (all weight-gradient / accumulate entry points): `ws` is caller-owned scratch of
 * `ws_floats` floats.  When it can hold one [rows x cols] slab per reduction slice the slices are summed by a
 * second kernel in a fixed order (deterministic, no atomics); with ws == NULL or too small the slices fall
 * back to fp32 atomicAdd.  mvk_splitk_workspace_floats() returns a size that is always sufficient. */
int mvk_linear_bwd_weight(const float* dY, const float* X, float* dW, float* db, int M, int N, int K,
                          const float* y_out, int y_act, float* ws, int64_t ws_floats, void* stream);
int64_t mvk_splitk_workspace_floats(int rows, int cols, int reduce_len);
/* dPre[M,N] = dY * act'(Y) and, if db != NULL, db[n] += sum_m dPre[m,n], in one pass over dY and Y (ordered
 * per-workgroup partials in the caller-owned scratch ws).  The output-layer backward of the MLP decoders. */
int mvk_act_bwd_colsum(const float* dY, const float* Y, int act, int M, int N, float* dPre, float* db, float* ws,
                       int64_t ws_floats, void* stream);
/* db[N] += column sums of dY[M,N] (* act'(y_out)): per-workgroup partials in the caller-owned scratch ws, summed in a
 * fixed order (bit-reproducible); ws == NULL (or too small) falls back to fp32 atomics. */
int mvk_colsum_acc(const float* dY, const float* y_out, int y_act, float* db, int M, int N, float* ws, int64_t ws_floats,
                   void* stream);
/* db[c] += sum over n and spatial positions of dY[n,c,hw] (* act'(y_out)) for NCHW tensors. */
int mvk_nchw_channel_sum_acc(const float* dY, const float* y_out, int y_act, float* db, int n, int c, int hw,
                             float* ws, int64_t ws_floats, void* stream);
/* In-place dY *= act'(Y) (sigmoid: y(1-y), relu: y>0). */
int mvk_act_bwd(float* dY, const float* Y, int64_t n, int act, void* stream);
/* out = a * g * act'(Y) in one pass (out may alias g): the gradient through `a * act(.)`, e.g. the 0.1 of a ResNet block's
 * residual branch and the activation behind its second convolution (models/nn/mmnist.py:229-246). */
int mvk_act_bwd_scaled(const float* g, float a, const float* Y, int act, float* out, int64_t n, void* stream);

/* Row-major GEMM C[M,N] (+)= op(A) op(B) used for the packed 1x1-spatial layers:
 *   ta=0: A is [M,K]; ta=1: A is [K,M].  tb=0: B is [K,N]; tb=1: B is [N,K].
 *   bias[n % bias_mod] added when bias != NULL; act applied; a_act_src: A *= a_act'(a_act_src) on load;
 *   c_act_src: result *= c_act'(c_act_src[m,n]); accumulate != 0 => C += (split-K, atomic). */
int mvk_gemm(const float* A, const float* B, float* C, int M, int N, int K, int ta, int tb, const float* bias,
             int bias_mod, int act, int accumulate, const float* a_act_src, int a_act, const float* c_act_src,
             int c_act, float* ws, int64_t ws_floats, void* stream);

/* Pre-split ("bf3") tensor format of the split-bf16 GEMM engine: x = p0 + p1 + p2 with p0 = bf16(x),
 * p1 = bf16(x - p0), p2 = bf16(x - p0 - p1) (round to nearest); stored as three planes of n bf16 values,
 * plane i at byte offset i * 2 * n.  A producer writes it once; every consuming GEMM saves the per-k-tile split. */
#define MVK_FMT_IN_BF3 1
/* mvk_conv4s2_up only: take the tiled engine (three resident workgroups of <= 128 registers per SIMD) even where the
 * register-stationary kernel (one 512-register wave per SIMD) covers the shape — for a launch that has to run BESIDE such a
 * kernel of another stream instead of behind it (the convolutional encoder's backward-data launches in the step's tail). */
#define MVK_FMT_TILED 2
int mvk_f32_to_bf3(const float* x, int64_t n, void* planes, void* stream);
int mvk_bf3_to_f32(const void* planes, int64_t n, float* x, void* stream);

/* 4x4 / stride 2 / pad 1 convolution pair on NHWC activations (models/nn/svhn.py:7-70).
 * "U" is the large feature map [n,2h,2w,Cu], "V" the small one [n,h,w,Cv], W the reference weight
 * tensor indexed [Cv][Cu][4][4] (nn.Conv2d weight [out,in,kh,kw] with V = output; nn.ConvTranspose2d
 * weight [in,out,kh,kw] with V = input).
 *   pack:  Wdown[(kh*4+kw)*Cu + cu][col_off + cv] (row stride ld_down) and/or
 *          Wup[ph*2+pw][((a*2+b)*Cv + cv)][cu], kh = (1-ph)+2a, kw = (1-pw)+2b
 *   down:  V = act(conv(U) + b) (* v_act'(v_act_src))   — Conv2d forward / ConvTranspose2d backward-data
 *   up:    U = act(convT(V) + b) (* u_act'(u_act_src))  — ConvTranspose2d forward / Conv2d backward-data
 *   wgrad: dWref[Cv][Cu][4][4] += sum_pos U(gathered) V — both layer types
 * u_nchw != 0: U is stored NCHW (network input / output boundary tensors only).
 * u_act_src (down, wgrad): U is a gradient tensor that is multiplied by u_act'(u_act_src) while loading.
 * fmt (down, up): MVK_FMT_IN_BF3 — the gathered input (U for down, V for up) is a pre-split tensor (see
 *   mvk_f32_to_bf3): three bf16 planes [3][n*H*W*C]; the GEMM then stages it without any conversion work.
 * colsum_acc (down, up; may be NULL): colsum_acc[c] += sum over positions of the stored output — the bias gradient
 *   of the layer whose output gradient this launch produces (replaces a separate pass over that tensor; uses the
 *   caller-owned scratch ws for per-workgroup partials, reduced in a fixed order). */
int mvk_pack_conv4s2_weight(const float* Wref, int Cv, int Cu, float* Wdown, int ld_down, int col_off,
                            float* Wup, void* stream);
/* All weight packs of one network in ONE launch.  kind 0: as mvk_pack_conv4s2_weight (Wdown and/or Wup, ld_down
 * <= 0 means Cv); kind 1: as mvk_pack_unflatten_weight with Cv = Cin, Cu = Cout, destination in Wup; kind 2: 3x3
 * convolution weight Wref[Cv = Cout][Cu = Cin][3][3] -> Wdown[(tap*Cu + cu)][cv] (forward operand of mvk_conv3x3) and/or
 * Wup[((8-tap)*Cv + cv)][cu] (its backward-data operand: flipped window, channels swapped). */
#define MVK_PACK_MAX 16
typedef struct mvk_pack_desc {
  const float* Wref;
  float* Wdown;
  float* Wup;
  int Cv, Cu, ld_down, col_off, kind;
  void* Fdown; /* kind 0, optional: the weights as bf16-piece MFMA fragments in the order the register-stationary */
  void* Fup;   /* convolution kernels (csrc/imgconv.hip) load them: [role][k-step 16][piece 3][lane 64][8 bf16],  */
               /* mvk_imgconv_frag_bytes(Cu, Cv) bytes each; passed to mvk_conv4s2_down / _up as `wfrag` */
  float* amax; /* optional: receives max |Wref| (atomic max: must hold 0 before the launch) — the w_amax of mvk_conv3x3_s */
} mvk_pack_desc;
int64_t mvk_imgconv_frag_bytes(int Cu, int Cv); /* 0: this channel pair has no register-stationary kernel */
int mvk_pack_weights(const mvk_pack_desc* jobs, int n, void* stream);
int mvk_conv4s2_down(const float* U, const float* Wdown, const float* bias, float* V, int n, int h, int w,
                     int Cu, int Cv, int act, int u_nchw, const float* u_act_src, int u_act,
                     const float* v_act_src, int v_act, float* colsum_acc, float* ws, int64_t ws_floats, int fmt,
                     const void* wfrag, void* stream);
int mvk_conv4s2_up(const float* V, const float* Wup, const float* bias, float* U, int n, int h, int w, int Cu,
                   int Cv, int act, int u_nchw, const float* u_act_src, int u_act, float* colsum_acc, float* ws,
                   int64_t ws_floats, int fmt, const void* wfrag, void* stream);
/* The two launches above on the register-stationary kernels (csrc/imgconv.hip) with the amax protocol of mvk_conv3x3_s: x_amax
 * + w_amax (both or neither) select the scaled-fp16 form (3 fp16 MFMAs per product instead of 6 bf16 ones; the weights are
 * converted in the kernel from the fp32 pack Wdown / Wup, whose max |w| the `amax` slot of mvk_pack_weights delivers), y_amax
 * (optional, zeroed) receives max |result| for the launch that consumes it.  NHWC tensors, no fused input activation.
 * mvk_conv4s2_scaled_ok: 1 when the layer pair and the batch are covered (else both return MVK_EINVAL). */
/* Two producers of the SVHN decoder that publish the maximum of what they write (so that the layer behind them can take the
 * scaled-fp16 form without a pass over the tensor): the short-reduction first layer and the fused-tail backward of the image
 * layer.  Same arguments as mvk_gemm (ta = 0, no fused operand activations) / mvk_conv4s2_small_up_bwd_pre. */
int mvk_gemm_smallk_amax(const float* A, const float* B, float* C, int M, int N, int K, int tb, const float* bias, int bias_mod,
                         int act, float* y_amax, void* stream);
int mvk_conv4s2_small_up_bwd_pre_y(const float* dpre, const float* rowscale, const float* V, int v_act, const float* Wref,
                                   float* dV, float* dWref, float* db, float* db_v, float* ws, int64_t ws_floats, int n, int h,
                                   int w, int Cu, int Cv, float* dv_amax, void* stream);
int mvk_conv4s2_scaled_ok(int n, int h, int w, int Cu, int Cv);
/* mvk_conv4s2_wgrad on scaled fp16 pairs (one accumulator per tile; V carries the 2^11 between main and cross terms in a third
 * piece): u_amax / v_amax = device scalars >= max |U| / max |V|.  NHWC U, no fused activation. */
int mvk_conv4s2_wgrad_scaled_ok(int n, int h, int w, int Cu, int Cv);
int mvk_conv4s2_wgrad_s(const float* U, const float* V, float* dWref, int n, int h, int w, int Cu, int Cv, const float* u_amax,
                        const float* v_amax, float* ws, int64_t ws_floats, void* stream);
int mvk_conv4s2_down_s(const float* U, const float* Wdown, const float* bias, float* V, int n, int h, int w, int Cu, int Cv,
                       int act, const float* v_act_src, int v_act, float* colsum_acc, const float* x_amax,
                       const float* w_amax, float* y_amax, float* ws, int64_t ws_floats, const void* wfrag, void* stream);
int mvk_conv4s2_up_s(const float* V, const float* Wup, const float* bias, float* U, int n, int h, int w, int Cu, int Cv,
                     int act, const float* u_act_src, int u_act, float* colsum_acc, const float* x_amax, const float* w_amax,
                     float* y_amax, float* ws, int64_t ws_floats, const void* wfrag, void* stream);
int mvk_conv4s2_wgrad(const float* U, const float* V, float* dWref, int n, int h, int w, int Cu, int Cv,
                      int u_nchw, const float* u_act_src, int u_act, float* ws, int64_t ws_floats, void* stream);
/* Two of those in ONE launch: the weight gradients of two 4x4 / stride-2 layers of one network at the same batch (the two inner
 * layers of Encoder_VAE_SVHN, models/nn/svhn.py:19-28, at the training batch, where each is a split-K GEMM of ~35 us on the
 * step's last dependent chain).  Both problems' workgroups share one grid (igemm_bf_pair_kernel); slabs and ordered finishes are
 * those of two separate mvk_conv4s2_wgrad launches (bit-identical gradients); shapes the pair launch does not cover run as two
 * launches.  NHWC U, no fused activation.  */
int mvk_conv4s2_wgrad_pair(const float* U0, const float* V0, float* dW0, int h0, int w0, int Cu0, int Cv0, const float* U1,
                           const float* V1, float* dW1, int h1, int w1, int Cu1, int Cv1, int n, float* ws, int64_t ws_floats,
                           void* stream);

/* 3x3 / stride 1 / pad 1 convolution on NHWC activations — the ResNet blocks of models/nn/mmnist.py:214-366 and
 * models/nn/cub.py:144-293.  Y[n,H,W,Cout] = act(conv(X[n,H,W,Cin]) + bias) (* src_act'(y_act_src) elementwise), with
 * Wp the kind-2 pack of mvk_pack_weights.  Backward data is the same launch on the output gradient with the Wup pack
 * (Cin / Cout swapped); colsum_acc as in mvk_conv4s2_down.  wgrad: dWref[Cout][Cin][3][3] += sum_pos X(gathered) dY. */
int mvk_conv3x3(const float* X, const float* Wp, const float* bias, float* Y, int n, int H, int W, int Cin, int Cout,
                int act, const float* y_act_src, int y_src_act, float* colsum_acc, float* ws, int64_t ws_floats,
                void* stream);
/* The same launch with a residual in the epilogue: Y = res + res_alpha * (act(conv(X) + bias) * src_act'(y_act_src)) —
 * `xs + 0.1 * conv2(...)` of the ResNet blocks (models/nn/cub.py:274-280) and `d_block_input + d_shortcut` of their backward
 * pass in one pass over the output. */
int mvk_conv3x3_res(const float* X, const float* Wp, const float* bias, float* Y, int n, int H, int W, int Cin,
                    int Cout, int act, const float* y_act_src, int y_src_act, const float* res, float res_alpha,
                    float* ws, int64_t ws_floats, void* stream);
int mvk_conv3x3_wgrad(const float* X, const float* dY, float* dWref, int n, int H, int W, int Cin, int Cout,
                      float* ws, int64_t ws_floats, void* stream);
/* Fused forms of the three launches above, taken only by the register-stationary kernels (csrc/conv3rs.hip); they fold the
 * elementwise passes of a ResnetBlock (models/nn/cub.py:274-280 `x_s + 0.1 * conv_1(actvn(conv_0(actvn(x))))`, mmnist.py:229-246)
 * into the convolutions on either side:
 *   mvk_conv3x3_fused_ok   1 when both fused launches take this problem (ask first: they return MVK_EINVAL otherwise)
 *   mvk_conv3x3_f          Y = [res + res_alpha *] (act(pre_scale * conv(x_act(X)) + bias) * src_act'(y_act_src)): x_act is the
 *                          activation of the layer that produced X, applied while X is staged (the activated tensor is never
 *                          written); res may be NULL; colsum_acc as in mvk_conv3x3 (backward-data form, exclusive with res)
 *   mvk_conv3x3_wgrad_f    dWref += dy_scale * sum_pos x_act(X)(gathered) dY and, when db != NULL, db[Cout] += dy_scale * sum_pos dY
 *                          (the bias gradient without a pass of its own; dy_scale = the 0.1 of the residual branch) */
int mvk_conv3x3_fused_ok(int n, int H, int W, int Cin, int Cout);
int mvk_conv3x3_f(const float* X, const float* Wp, const float* bias, float* Y, int n, int H, int W, int Cin, int Cout,
                  int act, const float* y_act_src, int y_src_act, const float* res, float res_alpha, float* colsum_acc,
                  int x_act, float pre_scale, float* ws, int64_t ws_floats, void* stream);
int mvk_conv3x3_wgrad_f(const float* X, const float* dY, float* dWref, float* db, int n, int H, int W, int Cin, int Cout,
                        int x_act, float dy_scale, float* ws, int64_t ws_floats, void* stream);
/* Scaled-fp16 form of mvk_conv3x3_f: every fp32 product is 3 fp16 MFMAs (hi hi' + (hi lo' + lo hi') / 2048, error-corrected
 * fp16 pairs after Ootomo & Yokota 2022; csrc/bf3.hpp states the error bound) instead of the 6 bf16 ones.  fp16 has a narrow
 * exponent, so each operand tensor is scaled by a power of two taken from an UPPER BOUND of its largest magnitude:
 *   x_amax, w_amax  device scalars >= max |X|, max |Wp| (finite inputs; a bound that is too small overflows to inf)
 *   y_amax          optional: max |Y| of this launch lands here by atomic max (*y_amax must hold 0 before the launch) and
 *                   can be the x_amax of the launch that consumes Y
 * mvk_amax computes the bound of any tensor (same protocol: *out holds 0 or a lower bound to keep); mvk_pack_weights fills
 * the `amax` slot of a descriptor.  mvk_conv3x3_scaled_ok: 1 when this form takes the problem (a superset of the shapes of
 * mvk_conv3x3_fused_ok's forward side: 128 input channels reach maps up to 62 wide). */
int mvk_amax(const float* x, int64_t n, float* out, void* stream);
/* mvk_conv3x3 with an image (Cin <= 4) on the input side that also publishes max |Y| into y_amax (atomic max, must hold 0):
 * conv_img of the ResNet encoders (models/nn/mmnist.py:262, cub.py:160) and the backward-data pass of the decoders' conv_img —
 * the stack behind it needs that bound for its scaled-fp16 launches and used to get it from a pass over Y (mvk_amax).
 * MVK_EINVAL when the direct image kernel does not take the shape (the caller then uses mvk_conv3x3 + mvk_amax). */
int mvk_conv3x3_y(const float* X, const float* Wp, const float* bias, float* Y, int n, int H, int W, int Cin, int Cout, int act,
                  const float* y_act_src, int y_src_act, float* colsum_acc, float* y_amax, float* ws, int64_t ws_floats,
                  void* stream);
int mvk_conv3x3_scaled_ok(int n, int H, int W, int Cin, int Cout);
int mvk_conv3x3_s(const float* X, const float* Wp, const float* bias, float* Y, int n, int H, int W, int Cin, int Cout,
                  int act, const float* y_act_src, int y_src_act, const float* res, float res_alpha, float* colsum_acc,
                  int x_act, float pre_scale, const float* x_amax, const float* w_amax, float* y_amax, float* ws,
                  int64_t ws_floats, void* stream);
/* mvk_conv3x3_s in its residual form with a second store (round 5): Y <- res + res_alpha * a and y_pre <- a, a = act(conv + bias).
 * The post-activation ResnetBlock of models/nn/mmnist.py:229-246 (x_s + 0.1 * lrelu(conv2(.))) keeps `a` for its backward pass:
 * one launch instead of the convolution and an elementwise pass over three tensors.  Where mvk_conv3x3_scaled_ok says so. */
int mvk_conv3x3_s2(const float* X, const float* Wp, const float* bias, float* Y, float* y_pre, int n, int H, int W, int Cin, int Cout,
                   int act, const float* res, float res_alpha, const float* x_amax, const float* w_amax, float* y_amax, float* ws,
                   int64_t ws_floats, void* stream);
/* One launch of mvk_conv3x3_s over a SLICE of the layer's input channels (round 5): X [n][H][W][x_channels], the slice = Cin
 * channels from x_off; Wp = the layer's whole pack [9 x_channels][Cout].  res_pre != 0: Y = act(conv + bias + res), `res` = the
 * partial sum the other slice's launch left (act = none, no bias there).  The register-stationary kernels hold the weights of at
 * most 128 input channels: a 256-channel layer (models/nn/mmnist.py:345-352, cub.py:233-240) is two such launches. */
int mvk_conv3x3_s_part(const float* X, int x_channels, int x_off, const float* Wp, const float* bias, float* Y, int n, int H, int W,
                       int Cin, int Cout, int act, const float* y_act_src, int y_src_act, const float* res, int res_pre, int x_act,
                       float pre_scale, const float* x_amax, const float* w_amax, float* y_amax, void* stream);
/* mvk_conv3x3_wgrad_f on scaled fp16 pairs (one accumulator per tap tile: the 2^11 between main and cross terms sits in a
 * third piece of dY, csrc/conv3rs.hip); x_amax / dy_amax: device scalars >= max |X| / max |dY|. */
int mvk_conv3x3_wgrad_scaled_ok(int n, int H, int W, int Cin, int Cout);
int mvk_conv3x3_wgrad_s(const float* X, const float* dY, float* dWref, float* db, int n, int H, int W, int Cin, int Cout,
                        int x_act, float dy_scale, const float* x_amax, const float* dy_amax, float* ws, int64_t ws_floats,
                        void* stream);
/* nn.AvgPool2d(3, stride=2, padding=1) (count_include_pad: every window divides by 9) and nn.Upsample(scale_factor=2)
 * (nearest) on NHWC tensors, forward and backward; out = act(a*x + b*y) (x or y may be NULL). */
int mvk_avgpool3s2_fwd(const float* x, float* y, int n, int H, int W, int C, void* stream);
int mvk_avgpool3s2_bwd(const float* dy, float* dx, int n, int H, int W, int C, void* stream);
int mvk_upsample2_fwd(const float* x, float* y, int n, int H, int W, int C, void* stream);
int mvk_upsample2_bwd(const float* dy, float* dx, int n, int H, int W, int C, void* stream);
int mvk_axpby(const float* x, float a, const float* y, float b, int64_t n, int act, float* out, void* stream);
/* Measurement aid of bench.py (not on the training path): one launch of a register-only loop of v_mfma_f32_32x32x16_bf16 — 256
 * workgroups, one wave per SIMD, 4 accumulator chains, 64 * iters MFMAs per chain — with hashed operand values (random_operands = 1)
 * constants (0), or hashed fp16 values on v_mfma_f32_32x32x16_f16 (2: the scaled-fp16 kernels' instruction).  Timed by the caller, it gives the matrix-pipe rate the chip SUSTAINS for a launch of that length; with
 * realistic operands that is 1.4-1.9 PFLOP/s, not the 2.5 of the data sheet (power-limited clock).  out: >= 65536 floats. */
int mvk_probe_mfma_bf16(float* out, int iters, int random_operands, void* stream);
/* Measurement aid of bench.py: a float4 streaming copy of n floats (16-byte aligned, n % 4 == 0; nontemporal, four loads in
 * flight per lane) — the measured HBM denominator beside the 8 TB/s of the data sheet. */
int mvk_probe_stream_copy(float* dst, const float* src, int64_t n, void* stream);
/* y[b][c][r] = act(x[b][r][c]) * dact'(msrc[b][r][c]) (msrc may be NULL; the derivative is taken through the OUTPUT of dact,
 * as everywhere here): the NCHW flatten in front of `fc_mu / fc_logvar` of the ResNet encoders fused with their last activation
 * (models/nn/cub.py:190-195, mmnist.py:300-306), the un-flatten behind `fc` of the decoders (cub.py:232-240), and their
 * backward passes — x, msrc: [batch][rows][cols], y: [batch][cols][rows]; batch <= 65535. */
int mvk_transpose_act(const float* x, float* y, int batch, int rows, int cols, int act, const float* msrc, int dact,
                      void* stream);

/* Noise drawn with the generator state in device memory (the reparameterisation noise of models/base/base_utils.py:129-160
 * `rsample_from_gaussian` and of every `rsample` on the path; the reference calls torch's generator).  state: 3 x uint64
 * {seed, offset, arrival ticket} owned by the caller, advanced by the launch itself — a captured training step draws fresh
 * noise at every replay with no host-issued launch in front of it.  Philox4x32-10, counter = offset + index / 4; uniform = 0:
 * N(0, 1) by Box-Muller; uniform = 1: U[lo, hi).  The values depend only on (seed, offset, index). */
int mvk_device_rng(float* out, int64_t n, uint64_t* state, int uniform, float lo, float hi, void* stream);
/* Direct kernel for the 3-channel image-producing layer: U[n,Cu,2h,2w] (NCHW) = act(convT(V) + b), Cu <= 4,
 * reading the reference weight tensor directly (models/nn/svhn.py:58-60). */
int mvk_conv4s2_up_nchw_small(const float* V, const float* Wref, const float* bias, float* U, int n, int h,
                              int w, int Cu, int Cv, int act, void* stream);

/* MFMA kernels for the same layer when h*w is a multiple of 64 (<= 256), Cu <= 4 and Cv in {16,32,64}
 * (mvk_conv4s2_small_up_supported != 0): one workgroup per image computes the column matrix
 * V[pos,:] . W[:,(cu,kh,kw)] on the matrix cores and gathers the output pixels from LDS (no wasted MFMA columns).
 * The backward entry point fuses backward-data (dV = conv(dUpre) * v_act'(V)), backward-weight (dWref +=) and
 * the bias gradient (db +=, nullable) with dUpre = dU * u_act'(Uout) applied while loading; db_v (nullable) +=
 * the per-channel sums of dV, i.e. the bias gradient of the layer that produced V; ws is split-K style scratch
 * (>= 512 * (16*Cu*Cv + Cu + Cv) floats for full parallelism). */
/* Forward of the image-CONSUMING layer (svhn.py:13-15, the encoder's first Conv2d(Cu <= 4, Cv, 4, 2, 1) on the NCHW
 * network input): V[n,h,w,Cv] = act(conv(U[n,Cu,2h,2w]) + bias), Wdown = the packed [16 Cu][Cv] weight of
 * mvk_pack_conv4s2_weight.  Same support set as mvk_conv4s2_small_up_supported; mvk_conv4s2_down routes here. */
int mvk_conv4s2_small_down_fwd(const float* U, const float* Wdown, const float* bias, float* V, int n, int h, int w,
                               int Cu, int Cv, int act, void* stream);
/* ... from the reference weight layout Wref[Cv][Cu][4][4]: the encoder's first kernel does not wait for mvk_pack_weights */
int mvk_conv4s2_small_down_fwd_wref(const float* U, const float* Wref, const float* bias, float* V, int n, int h, int w,
                                    int Cu, int Cv, int act, void* stream);
int mvk_conv4s2_small_up_supported(int h, int w, int Cu, int Cv);
int mvk_conv4s2_small_up_fwd(const float* V, const float* Wref, const float* bias, float* U, int n, int h, int w,
                             int Cu, int Cv, int act, void* stream);
int mvk_conv4s2_small_up_bwd(const float* dU, const float* Uout, int u_act, const float* V, int v_act,
                             const float* Wref, float* dV, float* dWref, float* db, float* db_v, float* ws,
                             int64_t ws_floats, int n, int h, int w, int Cu, int Cv, void* stream);
/* The fused decoder tail (round 3, opt-in for the in-package Decoder_VAE_SVHN, models/nn/svhn.py:52-70): the image-producing
 * layer scores its output against the data by a Normal(scale) likelihood (models/base/base_utils.py:62-87, mopoe_model.py:192-199)
 * in its own epilogue — rows[n] = -log p(x | image) summed over the image (what mvk_recon_nll_fwd produces from the image),
 * dpre[n,Cu,2h,2w] = d rows / d pre-activation stored where the image would be; X: [xrows][Cu * 4 h w] targets, image i is scored
 * against X[i % xrows] (the K samples of a data point share it).  mvk_conv4s2_small_up_bwd_pre is the layer's backward for that
 * buffer: dpre * rowscale[n] (d loss / d rows; NULL = 1) replaces dU * u_act'(Uout), the image is not read.  3 of the 5 passes over
 * the [K B, 3, 32, 32] tensor disappear.  Only where mvk_conv4s2_small_up_nll_supported returns 1 (16x16x32 -> 3x32x32). */
int mvk_conv4s2_small_up_nll_supported(int h, int w, int Cu, int Cv);
int mvk_conv4s2_small_up_fwd_nll(const float* V, const float* Wref, const float* bias, const float* X, int xrows, float scale,
                                 float* dpre, float* rows, int n, int h, int w, int Cu, int Cv, int act, void* stream);
int mvk_conv4s2_small_up_bwd_pre(const float* dpre, const float* rowscale, const float* V, int v_act, const float* Wref,
                                 float* dV, float* dWref, float* db, float* db_v, float* ws, int64_t ws_floats, int n, int h,
                                 int w, int Cu, int Cv, void* stream);
/* ..._fwd_nll with the stored gradient pre-multiplied by grad_weight (the weight the rows enter the loss with: a backward pass
 * whose upstream row gradient is exactly that constant passes rowscale = NULL and reads no row gradient) */
int mvk_conv4s2_small_up_fwd_nll_w(const float* V, const float* Wref, const float* bias, const float* X, int xrows, float scale,
                                   float grad_weight, float* dpre, float* rows, int n, int h, int w, int Cu, int Cv, int act,
                                   void* stream);
/* The image-producing layer (plain and fused-tail form) on scaled fp16 pairs: v_amax = device scalar bounding max |V|, published by
 * the launch that produced V (amax protocol above); the weight scale is derived in the kernel.  3 MFMAs per product, the weights as
 * the A operand (one 16-byte column-matrix write per tile), the fast sigmoid.  Same reference layer (models/nn/svhn.py:59-60
 * ConvTranspose2d(32, 3, 4, 2, 1) + Sigmoid) and likelihood (models/base/base_utils.py:62-87) as the entry points above; only
 * where mvk_conv4s2_small_up_nll_supported returns 1. */
int mvk_conv4s2_small_up_fwd_s(const float* V, const float* Wref, const float* bias, float* U, int n, int h, int w, int Cu, int Cv,
                               int act, const float* v_amax, void* stream);
int mvk_conv4s2_small_up_fwd_nll_s(const float* V, const float* Wref, const float* bias, const float* X, int xrows, float scale,
                                   float grad_weight, float* dpre, float* rows, int n, int h, int w, int Cu, int Cv, int act,
                                   const float* v_amax, void* stream);
/* The image layer's backward on scaled fp16 pairs (round 5; same reference lines as mvk_conv4s2_small_up_bwd_pre:
 * models/nn/svhn.py:58-60 ConvTranspose2d(32, 3, 4, 2, 1) + Sigmoid under models/base/base_utils.py:62-87's log-probability).
 * _fwd_nll_sy = _fwd_nll_s that also publishes an upper bound of max |dpre| into *dpre_amax (atomic max; must hold 0 before the
 * launch; from the largest row sum: at most sqrt(Cu 4 h w) times the true maximum); _bwd_pre_s = _bwd_pre_y with the gradient
 * image scaled under dpre_amax x max |rowscale| and V under v_amax (dv_amax may be NULL). */
int mvk_conv4s2_small_up_fwd_nll_sy(const float* V, const float* Wref, const float* bias, const float* X, int xrows, float scale,
                                    float grad_weight, float* dpre, float* rows, int n, int h, int w, int Cu, int Cv, int act,
                                    const float* v_amax, float* dpre_amax, void* stream);
int mvk_conv4s2_small_up_bwd_pre_s(const float* dpre, const float* rowscale, const float* V, int v_act, const float* Wref,
                                   float* dV, float* dWref, float* db, float* db_v, float* ws, int64_t ws_floats, int n, int h,
                                   int w, int Cu, int Cv, const float* dpre_amax, const float* v_amax, float* dv_amax,
                                   void* stream);

/* 1x1-spatial layers:
 *   unflatten  Y[n,(tap,co)] = act(z[n,Cin] Wp + b[co]), Wp[ci][tap*Cout+co] = Wref[ci][co][tap]
 *              (ConvTranspose2d(L,128,4,1,0), svhn.py:51) — forward / backward-data via mvk_gemm on Wp;
 *   flatten    Y[n,cv] = H[n,(tap,cu)] Wdown + b  (Conv2d(128,L,4,2,0) heads, svhn.py:29-30) — via mvk_gemm
 *              on the `down` packing.
 * The weight gradients scatter straight back into the reference layouts: */
int mvk_pack_unflatten_weight(const float* Wref, int Cin, int Cout, float* Wp, void* stream);
int mvk_unflatten_wgrad(const float* Z, const float* dY, float* dWref, int n, int Cin, int Cout, float* ws,
                        int64_t ws_floats, void* stream);
int mvk_flatten_wgrad(const float* H, const float* dY, float* dWref, int n, int Cu, int Cv, float* ws,
                      int64_t ws_floats, void* stream);

/* NCHW <-> NHWC copies at the plugin boundary. */
int mvk_nchw_to_nhwc(const float* src, float* dst, int n, int c, int h, int w, void* stream);
int mvk_nhwc_to_nchw(const float* src, float* dst, int n, int c, int h, int w, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The reference's public helper functions (multivae/models/base/base_utils.py) for user code that imports them
 * ------------------------------------------------------------------------------------------------ */
/* poe(mus, logvars, eps) (base_utils.py:122-130; stable = 0) and stable_poe (:133-147; stable = 1, no eps, an expert with
 * logvar = +inf has weight exactly 0, a single expert is returned as is): mus, lvs [E, n] -> mu, lv [n]; backward from
 * the gradients of both outputs (either may be NULL) to dmus, dlvs [E, n]. */
int mvk_poe_fwd(const float* mus, const float* lvs, int E, int64_t n, float eps, int stable, float* mu, float* lv,
                void* stream);
int mvk_poe_bwd(const float* mus, const float* lvs, int E, int64_t n, float eps, int stable, const float* gmu,
                const float* glv, float* dmus, float* dlvs, void* stream);
/* kl_divergence(mean, log_var, prior_mean, prior_log_var).sum(-1) (base_utils.py:90-119): kl [rows]; every operand holds
 * n_* elements and is indexed modulo that count (trailing-dimension broadcasting: a [1, L] prior under [rows, L]
 * posteriors).  Backward writes FULL-shape [rows, L] partial derivatives (NULL outputs are skipped); the caller column-
 * sums those of broadcast operands (mvk_colsum_acc). */
int mvk_kl_gauss_fwd(const float* mean, int64_t n_mean, const float* lv, int64_t n_lv, const float* pmean, int64_t n_pmean,
                     const float* plv, int64_t n_plv, int64_t rows, int L, float* kl, void* stream);
int mvk_kl_gauss_bwd(const float* mean, int64_t n_mean, const float* lv, int64_t n_lv, const float* pmean, int64_t n_pmean,
                     const float* plv, int64_t n_plv, int64_t rows, int L, const float* g, float* dmean, float* dlv,
                     float* dpmean, float* dplv, void* stream);
/* set_decoder_dist(dist, params)(recon, target) (base_utils.py:62-87) and cross_entropy (:28-57): ELEMENT-WISE log-
 * probabilities lp [n] of recon [n] against target [n_target] (indexed modulo n_target: a [B, D] target under [K, B, D]
 * reconstructions); dist = MVK_DIST_*; scale for normal / laplace; categorical: x * log_softmax(recon + eps) over the last
 * dimension C.  Backward: drecon = g * d lp / d recon. */
int mvk_logprob_fwd(const float* recon, const float* target, int64_t n, int64_t n_target, int dist, float scale, int C,
                    float eps, float* lp, void* stream);
int mvk_logprob_bwd(const float* recon, const float* target, int64_t n, int64_t n_target, int dist, float scale, int C,
                    float eps, const float* g, float* drecon, void* stream);

/* Up to MVK_COPY_MAX device-to-device copies in one launch (dst, src 16-byte aligned, bytes a multiple of 16): the tensors of one
 * collated batch (trainers/base/base_trainer.py:682-700: `inputs = set_inputs_to_device(inputs, device)`) into the input buffers a
 * replayed step reads.  Host array of descriptors, copied into the launch. */
#define MVK_COPY_MAX 8
typedef struct {
  void* dst;
  const void* src;
  int64_t bytes;
} mvk_copy_desc;
int mvk_copy_batch(const mvk_copy_desc* descs, int n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer
 * ------------------------------------------------------------------------------------------------ */
/* torch.optim.Adam (amsgrad=False) on flat buffers; step is 1-based; grad_scale multiplies g first
 * (1/world_size for DDP averaging).  trainers/base/base_trainer_config.py:58,62; base_trainer.py:350-361. */
int mvk_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                  double eps, double weight_decay, int step, double grad_scale, void* stream);
/* torch.optim.Adam(amsgrad=True) (the reference's MMVAE+ PolyMNIST setting, examples/mmvae_plus/mmnist.py:61-62): vmax is
 * the running maximum of exp_avg_sq (`max_exp_avg_sq`), updated in place and used in the denominator; vmax == NULL is
 * mvk_adam_step.  lr is a host scalar per call, so learning-rate schedulers (base_trainer_config.py:60) cost nothing. */
int mvk_adam_step_amsgrad(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, double lr, double beta1,
                          double beta2, double eps, double weight_decay, int step, double grad_scale, void* stream);
/* The same update (vmax = NULL: plain Adam) that also CLEARS the gradient buffer as it consumes it when zero_grad != 0: the next
 * step's `optimizer.zero_grad()` (base_trainer.py:350) needs no pass over the buffer and no launch. */
int mvk_adam_step_fused(float* p, float* g, float* m, float* v, float* vmax, int64_t n, double lr, double beta1,
                        double beta2, double eps, double weight_decay, int step, double grad_scale, int zero_grad,
                        void* stream);
/* The optimizer inside a replayed hipGraph: the scalars of the update live in DEVICE memory.
 *   state    8 doubles owned by the caller: lr, beta1, beta2, eps, weight_decay, grad_scale, step (updates applied so far), 0
 *   scalars  8 floats (16-byte aligned) written by mvk_adam_prepare and read by mvk_adam_step_dev
 * mvk_adam_prepare (one thread) advances state[6] by one and derives the bias-corrected step size etc. in double, exactly the
 * arithmetic mvk_adam_step_fused does on the host; mvk_adam_step_dev is mvk_adam_step_fused with those scalars (buffers
 * 16-byte aligned, n % 4 == 0: the flat buffers of trainers.FlatParams).  Captured once, replayed every step: a learning-rate
 * scheduler (base_trainer_config.py:60) writes state[0] between replays; the pair is equivalent to one
 * optimizer.step() of base_trainer.py:350-361. */
int mvk_adam_prepare(double* state, float* scalars, void* stream);
int mvk_adam_step_dev(float* p, float* g, float* m, float* v, float* vmax, int64_t n, const float* scalars, int zero_grad,
                      void* stream);
/* The rotated step (trainers/graph.py: the decoder's late weight gradients of step N are produced at the head of step N + 1, beside
 * the encoders' forward pass).  optimizer.step() of base_trainer.py:350-361 is then TWO launches over disjoint ranges of the flat
 * buffers with the SAME scalars: mvk_adam_step_pub — mvk_adam_step_fused over the range whose gradients are final when the
 * step ends, which also leaves the update's scalars (step size, bias corrections, ...: 8 floats) in `publish` — and, inside the
 * next replay and behind the late gradients, mvk_adam_step_dev over the other range reading `publish`.  Every parameter is
 * updated exactly once per step and before it is next read.  n == 0 publishes only.  mvk_adam_identity writes the scalars
 * under which mvk_adam_step_dev changes nothing (the state before the first step and after a drain). */
int mvk_adam_step_pub(float* p, float* g, float* m, float* v, float* vmax, int64_t n, double lr, double beta1, double beta2,
                      double eps, double weight_decay, int step, double grad_scale, int zero_grad, float* publish, void* stream);
int mvk_adam_identity(float* scalars, void* stream);

/* The encoder heads in ONE launch: Y_h[m][n] = sum_k X[m][k] W_h(k, n) + b_h[n] for h = 0 (and 1 when W1 != NULL),
 * n < N <= 32, W_h(k, n) = W_h[k * w_sk + n * w_sn] (a torch Linear weight [N][K]: w_sk = 1, w_sn = K; the packed
 * Conv2d(C, L, 4, 2, 0) head [(kh,kw,c)][L]: w_sk = L, w_sn = 1).  Replaces `self.embedding(h)`, `self.log_var(h)` of
 * Encoder_VAE_MLP (models/nn/default_architectures.py:40-56) and `self.c1(e)`, `self.c2(e)` of Encoder_VAE_SVHN
 * (models/nn/svhn.py:29-34).  K % 4 == 0, X 16-byte aligned; exact fp32 (v_mfma_f32_16x16x4_f32), fixed summation
 * order.  b_h may be NULL. */
int mvk_heads_fwd(const float* X, const float* W0, const float* b0, float* Y0, const float* W1, const float* b1,
                  float* Y1, int M, int N, int K, int64_t w_sk, int64_t w_sn, void* stream);

/* Their backward in ONE launch (6 launches on the GEMM engine before: 2 weight gradients, 2 bias column sums, 2
 * backward-data products):  dX[m][k] = (sum_h sum_n dY_h[m][n] W_h(k, n)) * x_act'(X[m][k])  (x_act from the OUTPUT value
 * X; dX may be NULL),  dW_h += dY_h^T X  stored at dW_h[n * K + k], or at dW_h[n * K + (k % flat_c) * (K / flat_c) +
 * k / flat_c] when flat_c > 0 (k = (tap, c) of an NHWC map -> the [L][C][4][4] layout of the Conv2d heads),
 * db_h += column sums of dY_h (may be NULL),  dbprev += column sums of dX (may be NULL: the bias gradient of the layer that
 * produced X; K entries, or flat_c channel sums when flat_c > 0).  Per 128-row group a workgroup writes partial sums to slabs (the deferred arena of mvk_defer_begin when the
 * target lies in the flat gradient buffer, else `ws`: (2 N K + 2 N + K) * ceil(M / 128) floats at most); the groups are
 * added in order: deterministic.  N <= 32, K % 16 == 0, X / dX 16-byte aligned. */
int mvk_heads_bwd(const float* X, int x_act, const float* dY0, const float* dY1, const float* W0, const float* W1,
                  int64_t w_sk, int64_t w_sn, int flat_c, float* dX, float* dW0, float* dW1, float* db0, float* db1,
                  float* dbprev, int M, int N, int K, float* ws, int64_t ws_floats, void* stream);

/* Deferred leaf reductions.  Parameter gradients are leaves of the backward pass (reference: autograd accumulates them
 * into `.grad`, nothing reads them before `optimizer.step()`, trainers/base/base_trainer.py:405-420), so the ordered
 * finishes that complete them - split-K slabs of the weight-gradient GEMMs, per-workgroup column sums of the bias
 * gradients, the slabs of the convolution weight-gradient kernels - need not run one launch each behind their producer.
 * Between mvk_defer_begin and mvk_defer_end every entry point whose result ACCUMULATES into [grad, grad + grad_floats)
 * (the flat gradient buffer of trainers.FlatParams) writes its partial results into a private region of `arena` instead
 * of the caller's shared scratch and queues the finish; mvk_defer_flush(stream) orders `stream` behind every stream such a
 * producer ran on and runs all queued finishes in ONE launch per 24 (same fixed summation order per element as the
 * immediate kernels: results are bit-reproducible run to run).  A producer that does not fit (arena full, a second
 * accumulation into the same parameter while one is queued, a target outside the gradient buffer) finishes immediately
 * as without deferral.  The caller must flush before anything reads the gradient buffer (all-reduce, optimizer step) and
 * must keep `arena` alive and unused by others until then.  mvk_defer_flush may be called on ANY stream in the middle of
 * the backward pass (e.g. a side stream, once the decoders are done: the finishes then run beside the latency-bound
 * encoder backward); the arena regions it reads stay reserved and the next flush waits for it.  mvk_defer_end(stream) =
 * final flush on `stream` + off: after it (in stream order) the gradient buffer is complete.  mvk_defer_begin returns
 * MVK_EINVAL while finishes are pending; mvk_defer_pending = number of queued finishes. */
int mvk_defer_begin(float* arena, int64_t arena_floats, const float* grad, int64_t grad_floats);
int mvk_defer_flush(void* stream);
int mvk_defer_end(void* stream);
/* Arena floats asked for since the last mvk_defer_begin, granted or declined for lack of room: the host sizes the arena
 * from it (multivae_amd/kernels.py grows its arena between steps instead of reserving a fixed 512 MB). */
int64_t mvk_defer_wanted(void);
int mvk_defer_pending(void);

/* The collective of the data-parallel step (`allreduce_avg`; reference: the DDP wrapper of
 * trainers/base/base_trainer.py:92-117 averages the gradients over the ranks, :350-361 then steps the optimizer).  One process
 * per GPU; RCCL over xGMI, resolved at run time (dlopen: single-GPU processes never load it).
 *   mvk_comm_unique_id   rank 0 makes a rendezvous id of mvk_comm_id_bytes() bytes and hands it to every rank (any channel)
 *   mvk_comm_init        collective: every rank, same id, called with its own device current
 *   mvk_allreduce_avg    buf[i] <- mean over ranks, in place, enqueued on `stream` (no host synchronisation); nseg > 1 issues
 *                        nseg equal segments as one RCCL group
 *   mvk_comm_destroy */
int mvk_comm_id_bytes(void);
int mvk_comm_unique_id(void* id);
int mvk_comm_init(void** comm, int world, int rank, const void* id);
int mvk_comm_destroy(void* comm);
int mvk_allreduce_avg(float* buf, int64_t n, int nseg, void* comm, void* stream);
/* 1 when RCCL resolves in this process (no collective, no device work): ranks agree on the RCCL path BEFORE mvk_comm_init */
int mvk_comm_available(void);
/* world size and rank as the COMMUNICATOR reports them (ncclCommCount / ncclCommUserRank); either output may be NULL */
int mvk_comm_size(void* comm, int* world, int* rank);
/* The mean over the ranks of `nrange` disjoint ranges of one buffer (off / cnt: HOST arrays, in floats) as ONE RCCL group on
 * `stream`; a range longer than seg_floats (> 0) is cut into segments.  The overlapped data-parallel step issues the ranges
 * whose gradients are final early on a communication stream, behind an event of the replayed graph (mvk_event_record), and
 * the remaining ranges behind the end of the backward pass — DDP's bucket overlap (base_trainer.py:116-117,359) with two
 * buckets whose boundary is where the captured step says it is. */
int mvk_allreduce_avg_ranges(float* buf, const int64_t* off, const int64_t* cnt, int nrange, int64_t seg_floats, void* comm,
                             void* stream);
/* Events that cross the boundary of a replayed hipGraph: mvk_event_record(ev, external = 1, stream) on a CAPTURING stream adds
 * an event-record NODE (hipEventRecordWithFlags + hipEventRecordExternal), so a stream outside the graph can wait for a point
 * inside every replay (mvk_stream_wait_event, issued after the replay was launched).  external = 0 / a non-capturing stream:
 * a plain hipEventRecord. */
int mvk_event_create(void** ev);
int mvk_event_destroy(void* ev);
int mvk_event_record(void* ev, int external, void* stream);
int mvk_stream_wait_event(void* stream, void* ev);

/* Dense layers on pre-split fp16 pair planes (csrc/dense16.hip): the MLP decoder of the MnistSvhn models at the decoder batch
 * (reference: models/nn/default_architectures.py:225-258 Decoder_AE_MLP; likelihood models/base/base_utils.py:62-87).
 * A "planes" tensor [R][C] is two fp16 arrays hi, lo [R][C] with x s = hi + lo / 2048 (csrc/bf3.hpp); s is a power of two
 * derived from a device scalar `bound` >= max |x| (activations, gradients: one per tensor) or kept per plane ROW as inv[r] = 1 / s_r
 * (weights).  Every operand is split ONCE by its producer; the GEMM kernels move 16-byte pieces and issue three fp16 MFMAs per
 * fp32 product with no conversion arithmetic in the loop.  C % 8 == 0, 16-byte aligned planes.
 *   mvk_dense16_pack      W [N][K] (nn.Linear) -> planes [N][K] + nk_inv[N] (forward operand) and planes [K][N] + kn_inv[K]
 *                         (backward-data operand), one launch per step
 *   mvk_dense16_first     H planes [M][N] = act(Z [M][K] W^T + bias), K <= 32, N <= 1024 with 256 % (N / 4) == 0; *bound receives
 *                         the a-priori bound max_n ||W[n]||_1 z_amax + max |bias| the planes are scaled by (z_amax: device scalar
 *                         >= max |Z|, e.g. from mvk_amax)
 *   mvk_dense16_fwd_nll   the output layer + its likelihood: r = sigmoid(H W^T + bias) scored against X[m % xrows] by Normal(scale);
 *                         rows_part [mvk_dense16_fwd_nll_rows(N)][M] receives partial NLL row sums (their sum over the first axis
 *                         is -log p(x | r) of the row), G planes [M][N] = grad_weight * d NLL / d pre-activation under the bound
 *                         grad_weight (1 + x_amax) / (4 scale^2) written to *g_bound, colsum_part [mvk_dense16_colsum_rows(M)][N]
 *                         (optional) the column sums of G per row tile (the bias gradient's partials).  Neither the reconstruction
 *                         nor its gradient is written as an fp32 tensor.
 *   mvk_dense16_bwd_data  dA [M][N] fp32 = (G [M][K] planes x W planes [N][K] (the [K][N]-orientation pack of the layer's weight))
 *                         * ReLU'(mask) with mask_hi = the hi plane of the activation dA lands in (or NULL); db (optional) +=
 *                         column sums of dA (deferred finish when db lies in the registered gradient buffer, else ws)
 *   mvk_dense16_wgrad     dW [N][K] += G^T H over the M rows (both planes tensors; ordered finish of per-slice slabs: deferred,
 *                         else ws of >= slices * N * K floats), db [N] += the rows of colsum_part in order
 *   mvk_dense16_unsplit   planes -> fp32, optionally times rowf[m] * rowf_scale — rowf_tile > 0: rowf[(n / rowf_tile) M + m], one factor
 *                         per column tile and row, the layout of rows_part — (tests; the general backward when the upstream
 *                         row gradient is not the weight folded into G) */
int mvk_dense16_ok(int M, int N, int K);
int mvk_dense16_pack(const float* W, int N, int K, void* nk_hi, void* nk_lo, float* nk_inv, void* kn_hi, void* kn_lo,
                     float* kn_inv, void* stream);
int mvk_dense16_first(const float* Z, const float* W, const float* bias, const float* z_amax, void* hi, void* lo, float* bound,
                      int M, int N, int K, int act, void* stream);
int mvk_dense16_fwd_nll_rows(int N);
int mvk_dense16_colsum_rows(int M);
int mvk_dense16_fwd_nll(const void* h_hi, const void* h_lo, const float* h_bound, const void* w_hi, const void* w_lo,
                        const float* w_inv, const float* bias, const float* X, int xrows, const float* x_amax, float scale,
                        float grad_weight, void* g_hi, void* g_lo, float* g_bound, float* rows_part, float* colsum_part, int M,
                        int N, int K, void* stream);
int mvk_dense16_bwd_data(const void* g_hi, const void* g_lo, const float* g_bound, const void* wt_hi, const void* wt_lo,
                         const float* wt_inv, const void* mask_hi, float* dA, float* db, float* ws, int64_t ws_floats, int M, int N,
                         int K, void* stream);
int mvk_dense16_wgrad(const void* g_hi, const void* g_lo, const float* g_bound, const void* h_hi, const void* h_lo,
                      const float* h_bound, const float* colsum_part, int cs_rows, float* dW, float* db, float* ws,
                      int64_t ws_floats, int M, int N, int K, void* stream);
int mvk_dense16_unsplit(const void* hi, const void* lo, const float* bound, const float* rowf, float rowf_scale, int rowf_tile,
                        int M, int N, float* out, void* stream);
void mvk_dense16_debug(int flags); /* ablation switches of tools/dense16_probe.py (0 = the shipped kernels) */
void mvk_dense16_debug_stamps(float* four_floats); /* flag 16: where the forward kernel's cycle stamps go (NULL: nowhere) */

/* Device-timestamp profiler (bench.py's roofline objects).  device_slots: nslots records of MVK_PROF_SLOT_U64 = 520
 * uint64 each: [0] sum of durations (clock ticks, first workgroup in -> last workgroup out), [1] launches accumulated,
 * [2] sum of (first workgroup in -> start of the one-wave fold kernel queued behind the launch: the launch has drained
 * its stores by then; this bracket contains what a kernel trace reports), [8 + 8 e] start stamps, [264 + 8 e] end stamps
 * (e < 32); the caller initialises the start stamps to ~0 and everything else to 0.  Every instrumented launch
 * (kind 1: fused reconstruction NLL forward, 2: imgconv up, 3: imgconv down, 4: imgconv weight gradient, 5 / 6: image-
 * layer forward / backward) takes the next record: its workgroups stamp the constant-rate clock on entry (min) and exit
 * (max), and a one-wave kernel behind it adds max(end) - min(start) to the sum and re-arms the record, so a launch
 * captured into a hipGraph accumulates one duration per replay.  host_kinds / host_work (nslots entries, may be NULL)
 * receive the kind and the algorithmic work of the launch (bytes for the HBM-bound kinds 1, 5, 6, FLOP for the
 * others).  device_slots == NULL switches the profiler off (later launches are not stamped; captured ones keep their
 * record). */
int mvk_prof_enable(void* device_slots, int nslots, int32_t* host_kinds, double* host_work);
int mvk_prof_count(void);     /* records taken since mvk_prof_enable */
/* n one-wave kernels back to back, each storing the clock at its first and last instruction into device_ticks[2 i],
 * device_ticks[2 i + 1] (uint64): first[i + 1] - last[i] is the dependent-kernel boundary that the retire bracket of a
 * record contains once. */
int mvk_prof_calibrate(void* device_ticks, int n, void* stream);
int mvk_prof_clock_khz(void); /* rate of the stamped clock */

/* Measurement hooks for experiment builds of the library (-DMVK_PHASES: per-phase cycle counters of the GEMM main
 * loop accumulated into an 8-entry device buffer; -DMVK_EXPER: ablation switches).  Inert in the shipped build. */
void mvk_debug_set_phase_buffer(unsigned long long* device_counters);
void mvk_debug_set_flags(int flags);

#ifdef __cplusplus
}
#endif
#endif /* MVK_H */
