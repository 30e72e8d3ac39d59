// Small HBM-bound helper kernels: weight packing, layout changes at the plugin boundary, activation
// backward, the 3-channel image-producing transposed convolution, flat Adam, scalar assembly.
#include "bf3.hpp"

namespace {

// ---- weight packing -------------------------------------------------------------------------------------
// Wref[cv][cu][kh][kw]  ->  Wdown[(kh*4+kw)*Cu + cu][cv]            (ld = ldd, column offset coff)
//                       ->  Wup[ph*2+pw][((a*2+b)*Cv + cv)][cu]     with kh = (1-ph)+2a, kw = (1-pw)+2b
__global__ void pack_conv4s2_kernel(const float* __restrict__ Wref, int Cv, int Cu, float* __restrict__ Wdown,
                                    int ldd, int coff, float* __restrict__ Wup) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  int total = Cv * Cu * 16;
  if (idx >= total) return;
  int tap = idx & 15;
  int cu = (idx >> 4) % Cu;
  int cv = (idx >> 4) / Cu;
  float v = Wref[idx];
  int kh = tap >> 2, kw = tap & 3;
  if (Wdown) Wdown[(long long)(tap * Cu + cu) * ldd + coff + cv] = v;
  if (Wup) {
    int ph = 1 - (kh & 1), a = kh >> 1;
    int pw = 1 - (kw & 1), b = kw >> 1;
    Wup[((long long)(ph * 2 + pw) * 4 * Cv + (a * 2 + b) * Cv + cv) * Cu + cu] = v;
  }
}

// All weights of one network in ONE launch: blockIdx.y selects the descriptor (kind 0: 4x4/stride-2 pack as
// above, kind 1: 1x1-spatial "unflatten" pack Wref[ci][co][tap] -> Wup[ci][tap*Cu + co] with Cv = Cin, Cu = Cout).
struct PackJobs {
  mvk_pack_desc j[MVK_PACK_MAX];
};
// max |Wref| of a descriptor (the weight's scale for the scaled-fp16 kernels, bf3.hpp): the first 32 workgroups of the
// descriptor read the (L2-resident) weight once more, eight independent loads in flight per thread, and publish — 32 atomics per
// weight.  (One atomic per workgroup of the launch: 2304 on 9 addresses made the 13-us launch 32 us; 8 workgroups with one
// load in flight at a time: 64 dependent L2 latencies, 32 us again.)
__device__ __forceinline__ void pack_amax(const mvk_pack_desc& d, int total, float* red) {
  constexpr int NB = 32;
  if (!d.amax || blockIdx.x >= NB) return;  // uniform per workgroup
  float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int stride = NB * 256;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += 8 * stride) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = i + j * stride;
      m[j] = fmaxf(m[j], k < total ? fabsf(d.Wref[k]) : 0.f);
    }
  }
  const float mm = fmaxf(fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3])), fmaxf(fmaxf(m[4], m[5]), fmaxf(m[6], m[7])));
  mvk::amax_publish(mm, d.amax, red);
}

__global__ __launch_bounds__(256) void pack_multi_kernel(const PackJobs jobs) {
  __shared__ float amax_red[4];
  const mvk_pack_desc& d = jobs.j[blockIdx.y];
  if (d.kind == 2) {  // 3x3: Wref[cv][cu][3][3] -> Wdown[(tap*Cu + cu)][cv] (forward), Wup[((8-tap)*Cv + cv)][cu] (bwd data)
    const int total9 = d.Cv * d.Cu * 9;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total9; idx += gridDim.x * 256) {
      const int tap = idx % 9;
      const int cu = (idx / 9) % d.Cu;
      const int cv = (idx / 9) / d.Cu;
      const float v = d.Wref[idx];
      if (d.Wdown) d.Wdown[(long long)(tap * d.Cu + cu) * d.ld_down + d.col_off + cv] = v;
      if (d.Wup) d.Wup[(long long)((8 - tap) * d.Cv + cv) * d.Cu + cu] = v;
    }
    pack_amax(d, total9, amax_red);
    return;
  }
  const int total = d.Cv * d.Cu * 16;
#ifndef MVK_PACK_TILED
#define MVK_PACK_TILED 1
#endif
  // Tiled form (round 4): the element-per-thread loop below reads coalesced and issues up to EIGHT scattered 2- / 4-byte stores per
  // element (two fp32 packs + 2 x 3 bf16 fragment pieces: 3.2 M partial-line writes per launch, ~20 us at the head of every
  // step).  Here a workgroup takes a tile of 8 cv x 8 cu x 16 taps through LDS and writes every destination in 16-byte pieces:
  // 8 consecutive cv of a (tap, cu) are contiguous in Wdown and form one fragment chunk of Fup, 8 consecutive cu of a (tap, cv)
  // are contiguous in Wup and form one chunk of Fdown — 1280 stores of 16 bytes per 1024 elements instead of 8192 small ones.
  // Same values in the same places.
  if (MVK_PACK_TILED && d.kind == 0 && d.Cu % 8 == 0 && d.Cv % 8 == 0 && d.ld_down % 4 == 0 && d.col_off % 4 == 0 &&
      mvk_dev_aligned16(d.Wref) && (!d.Wdown || mvk_dev_aligned16(d.Wdown)) && (!d.Wup || mvk_dev_aligned16(d.Wup))) {
    __shared__ __attribute__((aligned(16))) float T[8][8][16];  // [cv][cu][tap]
    const int tcu = d.Cu / 8, ntiles = (d.Cv / 8) * tcu, t = threadIdx.x;
    typedef float pf4 __attribute__((ext_vector_type(4)));
    typedef unsigned pu4 __attribute__((ext_vector_type(4)));
    float wmax = 0.f;  // max |Wref| over this workgroup's tiles: published once per workgroup (no second pass over the weight)
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const int cv0 = (tile / tcu) * 8, cu0 = (tile % tcu) * 8;
      __syncthreads();  // the previous tile's readers are done
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = t + 256 * r, cvl = e >> 7, rem = e & 127;  // 128 contiguous floats per cv
        const float v = d.Wref[((long long)(cv0 + cvl) * d.Cu + cu0) * 16 + rem];
        (&T[0][0][0])[e] = v;
        wmax = fmaxf(wmax, fabsf(v));
      }
      __syncthreads();
      const int pair = t >> 1, half = t & 1, tap = pair >> 3, l8 = pair & 7;
      const int kh = tap >> 2, kw = tap & 3, ph = 1 - (kh & 1), a = kh >> 1, pw = 1 - (kw & 1), b = kw >> 1;
      if (d.Wdown) {  // (tap, cu = l8): cv0 + 4 half .. + 3
        const pf4 v = {T[4 * half][l8][tap], T[4 * half + 1][l8][tap], T[4 * half + 2][l8][tap], T[4 * half + 3][l8][tap]};
        *reinterpret_cast<pf4*>(d.Wdown + (long long)(tap * d.Cu + cu0 + l8) * d.ld_down + d.col_off + cv0 + 4 * half) = v;
      }
      if (d.Wup) {  // (tap, cv = l8): cu0 + 4 half .. + 3
        const pf4 v = {T[l8][4 * half][tap], T[l8][4 * half + 1][tap], T[l8][4 * half + 2][tap], T[l8][4 * half + 3][tap]};
        *reinterpret_cast<pf4*>(d.Wup + ((long long)(ph * 2 + pw) * 4 * d.Cv + (a * 2 + b) * d.Cv + cv0 + l8) * d.Cu + cu0 + 4 * half) = v;
      }
      if (t < 128 && d.Fdown) {  // item (tap, cv = l8'): the 8 cu of the tile = one 16-byte chunk per piece
        const int tp = t >> 3, cvl = t & 7, cv = cv0 + cvl;
        const int chunks = d.Cu / 16, ntaps = 16 / chunks, ksplit = 16 / ntaps, nct = d.Cv / 32;
        const int ks = tp / ntaps, q = tp % ntaps, ct = cv / 32;
        const int role = nct * ksplit <= 4 ? ks * nct + ct : ct * 4 + ks;
        const int kk = q * chunks + cu0 / 16, lane = ((cu0 % 16) / 8) * 32 + cv % 32;
        pu4 pc[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned p0, p1, p2;
          mvk::bf3_split(T[cvl][2 * j][tp], T[cvl][2 * j + 1][tp], p0, p1, p2);
          pc[0][j] = p0;
          pc[1][j] = p1;
          pc[2][j] = p2;
        }
        pu4* f = static_cast<pu4*>(d.Fdown);
#pragma unroll
        for (int p = 0; p < 3; ++p) f[(((long long)role * 16 + kk) * 3 + p) * 64 + lane] = pc[p];
      }
      if (t >= 128 && d.Fup) {  // item (tap, cu): the 8 cv of the tile
        const int i = t - 128, tp = i >> 3, cul = i & 7, cu = cu0 + cul;
        const int kh2 = tp >> 2, kw2 = tp & 3, ph2 = 1 - (kh2 & 1), a2 = kh2 >> 1, pw2 = 1 - (kw2 & 1), b2 = kw2 >> 1;
        const int chunks = d.Cv / 16, ntaps = 16 / chunks, ksplit = 4 / ntaps, nct = d.Cu / 32;
        const int t4 = a2 * 2 + b2, ks = t4 / ntaps, q = t4 % ntaps, ct = cu / 32;
        const int role = (ph2 * 2 + pw2) * (ksplit * nct) + ks * nct + ct;
        const int kk = q * chunks + cv0 / 16, lane = ((cv0 % 16) / 8) * 32 + cu % 32;
        pu4 pc[3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned p0, p1, p2;
          mvk::bf3_split(T[2 * j][cul][tp], T[2 * j + 1][cul][tp], p0, p1, p2);
          pc[0][j] = p0;
          pc[1][j] = p1;
          pc[2][j] = p2;
        }
        pu4* f = static_cast<pu4*>(d.Fup);
#pragma unroll
        for (int p = 0; p < 3; ++p) f[(((long long)role * 16 + kk) * 3 + p) * 64 + lane] = pc[p];
      }
    }
    if (d.amax && blockIdx.x < ntiles) mvk::amax_publish(wmax, d.amax, amax_red);  // uniform per workgroup; <= ntiles atomics, most skipped
    return;
  }
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int tap = idx & 15;
    const int cu = (idx >> 4) % d.Cu;
    const int cv = (idx >> 4) / d.Cu;
    const float v = d.Wref[idx];
    if (d.kind == 1) {
      d.Wup[(long long)cv * 16 * d.Cu + tap * d.Cu + cu] = v;
      continue;
    }
    const int kh = tap >> 2, kw = tap & 3;
    if (d.Wdown) d.Wdown[(long long)(tap * d.Cu + cu) * d.ld_down + d.col_off + cv] = v;
    const int ph = 1 - (kh & 1), a = kh >> 1;
    const int pw = 1 - (kw & 1), b = kw >> 1;
    if (d.Wup) d.Wup[((long long)(ph * 2 + pw) * 4 * d.Cv + (a * 2 + b) * d.Cv + cv) * d.Cu + cu] = v;
    if (d.Fdown || d.Fup) {
      // bf16-piece MFMA B fragments of the register-stationary convolution kernels (imgconv.hip): element (k, n) of
      // the slice of wave `role` sits at [role][k-step][piece][lane = (k % 16 / 8) * 32 + n % 32][k % 8]
      unsigned p0, p1, p2;
      mvk::bf3_split(v, 0.f, p0, p1, p2);
      const unsigned short pc[3] = {(unsigned short)p0, (unsigned short)p1, (unsigned short)p2};
      if (d.Fdown) {  // GEMM k = (tap, cu), n = cv
        const int chunks = d.Cu / 16, ntaps = 16 / chunks, ksplit = 16 / ntaps, nct = d.Cv / 32;
        const int ks = tap / ntaps, q = tap % ntaps, ct = cv / 32;
        const int role = nct * ksplit <= 4 ? ks * nct + ct : ct * 4 + ks;
        const int kk = q * chunks + cu / 16, lane = ((cu % 16) / 8) * 32 + cv % 32;
        unsigned short* f = static_cast<unsigned short*>(d.Fdown);
#pragma unroll
        for (int p = 0; p < 3; ++p) f[((((long long)role * 16 + kk) * 3 + p) * 64 + lane) * 8 + cu % 8] = pc[p];
      }
      if (d.Fup) {  // per output parity class: k = (tap (a, b), cv), n = cu
        const int chunks = d.Cv / 16, ntaps = 16 / chunks, ksplit = 4 / ntaps, nct = d.Cu / 32;
        const int t4 = a * 2 + b, ks = t4 / ntaps, q = t4 % ntaps, ct = cu / 32;
        const int role = (ph * 2 + pw) * (ksplit * nct) + ks * nct + ct;
        const int kk = q * chunks + cv / 16, lane = ((cv % 16) / 8) * 32 + cu % 32;
        unsigned short* f = static_cast<unsigned short*>(d.Fup);
#pragma unroll
        for (int p = 0; p < 3; ++p) f[((((long long)role * 16 + kk) * 3 + p) * 64 + lane) * 8 + cv % 8] = pc[p];
      }
    }
  }
  pack_amax(d, total, amax_red);
}

// Wref[ci][co][tap] -> Wp[ci][tap*Cout + co]
__global__ void pack_unflatten_kernel(const float* __restrict__ Wref, int Cin, int Cout, float* __restrict__ Wp) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  int total = Cin * Cout * 16;
  if (idx >= total) return;
  int tap = idx & 15;
  int co = (idx >> 4) % Cout;
  int ci = (idx >> 4) / Cout;
  Wp[(long long)ci * 16 * Cout + tap * Cout + co] = Wref[idx];
}

// ---- layout ----------------------------------------------------------------------------------------------
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, long long total, int c,
                                    int hw) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // destination index (n, p, ch)
  if (idx >= total) return;
  int ch = idx % c;
  long long t = idx / c;
  int p = t % hw;
  long long n = t / hw;
  dst[idx] = src[(n * c + ch) * hw + p];
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, long long total, int c,
                                    int hw) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // destination index (n, ch, p)
  if (idx >= total) return;
  int p = idx % hw;
  long long t = idx / hw;
  int ch = t % c;
  long long n = t / c;
  dst[idx] = src[(n * hw + p) * c + ch];
}

__global__ void act_bwd_kernel(float* __restrict__ dY, const float* __restrict__ Y, long long n, int act) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dY[i] *= mvk_act_grad_from_out(Y[i], act);
}

__global__ __launch_bounds__(256) void act_bwd4_kernel(float4* __restrict__ dY, const float4* __restrict__ Y, long long n4,
                                                       int act) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 g = dY[i];
  const float4 y = Y[i];
  g.x *= mvk_act_grad_from_out(y.x, act);
  g.y *= mvk_act_grad_from_out(y.y, act);
  g.z *= mvk_act_grad_from_out(y.z, act);
  g.w *= mvk_act_grad_from_out(y.w, act);
  dY[i] = g;
}

// out = a * g * act'(Y): the gradient through `a * act(.)` in one pass (the 0.1 of a ResNet block's residual branch and the
// LeakyReLU behind its second convolution: an axpby pass + an in-place act_bwd pass before)
__global__ __launch_bounds__(256) void act_bwd_scaled_kernel(const float* __restrict__ g, float a, const float* __restrict__ Y,
                                                             long long n, int act, float* __restrict__ out) {
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 + 3 < n && ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
    const float4 gv = *reinterpret_cast<const float4*>(g + i4), y = *reinterpret_cast<const float4*>(Y + i4);
    float4 r;
    r.x = a * gv.x * mvk_act_grad_from_out(y.x, act);
    r.y = a * gv.y * mvk_act_grad_from_out(y.y, act);
    r.z = a * gv.z * mvk_act_grad_from_out(y.z, act);
    r.w = a * gv.w * mvk_act_grad_from_out(y.w, act);
    *reinterpret_cast<float4*>(out + i4) = r;
  } else {
    for (long long i = i4; i < n && i < i4 + 4; ++i) out[i] = a * g[i] * mvk_act_grad_from_out(Y[i], act);
  }
}

__global__ void scale_kernel(float* __restrict__ buf, long long n, const float* __restrict__ g) {
  float s = *g;
  if (s == 1.0f) return;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) buf[i] *= s;
}

// Backward of the loss assembly in ONE launch: buffers with fill == 0 are multiplied in place by the upstream gradient
// g (skipped when g == 1, the loss.backward() default); buffers with fill != 0 are set to g * coef (the gradient of a
// row-sum term that entered the loss with a constant weight).  blockIdx.y selects the buffer.
struct SeedJobs {
  mvk_seed_desc j[MVK_SEED_MAX];
};
__global__ __launch_bounds__(256) void loss_seed_kernel(const SeedJobs jobs, const float* __restrict__ g) {
  const mvk_seed_desc& d = jobs.j[blockIdx.y];
  const float s = *g;
  if (!d.fill && s == 1.0f) return;
  const float v = s * d.coef;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < d.n; i += stride) d.buf[i] = d.fill ? v : d.buf[i] * s;
}

// ---- U[n,Cu,2h,2w] (NCHW) = act(convT(V[n,h,w,Cv]) + b), Cu <= 4: direct, one thread per output pixel ------
template <int CU>
__global__ __launch_bounds__(256) void up_nchw_small_kernel(const float* __restrict__ V,
                                                            const float* __restrict__ Wref,
                                                            const float* __restrict__ bias, float* __restrict__ U,
                                                            int n, int h, int w, int Cv, int act) {
  extern __shared__ float wl[];  // [tap][cv][CU]
  for (int i = threadIdx.x; i < 16 * Cv * CU; i += blockDim.x) {
    int cu = i % CU;
    int cv = (i / CU) % Cv;
    int tap = i / (CU * Cv);
    wl[i] = Wref[((long long)cv * CU + cu) * 16 + tap];
  }
  __syncthreads();
  const int H = 2 * h, W2 = 2 * w;
  long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)n * H * W2;
  if (pix >= total) return;
  int ow = pix % W2;
  long long t = pix / W2;
  int oh = t % H;
  long long img = t / H;
  float acc[CU];
#pragma unroll
  for (int c = 0; c < CU; ++c) acc[c] = bias ? bias[c] : 0.f;
  const int ph = oh & 1, pw = ow & 1;
  const int i0 = oh >> 1, j0 = ow >> 1;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    int kh = (1 - ph) + 2 * a;
    int ih = i0 + ph - a;
    if (ih < 0 || ih >= h) continue;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      int kw = (1 - pw) + 2 * b;
      int iw = j0 + pw - b;
      if (iw < 0 || iw >= w) continue;
      const float* vp = V + ((img * h + ih) * w + iw) * Cv;
      const float* wp = wl + (kh * 4 + kw) * Cv * CU;
      for (int cv = 0; cv < Cv; cv += 4) {
        float4 x = *reinterpret_cast<const float4*>(vp + cv);
#pragma unroll
        for (int c = 0; c < CU; ++c) {
          acc[c] = fmaf(x.x, wp[(cv + 0) * CU + c], acc[c]);
          acc[c] = fmaf(x.y, wp[(cv + 1) * CU + c], acc[c]);
          acc[c] = fmaf(x.z, wp[(cv + 2) * CU + c], acc[c]);
          acc[c] = fmaf(x.w, wp[(cv + 3) * CU + c], acc[c]);
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CU; ++c) U[((img * CU + c) * H + oh) * W2 + ow] = mvk_act(acc[c], act);
}

// ---- Adam on flat buffers (torch.optim.Adam single-tensor rule, amsgrad=False) -----------------------------
struct AdamScalars {
  float step_size, omb1, b2, omb2, eps, wd, bc2_sqrt, gscale;
};
// one element of the update; vmx = the amsgrad running maximum (read and updated only when AMSGRAD)
template <bool AMSGRAD>
__device__ __forceinline__ void adam_one(float& pi, float gi, float& mi, float& vi, float& vmx, const AdamScalars& a) {
  gi *= a.gscale;
  if (a.wd != 0.f) gi = fmaf(a.wd, pi, gi);
  mi = mi + (gi - mi) * a.omb1;            // exp_avg.lerp_(grad, 1 - beta1), weight < 0.5 branch
  vi = fmaf(a.omb2, gi * gi, vi * a.b2);   // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
  float vd = vi;
  if (AMSGRAD) {  // max_exp_avg_sq = max(max_exp_avg_sq, exp_avg_sq); the denominator uses the running maximum
    vd = fmaxf(vmx, vi);
    vmx = vd;
  }
  const float denom = sqrtf(vd) / a.bc2_sqrt + a.eps;
  pi = pi - a.step_size * (mi / denom);
}

// ZERO: the gradient is cleared as it is consumed (the next step's zero_grad pass and its launch disappear)
template <bool AMSGRAD, bool ZERO>
__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            float* __restrict__ vmax, long long n, const AdamScalars a, AdamScalars* __restrict__ pub) {
  if (pub && blockIdx.x == 0 && threadIdx.x == 0) *pub = a;  // mvk_adam_step_pub: the scalars of THIS update, for a later mvk_adam_step_dev
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float pi = p[i], mi = m[i], vi = v[i], vmx = AMSGRAD ? vmax[i] : 0.f;
    adam_one<AMSGRAD>(pi, g[i], mi, vi, vmx, a);
    p[i] = pi, m[i] = mi, v[i] = vi;
    if (AMSGRAD) vmax[i] = vmx;
    if (ZERO) g[i] = 0.f;
  }
}

// 16 bytes per lane and tensor (n4 = n / 4; all buffers 16-byte aligned)
template <bool AMSGRAD, bool ZERO>
__global__ __launch_bounds__(256) void adam4_kernel(float4* __restrict__ p, float4* __restrict__ g, float4* __restrict__ m,
                                                    float4* __restrict__ v, float4* __restrict__ vmax, long long n4,
                                                    const AdamScalars a, AdamScalars* __restrict__ pub) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pub && i == 0) *pub = a;
  if (i >= n4) return;
  float4 pi = p[i], mi = m[i], vi = v[i];
  const float4 gi = g[i];
  float4 vx = AMSGRAD ? vmax[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  adam_one<AMSGRAD>(pi.x, gi.x, mi.x, vi.x, vx.x, a);
  adam_one<AMSGRAD>(pi.y, gi.y, mi.y, vi.y, vx.y, a);
  adam_one<AMSGRAD>(pi.z, gi.z, mi.z, vi.z, vx.z, a);
  adam_one<AMSGRAD>(pi.w, gi.w, mi.w, vi.w, vx.w, a);
  p[i] = pi, m[i] = mi, v[i] = vi;
  if (AMSGRAD) vmax[i] = vx;
  if (ZERO) g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- the same update with its scalars in DEVICE memory (capturable: trainers/graph.py puts the optimizer into the hipGraph) ----
// state (8 doubles, owned by the caller): [0] lr, [1] beta1, [2] beta2, [3] eps, [4] weight_decay, [5] grad_scale,
// [6] step (the number of updates applied so far), [7] unused.  adam_prepare_kernel (one thread) advances the step and derives
// the eight floats the update kernel needs — in double, like the host path and torch.optim.Adam's Python side — so a replayed
// graph needs no host scalar: a learning-rate scheduler writes state[0], nothing else changes between replays.
__global__ void adam_prepare_kernel(double* __restrict__ state, AdamScalars* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double lr = state[0], b1 = state[1], b2 = state[2];
  const double step = state[6] + 1.0;
  state[6] = step;
  const double bc1 = 1.0 - pow(b1, step);
  const double bc2 = 1.0 - pow(b2, step);
  AdamScalars a;
  a.step_size = (float)(lr / bc1);
  a.omb1 = (float)(1.0 - b1);
  a.b2 = (float)b2;
  a.omb2 = (float)(1.0 - b2);
  a.eps = (float)state[3];
  a.wd = (float)state[4];
  a.bc2_sqrt = (float)sqrt(bc2);
  a.gscale = (float)state[5];
  *out = a;
}

template <bool AMSGRAD, bool ZERO>
__global__ __launch_bounds__(256) void adam4_dev_kernel(float4* __restrict__ p, float4* __restrict__ g, float4* __restrict__ m,
                                                        float4* __restrict__ v, float4* __restrict__ vmax, long long n4,
                                                        const AdamScalars* __restrict__ sc) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const AdamScalars a = *sc;  // uniform address: scalar loads
  float4 pi = p[i], mi = m[i], vi = v[i];
  const float4 gi = g[i];
  float4 vx = AMSGRAD ? vmax[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  adam_one<AMSGRAD>(pi.x, gi.x, mi.x, vi.x, vx.x, a);
  adam_one<AMSGRAD>(pi.y, gi.y, mi.y, vi.y, vx.y, a);
  adam_one<AMSGRAD>(pi.z, gi.z, mi.z, vi.z, vx.z, a);
  adam_one<AMSGRAD>(pi.w, gi.w, mi.w, vi.w, vx.w, a);
  p[i] = pi, m[i] = mi, v[i] = vi;
  if (AMSGRAD) vmax[i] = vx;
  if (ZERO) g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- scalar assembly ----------------------------------------------------------------------------------------
struct TermTable {
  mvk_term_desc t[MVK_MAX_TERMS];
  int n;
  float loss_sum_scale;
};

// All terms in one pass: a thread accumulates its strided partial sums of 8 terms at a time (their loads overlap), the
// waves' sums go to LDS and ONE barrier later thread i finishes term i (the waves in order: the same summation order as a
// term-by-term loop with a barrier per term, which cost ~1.5 us per term on the chain between forward and backward).
// gfill: where a term's gradient w.r.t. its rows is the constant coef * lossw (KL rows), it is written here, so that a
// backward pass whose upstream gradient is known to be 1 launches nothing (ReconLossFn.backward).
#define MVK_TERMS_WG 32  // workgroups of the multi-workgroup form (include/mvk.h: MVK_REDUCE_TERMS_WS_FLOATS)

// ws == null: ONE workgroup (gridDim.x == 1).  ws given: gridDim.x workgroups, workgroup g sums the g-th contiguous slice of
// every term (slice = ceil(n / G) rounded up to 1024 entries), writes its per-term partials to ws[1 + g * MVK_MAX_TERMS + i],
// takes a ticket (ws[0], an unsigned counter that is 0 between launches) and the LAST workgroup to arrive adds the partials in
// workgroup order: the result does not depend on which workgroup that is.
template <int NT>
__global__ __launch_bounds__(NT) void reduce_terms_kernel(const TermTable tt, float* __restrict__ out,
                                                            float* __restrict__ loss_out, float* __restrict__ ws,
                                                            mvk_prof_slot* prof) {
  mvk_prof_begin(prof);
  __shared__ float red[MVK_MAX_TERMS][NT / 64];
  __shared__ float vals[MVK_MAX_TERMS];
  __shared__ unsigned ticket;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int G = gridDim.x, g = blockIdx.x;
  for (int i0 = 0; i0 < tt.n; i0 += 8) {
    float s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s[u] = 0.f;
      if (i0 + u < tt.n) {
        const mvk_term_desc& t = tt.t[i0 + u];
        const long long slice = ((t.n + G - 1) / G + NT - 1) / NT * NT;
        const long long lo = (long long)g * slice, hi = lo + slice < t.n ? lo + slice : t.n;
        // four loads in flight per thread (a term of 35840 rows was 35 dependent load latencies: 18 us for 160 KB)
        long long j = lo + threadIdx.x;
        for (; j + 3 * NT < hi; j += 4 * NT) {
          float v0 = t.v[j], v1 = t.v[j + NT], v2 = t.v[j + 2 * NT], v3 = t.v[j + 3 * NT];
          if (t.mask) {
            v0 = t.mask[j % t.period] ? v0 : 0.f;
            v1 = t.mask[(j + NT) % t.period] ? v1 : 0.f;
            v2 = t.mask[(j + 2 * NT) % t.period] ? v2 : 0.f;
            v3 = t.mask[(j + 3 * NT) % t.period] ? v3 : 0.f;
          }
          s[u] += (v0 + v1) + (v2 + v3);
        }
        for (; j < hi; j += NT) {
          float v = t.v[j];
          if (t.mask) v = t.mask[j % t.period] ? v : 0.f;
          s[u] += v;
        }
        if (t.gfill) {
          const float gv = t.coef * t.lossw;
          for (long long j2 = lo + threadIdx.x; j2 < hi; j2 += NT) t.gfill[j2] = gv;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float w = wave_sum(s[u]);
      if (lane == 0 && i0 + u < tt.n) red[i0 + u][wave] = w;
    }
  }
  __syncthreads();
  if (ws) {  // publish this workgroup's partials; only the last arrival goes on
    if ((int)threadIdx.x < tt.n) {
      float tot = 0.f;
      for (int wv = 0; wv < NT / 64; ++wv) tot += red[threadIdx.x][wv];
      __hip_atomic_store(ws + 1 + g * MVK_MAX_TERMS + threadIdx.x, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0)
      ticket = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(ws), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket != (unsigned)(G - 1)) {
      mvk_prof_end(prof);
      return;
    }
    __threadfence();
    if ((int)threadIdx.x < tt.n) {
      float tot = 0.f;
      for (int q = 0; q < G; ++q)
        tot += __hip_atomic_load(ws + 1 + q * MVK_MAX_TERMS + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      vals[threadIdx.x] = tot * tt.t[threadIdx.x].coef;
    }
    if (threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<unsigned*>(ws), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if ((int)threadIdx.x < tt.n) {
    float tot = 0.f;
    for (int wv = 0; wv < NT / 64; ++wv) tot += red[threadIdx.x][wv];
    vals[threadIdx.x] = tot * tt.t[threadIdx.x].coef;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float loss = 0.f;
    for (int i = 0; i < tt.n; ++i) {
      out[i] = vals[i];
      loss += tt.t[i].lossw * vals[i];
    }
    out[tt.n] = loss;
    out[tt.n + 1] = loss * tt.loss_sum_scale;
    if (loss_out) *loss_out = loss;
  }
  mvk_prof_end(prof);
}

static int grid_for(long long n, int block, int cap = 2048) {
  long long g = (n + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

// ---- device-timestamp profiler ---------------------------------------------------------------------------------------
static mvk_prof_slot* g_prof_slots = nullptr;
static int g_prof_n = 0, g_prof_used = 0;
static int32_t* g_prof_kinds = nullptr;
static double* g_prof_work = nullptr;
__global__ __launch_bounds__(64) void prof_fold_kernel(mvk_prof_slot* s) {
  const unsigned long long now = (unsigned long long)wall_clock64();  // the launch in front has fully retired
  const int l = threadIdx.x & (MVK_PROF_ENTRIES - 1);
  unsigned long long t0 = s->w[8 + 8 * l], t1 = s->w[8 + 8 * MVK_PROF_ENTRIES + 8 * l];
#pragma unroll
  for (int off = MVK_PROF_ENTRIES / 2; off > 0; off >>= 1) {
    const unsigned long long a = __shfl_xor(t0, off, 64), b = __shfl_xor(t1, off, 64);
    t0 = a < t0 ? a : t0;
    t1 = b > t1 ? b : t1;
  }
  if (threadIdx.x == 0 && t1 > t0) {
    s->w[0] += t1 - t0;   // first workgroup in -> last workgroup out
    s->w[1] += 1;
    s->w[2] += now - t0;  // ... -> first instruction of the next kernel on the stream (stores drained, + one boundary)
  }
  if (threadIdx.x < MVK_PROF_ENTRIES) {  // re-arm
    s->w[8 + 8 * l] = ~0ull;
    s->w[8 + 8 * MVK_PROF_ENTRIES + 8 * l] = 0ull;
  }
}
__global__ __launch_bounds__(64) void prof_stamp_kernel(unsigned long long* out) {
  const unsigned long long t = (unsigned long long)wall_clock64();
  if (threadIdx.x == 0) {
    out[0] = t;                                      // first instruction
    out[1] = (unsigned long long)wall_clock64();     // last instruction (as mvk_prof_end stamps it)
  }
}
namespace mvk {
mvk_prof_slot* prof_next(int kind, double work) {
  if (!g_prof_slots || g_prof_used >= g_prof_n) return nullptr;
  const int i = g_prof_used++;
  if (g_prof_kinds) g_prof_kinds[i] = kind;
  if (g_prof_work) g_prof_work[i] = work;
  return g_prof_slots + i;
}
void prof_fold(mvk_prof_slot* slot, hipStream_t s) {
  if (slot) hipLaunchKernelGGL(prof_fold_kernel, dim3(1), dim3(64), 0, s, slot);
}
}  // namespace mvk

extern "C" {

int mvk_version(void) { return 100; }

int mvk_prof_enable(void* device_slots, int nslots, int32_t* host_kinds, double* host_work) {
  g_prof_slots = static_cast<mvk_prof_slot*>(device_slots);
  g_prof_n = device_slots ? nslots : 0;
  g_prof_used = 0;
  g_prof_kinds = host_kinds;
  g_prof_work = host_work;
  return MVK_OK;
}
int mvk_prof_count(void) { return g_prof_used; }
int mvk_prof_calibrate(void* device_ticks, int n, void* stream) {
  if (!device_ticks || n < 2) return MVK_EINVAL;
  unsigned long long* out = static_cast<unsigned long long*>(device_ticks);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(prof_stamp_kernel, dim3(1), dim3(64), 0, mvk_stream(stream), out + 2 * i);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}
int mvk_prof_clock_khz(void) {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
  return khz;
}

int mvk_pack_conv4s2_weight(const float* Wref, int Cv, int Cu, float* Wdown, int ld_down, int col_off,
                            float* Wup, void* stream) {
  if (!Wref || Cv <= 0 || Cu <= 0 || (!Wdown && !Wup)) return MVK_EINVAL;
  int total = Cv * Cu * 16;
  hipLaunchKernelGGL(pack_conv4s2_kernel, dim3((total + 255) / 256), dim3(256), 0, mvk_stream(stream), Wref, Cv, Cu,
                     Wdown, ld_down > 0 ? ld_down : Cv, col_off, Wup);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int64_t mvk_imgconv_frag_bytes(int Cu, int Cv) {
  if ((Cu == 32 && Cv == 64) || (Cu == 64 && Cv == 128)) return (int64_t)16 * Cu * Cv * 3 * 2;  // every element, 3 bf16 pieces
  return 0;
}

int mvk_pack_weights(const mvk_pack_desc* jobs, int n, void* stream) {
  if (!jobs || n <= 0 || n > MVK_PACK_MAX) return MVK_EINVAL;
  PackJobs pj{};
  int maxtot = 0;
  for (int i = 0; i < n; ++i) {
    const mvk_pack_desc& d = jobs[i];
    if (!d.Wref || d.Cv <= 0 || d.Cu <= 0 || d.kind < 0 || d.kind > 2 || (!d.Wdown && !d.Wup && !d.Fdown && !d.Fup) ||
        (d.kind == 1 && !d.Wup))
      return MVK_EINVAL;
    if ((d.Fdown || d.Fup) && (d.kind != 0 || mvk_imgconv_frag_bytes(d.Cu, d.Cv) == 0)) return MVK_EINVAL;
    pj.j[i] = d;
    if (pj.j[i].ld_down <= 0) pj.j[i].ld_down = d.Cv;
    const int tot = d.Cv * d.Cu * 16;
    if (tot > maxtot) maxtot = tot;
  }
  int gx = (maxtot + 255) / 256;
  if (gx > 256) gx = 256;  // grid-stride inside
  hipLaunchKernelGGL(pack_multi_kernel, dim3(gx, n), dim3(256), 0, mvk_stream(stream), pj);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_pack_unflatten_weight(const float* Wref, int Cin, int Cout, float* Wp, void* stream) {
  if (!Wref || !Wp || Cin <= 0 || Cout <= 0) return MVK_EINVAL;
  int total = Cin * Cout * 16;
  hipLaunchKernelGGL(pack_unflatten_kernel, dim3((total + 255) / 256), dim3(256), 0, mvk_stream(stream), Wref, Cin,
                     Cout, Wp);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_nchw_to_nhwc(const float* src, float* dst, int n, int c, int h, int w, void* stream) {
  if (!src || !dst) return MVK_EINVAL;
  long long total = (long long)n * c * h * w;
  if (total == 0) return MVK_OK;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, mvk_stream(stream), src,
                     dst, total, c, h * w);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_nhwc_to_nchw(const float* src, float* dst, int n, int c, int h, int w, void* stream) {
  if (!src || !dst) return MVK_EINVAL;
  long long total = (long long)n * c * h * w;
  if (total == 0) return MVK_OK;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, mvk_stream(stream), src,
                     dst, total, c, h * w);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_act_bwd(float* dY, const float* Y, int64_t n, int act, void* stream) {
  if (!dY || !Y) return MVK_EINVAL;
  if (n == 0) return MVK_OK;
  if (n % 4 == 0 && mvk_aligned16(dY) && mvk_aligned16(Y))
    hipLaunchKernelGGL(act_bwd4_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, mvk_stream(stream),
                       reinterpret_cast<float4*>(dY), reinterpret_cast<const float4*>(Y), (long long)(n / 4), act);
  else
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, mvk_stream(stream), dY, Y, (long long)n, act);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_act_bwd_scaled(const float* g, float a, const float* Y, int act, float* out, int64_t n, void* stream) {
  if (!g || !Y || !out || n < 0) return MVK_EINVAL;
  if (n == 0) return MVK_OK;
  hipLaunchKernelGGL(act_bwd_scaled_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, mvk_stream(stream), g, a, Y,
                     (long long)n, act, out);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_scale_by_device_scalar(float* buf, int64_t n, const float* gscale, void* stream) {
  if (!buf || !gscale) return MVK_EINVAL;
  if (n == 0) return MVK_OK;
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n, 256)), dim3(256), 0, mvk_stream(stream), buf, (long long)n, gscale);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_loss_backward_seed(const mvk_seed_desc* jobs, int n, const float* gscale, void* stream) {
  if (!jobs || !gscale || n <= 0 || n > MVK_SEED_MAX) return MVK_EINVAL;
  SeedJobs sj{};
  long long maxn = 0;
  for (int i = 0; i < n; ++i) {
    if (!jobs[i].buf || jobs[i].n < 0) return MVK_EINVAL;
    sj.j[i] = jobs[i];
    if (jobs[i].n > maxn) maxn = jobs[i].n;
  }
  if (maxn == 0) return MVK_OK;
  long long gx = (maxn + 255) / 256;
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(loss_seed_kernel, dim3((unsigned)gx, n), dim3(256), 0, mvk_stream(stream), sj, gscale);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_conv4s2_up_nchw_small(const float* V, const float* Wref, const float* bias, float* U, int n, int h, int w,
                              int Cu, int Cv, int act, void* stream) {
  if (!V || !Wref || !U || Cu <= 0 || Cu > 4 || Cv <= 0 || (Cv % 4) != 0 || !mvk_aligned16(V)) return MVK_EINVAL;
  long long total = (long long)n * 4 * h * w;
  if (total == 0) return MVK_OK;
  dim3 grid((unsigned)((total + 255) / 256));
  size_t lds = (size_t)16 * Cv * Cu * sizeof(float);
  hipStream_t s = mvk_stream(stream);
  switch (Cu) {
    case 1: hipLaunchKernelGGL(up_nchw_small_kernel<1>, grid, dim3(256), lds, s, V, Wref, bias, U, n, h, w, Cv, act); break;
    case 2: hipLaunchKernelGGL(up_nchw_small_kernel<2>, grid, dim3(256), lds, s, V, Wref, bias, U, n, h, w, Cv, act); break;
    case 3: hipLaunchKernelGGL(up_nchw_small_kernel<3>, grid, dim3(256), lds, s, V, Wref, bias, U, n, h, w, Cv, act); break;
    default: hipLaunchKernelGGL(up_nchw_small_kernel<4>, grid, dim3(256), lds, s, V, Wref, bias, U, n, h, w, Cv, act); break;
  }
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

// the scalar arithmetic of one update: in double like the Python side of torch.optim.Adam, cast once
static AdamScalars adam_scalars(double lr, double beta1, double beta2, double eps, double weight_decay, int step, double grad_scale) {
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  return AdamScalars{(float)(lr / bc1), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                     (float)eps,        (float)weight_decay,  (float)sqrt(bc2), (float)grad_scale};
}

__global__ void adam_publish_kernel(AdamScalars* __restrict__ pub, const AdamScalars a) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *pub = a;
}

int mvk_adam_step_pub(float* p, float* g, float* m, float* v, float* vmax, int64_t n, double lr, double beta1, double beta2,
                      double eps, double weight_decay, int step, double grad_scale, int zero_grad, float* publish, void* stream) {
  if (n < 0 || step < 1 || (n > 0 && (!p || !g || !m || !v)) || (publish && !mvk_aligned16(publish))) return MVK_EINVAL;
  const AdamScalars a = adam_scalars(lr, beta1, beta2, eps, weight_decay, step, grad_scale);
  hipStream_t s = mvk_stream(stream);
  AdamScalars* pub = reinterpret_cast<AdamScalars*>(publish);
  if (n == 0) {
    if (pub) {
      hipLaunchKernelGGL(adam_publish_kernel, dim3(1), dim3(64), 0, s, pub, a);
      MVK_CHECK_LAUNCH();
    }
    return MVK_OK;
  }
  const bool v4 = n % 4 == 0 && mvk_aligned16(p) && mvk_aligned16(g) && mvk_aligned16(m) && mvk_aligned16(v) &&
                  (!vmax || mvk_aligned16(vmax));
#define MVK_ADAM_LAUNCH(AMS_, Z_)                                                                                              \
  do {                                                                                                                         \
    if (v4)                                                                                                                    \
      hipLaunchKernelGGL((adam4_kernel<AMS_, Z_>), dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s,                     \
                         reinterpret_cast<float4*>(p), reinterpret_cast<float4*>(g), reinterpret_cast<float4*>(m),             \
                         reinterpret_cast<float4*>(v), reinterpret_cast<float4*>(vmax), (long long)(n / 4), a, pub);           \
    else                                                                                                                       \
      hipLaunchKernelGGL((adam_kernel<AMS_, Z_>), dim3(grid_for(n, 256)), dim3(256), 0, s, p, g, m, v, vmax, (long long)n, a,  \
                         pub);                                                                                                 \
  } while (0)
  if (vmax && zero_grad) MVK_ADAM_LAUNCH(true, true);
  else if (vmax) MVK_ADAM_LAUNCH(true, false);
  else if (zero_grad) MVK_ADAM_LAUNCH(false, true);
  else MVK_ADAM_LAUNCH(false, false);
#undef MVK_ADAM_LAUNCH
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_adam_step_fused(float* p, float* g, float* m, float* v, float* vmax, int64_t n, double lr, double beta1, double beta2,
                        double eps, double weight_decay, int step, double grad_scale, int zero_grad, void* stream) {
  if (!p || !g || !m || !v || step < 1) return MVK_EINVAL;
  return mvk_adam_step_pub(p, g, m, v, vmax, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, zero_grad, nullptr, stream);
}

// The scalars under which mvk_adam_step_dev changes NOTHING (p, m, v, vmax bit for bit; a finite gradient is consumed — and
// cleared when zero_grad is set): step_size 0, 1 - beta1 = 0, beta2 = 1, 1 - beta2 = 0, eps 1, weight_decay 0, sqrt(bc2) 1,
// grad_scale 0.  What a rotated step's update of the late leaves reads before the first step and after a drain.
int mvk_adam_identity(float* scalars, void* stream) {
  if (!scalars || !mvk_aligned16(scalars)) return MVK_EINVAL;
  const AdamScalars a{0.f, 0.f, 1.f, 0.f, 1.f, 0.f, 1.f, 0.f};
  hipLaunchKernelGGL(adam_publish_kernel, dim3(1), dim3(64), 0, mvk_stream(stream), reinterpret_cast<AdamScalars*>(scalars), a);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_adam_prepare(double* state, float* scalars, void* stream) {
  if (!state || !scalars || !mvk_aligned16(scalars)) return MVK_EINVAL;
  hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, mvk_stream(stream), state, reinterpret_cast<AdamScalars*>(scalars));
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_adam_step_dev(float* p, float* g, float* m, float* v, float* vmax, int64_t n, const float* scalars, int zero_grad,
                      void* stream) {
  if (!p || !g || !m || !v || !scalars) return MVK_EINVAL;
  if (n == 0) return MVK_OK;
  if (n % 4 != 0 || !mvk_aligned16(p) || !mvk_aligned16(g) || !mvk_aligned16(m) || !mvk_aligned16(v) || (vmax && !mvk_aligned16(vmax)))
    return MVK_EINVAL;  // flat buffers of trainers.FlatParams are padded to 64 floats and 256-byte aligned
  hipStream_t s = mvk_stream(stream);
  const AdamScalars* sc = reinterpret_cast<const AdamScalars*>(scalars);
#define MVK_ADAM_DEV(AMS_, Z_)                                                                                              \
  hipLaunchKernelGGL((adam4_dev_kernel<AMS_, Z_>), dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s,                  \
                     reinterpret_cast<float4*>(p), reinterpret_cast<float4*>(g), reinterpret_cast<float4*>(m),              \
                     reinterpret_cast<float4*>(v), reinterpret_cast<float4*>(vmax), (long long)(n / 4), sc)
  if (vmax && zero_grad) MVK_ADAM_DEV(true, true);
  else if (vmax) MVK_ADAM_DEV(true, false);
  else if (zero_grad) MVK_ADAM_DEV(false, true);
  else MVK_ADAM_DEV(false, false);
#undef MVK_ADAM_DEV
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_adam_step_amsgrad(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, double lr, double beta1,
                          double beta2, double eps, double weight_decay, int step, double grad_scale, void* stream) {
  return mvk_adam_step_fused(p, const_cast<float*>(g), m, v, vmax, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, 0,
                             stream);
}

int mvk_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                  double eps, double weight_decay, int step, double grad_scale, void* stream) {
  return mvk_adam_step_amsgrad(p, g, m, v, nullptr, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, stream);
}

int mvk_reduce_terms_ws(const mvk_term_desc* terms, int n_terms, float loss_sum_scale, float* out, float* loss_out, float* ws,
                        int64_t ws_floats, void* stream) {
  if (!terms || !out || n_terms < 1 || n_terms > MVK_MAX_TERMS) return MVK_EINVAL;
  TermTable tt;
  long long longest = 0;
  for (int i = 0; i < n_terms; ++i) {
    tt.t[i] = terms[i];
    if (!terms[i].v || terms[i].n < 0) return MVK_EINVAL;
    if (tt.t[i].period <= 0) tt.t[i].period = 1;
    longest = terms[i].n > longest ? terms[i].n : longest;
  }
  tt.n = n_terms;
  tt.loss_sum_scale = loss_sum_scale;
  double bytes = 0.0;
  for (int i = 0; i < n_terms; ++i) bytes += 4.0 * (double)terms[i].n * (terms[i].gfill ? 2.0 : 1.0);
  // several workgroups only where a term is long enough to give each of them a slice, and the caller lent the workspace
  int G = 1;
  if (ws && ws_floats >= MVK_REDUCE_TERMS_WS_FLOATS && longest > 4096) {
    G = (int)((longest + 2047) / 2048);
    if (G > MVK_TERMS_WG) G = MVK_TERMS_WG;
  }
  mvk_prof_slot* prof = mvk::prof_next(11, bytes);
  if (G > 1)
    hipLaunchKernelGGL(reduce_terms_kernel<256>, dim3(G), dim3(256), 0, mvk_stream(stream), tt, out, loss_out, ws, prof);
  else
    hipLaunchKernelGGL(reduce_terms_kernel<1024>, dim3(1), dim3(1024), 0, mvk_stream(stream), tt, out, loss_out, nullptr, prof);
  MVK_CHECK_LAUNCH();
  mvk::prof_fold(prof, mvk_stream(stream));
  return MVK_OK;
}

int mvk_reduce_terms(const mvk_term_desc* terms, int n_terms, float loss_sum_scale, float* out, float* loss_out,
                     void* stream) {
  return mvk_reduce_terms_ws(terms, n_terms, loss_sum_scale, out, loss_out, nullptr, 0, stream);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// pre-split tensor format (three bf16 planes), see mvk.h
// ---------------------------------------------------------------------------------------------------------
namespace {
typedef __bf16 mvk_bf16x2 __attribute__((ext_vector_type(2)));
typedef float mvk_f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void f32_to_bf3_kernel(const float* __restrict__ x, long long n,
                                                         unsigned short* __restrict__ planes) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= n) return;
  const float a = x[i], b = (i + 1 < n) ? x[i + 1] : 0.f;
  const mvk_bf16x2 p0 = __builtin_convertvector(mvk_f32x2{a, b}, mvk_bf16x2);
  const unsigned u0 = __builtin_bit_cast(unsigned, p0);
  const float ra = a - __uint_as_float(u0 << 16), rb = b - __uint_as_float(u0 & 0xffff0000u);
  const mvk_bf16x2 p1 = __builtin_convertvector(mvk_f32x2{ra, rb}, mvk_bf16x2);
  const unsigned u1 = __builtin_bit_cast(unsigned, p1);
  const float sa = ra - __uint_as_float(u1 << 16), sb = rb - __uint_as_float(u1 & 0xffff0000u);
  const mvk_bf16x2 p2 = __builtin_convertvector(mvk_f32x2{sa, sb}, mvk_bf16x2);
  const unsigned u2 = __builtin_bit_cast(unsigned, p2);
  const unsigned u[3] = {u0, u1, u2};
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    unsigned short* dst = planes + (long long)p * n + i;
    dst[0] = (unsigned short)(u[p] & 0xffffu);
    if (i + 1 < n) dst[1] = (unsigned short)(u[p] >> 16);
  }
}

__global__ __launch_bounds__(256) void bf3_to_f32_kernel(const unsigned short* __restrict__ planes, long long n,
                                                         float* __restrict__ x) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float a = __uint_as_float((unsigned)planes[i] << 16), b = __uint_as_float((unsigned)planes[n + i] << 16),
              c = __uint_as_float((unsigned)planes[2 * n + i] << 16);
  x[i] = a + (b + c);
}
}  // namespace

extern "C" {

int mvk_f32_to_bf3(const float* x, int64_t n, void* planes, void* stream) {
  if (!x || !planes || n < 0) return MVK_EINVAL;
  if (n == 0) return MVK_OK;
  hipLaunchKernelGGL(f32_to_bf3_kernel, dim3((unsigned)((n / 2 + 256) / 256)), dim3(256), 0, mvk_stream(stream), x,
                     (long long)n, static_cast<unsigned short*>(planes));
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_bf3_to_f32(const void* planes, int64_t n, float* x, void* stream) {
  if (!x || !planes || n < 0) return MVK_EINVAL;
  if (n == 0) return MVK_OK;
  hipLaunchKernelGGL(bf3_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, mvk_stream(stream),
                     static_cast<const unsigned short*>(planes), (long long)n, x);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// ResNet helpers on NHWC tensors: AvgPool2d(3, 2, 1) (count_include_pad), nearest Upsample(2), out = a x + b y
// ---------------------------------------------------------------------------------------------------------
namespace {

// V floats per lane (V = 4: 16-byte accesses when C % 4 == 0 and the tensors are 16-byte aligned); C below is in units of V
template <int V>
struct VecT {
  float v[V];
};
template <int V>
__device__ __forceinline__ VecT<V> vload(const float* p) {
  VecT<V> r;
  if (V == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    r.v[0] = t.x, r.v[1 % V] = t.y, r.v[2 % V] = t.z, r.v[3 % V] = t.w;
  } else {
    r.v[0] = *p;
  }
  return r;
}
template <int V>
__device__ __forceinline__ void vstore(float* p, const VecT<V>& r) {
  if (V == 4)
    *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1 % V], r.v[2 % V], r.v[3 % V]);
  else
    *p = r.v[0];
}

template <int V>
__global__ __launch_bounds__(256) void avgpool3s2_fwd_kernel(const float* __restrict__ x, int n, int H, int W, int C,
                                                             float* __restrict__ y) {
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;  // floor((H + 2 - 3) / 2) + 1
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n * OH * OW * C) return;
  const int c = (int)(i % C);
  long long t = i / C;
  const int ow = (int)(t % OW);
  t /= OW;
  const int oh = (int)(t % OH);
  const long long b = t / OH;
  VecT<V> s;
#pragma unroll
  for (int e = 0; e < V; ++e) s.v[e] = 0.f;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int h = 2 * oh - 1 + kh;
    if (h < 0 || h >= H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int w = 2 * ow - 1 + kw;
      if (w < 0 || w >= W) continue;
      const VecT<V> t4 = vload<V>(x + (((b * H + h) * W + w) * C + c) * V);
#pragma unroll
      for (int e = 0; e < V; ++e) s.v[e] += t4.v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < V; ++e) s.v[e] *= (1.0f / 9.0f);
  vstore<V>(y + i * V, s);
}

// dx[b,h,w,c] = (1/9) sum over the output windows that contain (h,w)
template <int V>
__global__ __launch_bounds__(256) void avgpool3s2_bwd_kernel(const float* __restrict__ dy, int n, int H, int W, int C,
                                                             float* __restrict__ dx) {
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n * H * W * C) return;
  const int c = (int)(i % C);
  long long t = i / C;
  const int w = (int)(t % W);
  t /= W;
  const int h = (int)(t % H);
  const long long b = t / H;
  VecT<V> s;
#pragma unroll
  for (int e = 0; e < V; ++e) s.v[e] = 0.f;
  for (int oh = h / 2; oh <= (h + 1) / 2; ++oh) {  // windows rows 2*oh-1 .. 2*oh+1
    if (oh >= OH) continue;
    for (int ow = w / 2; ow <= (w + 1) / 2; ++ow) {
      if (ow >= OW) continue;
      const VecT<V> t4 = vload<V>(dy + (((b * OH + oh) * OW + ow) * C + c) * V);
#pragma unroll
      for (int e = 0; e < V; ++e) s.v[e] += t4.v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < V; ++e) s.v[e] *= (1.0f / 9.0f);
  vstore<V>(dx + i * V, s);
}

template <int V>
__global__ __launch_bounds__(256) void upsample2_fwd_kernel(const float* __restrict__ x, int n, int H, int W, int C,
                                                            float* __restrict__ y) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n * 4 * H * W * C) return;
  const int c = (int)(i % C);
  long long t = i / C;
  const int ow = (int)(t % (2 * W));
  t /= 2 * W;
  const int oh = (int)(t % (2 * H));
  const long long b = t / (2 * H);
  vstore<V>(y + i * V, vload<V>(x + (((b * H + (oh >> 1)) * W + (ow >> 1)) * C + c) * V));
}

template <int V>
__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const float* __restrict__ dy, int n, int H, int W, int C,
                                                            float* __restrict__ dx) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n * H * W * C) return;
  const int c = (int)(i % C);
  long long t = i / C;
  const int w = (int)(t % W);
  t /= W;
  const int h = (int)(t % H);
  const long long b = t / H;
  const float* p = dy + (((b * 2 * H + 2 * h) * 2 * W + 2 * w) * C + c) * V;
  const long long rowf = (long long)2 * W * C * V;
  const VecT<V> p00 = vload<V>(p), p01 = vload<V>(p + C * V), p10 = vload<V>(p + rowf), p11 = vload<V>(p + rowf + C * V);
  VecT<V> s;
#pragma unroll
  for (int e = 0; e < V; ++e) s.v[e] = (p00.v[e] + p01.v[e]) + (p10.v[e] + p11.v[e]);
  vstore<V>(dx + i * V, s);
}

__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, float a, const float* __restrict__ y,
                                                    float b, long long n, int act, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  out[i] = mvk_act(a * (x ? x[i] : 0.f) + b * (y ? y[i] : 0.f), act);
}

// 16 bytes per lane (n4 = n / 4 quads; x, y, out 16-byte aligned)
__global__ __launch_bounds__(256) void axpby4_kernel(const float4* __restrict__ x, float a, const float4* __restrict__ y,
                                                     float b, long long n4, int act, float4* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 xv = x ? x[i] : z, yv = y ? y[i] : z;
  float4 r;
  r.x = mvk_act(a * xv.x + b * yv.x, act);
  r.y = mvk_act(a * xv.y + b * yv.y, act);
  r.z = mvk_act(a * xv.z + b * yv.z, act);
  r.w = mvk_act(a * xv.w + b * yv.w, act);
  out[i] = r;
}
}  // namespace

extern "C" {

int mvk_avgpool3s2_fwd(const float* x, float* y, int n, int H, int W, int C, void* stream) {
  if (!x || !y || n < 0 || H <= 0 || W <= 0 || C <= 0) return MVK_EINVAL;
  const bool v4 = (C % 4 == 0) && mvk_aligned16(x) && mvk_aligned16(y);
  const int Cq = v4 ? C / 4 : C;
  const long long total = (long long)n * ((H + 1) / 2) * ((W + 1) / 2) * Cq;
  if (total == 0) return MVK_OK;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (v4)
    hipLaunchKernelGGL(avgpool3s2_fwd_kernel<4>, grid, dim3(256), 0, mvk_stream(stream), x, n, H, W, Cq, y);
  else
    hipLaunchKernelGGL(avgpool3s2_fwd_kernel<1>, grid, dim3(256), 0, mvk_stream(stream), x, n, H, W, Cq, y);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_avgpool3s2_bwd(const float* dy, float* dx, int n, int H, int W, int C, void* stream) {
  if (!dy || !dx || n < 0 || H <= 0 || W <= 0 || C <= 0) return MVK_EINVAL;
  const bool v4 = (C % 4 == 0) && mvk_aligned16(dy) && mvk_aligned16(dx);
  const int Cq = v4 ? C / 4 : C;
  const long long total = (long long)n * H * W * Cq;
  if (total == 0) return MVK_OK;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (v4)
    hipLaunchKernelGGL(avgpool3s2_bwd_kernel<4>, grid, dim3(256), 0, mvk_stream(stream), dy, n, H, W, Cq, dx);
  else
    hipLaunchKernelGGL(avgpool3s2_bwd_kernel<1>, grid, dim3(256), 0, mvk_stream(stream), dy, n, H, W, Cq, dx);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_upsample2_fwd(const float* x, float* y, int n, int H, int W, int C, void* stream) {
  if (!x || !y || n < 0 || H <= 0 || W <= 0 || C <= 0) return MVK_EINVAL;
  const bool v4 = (C % 4 == 0) && mvk_aligned16(x) && mvk_aligned16(y);
  const int Cq = v4 ? C / 4 : C;
  const long long total = (long long)n * 4 * H * W * Cq;
  if (total == 0) return MVK_OK;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (v4)
    hipLaunchKernelGGL(upsample2_fwd_kernel<4>, grid, dim3(256), 0, mvk_stream(stream), x, n, H, W, Cq, y);
  else
    hipLaunchKernelGGL(upsample2_fwd_kernel<1>, grid, dim3(256), 0, mvk_stream(stream), x, n, H, W, Cq, y);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_upsample2_bwd(const float* dy, float* dx, int n, int H, int W, int C, void* stream) {
  if (!dy || !dx || n < 0 || H <= 0 || W <= 0 || C <= 0) return MVK_EINVAL;
  const bool v4 = (C % 4 == 0) && mvk_aligned16(dy) && mvk_aligned16(dx);
  const int Cq = v4 ? C / 4 : C;
  const long long total = (long long)n * H * W * Cq;
  if (total == 0) return MVK_OK;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (v4)
    hipLaunchKernelGGL(upsample2_bwd_kernel<4>, grid, dim3(256), 0, mvk_stream(stream), dy, n, H, W, Cq, dx);
  else
    hipLaunchKernelGGL(upsample2_bwd_kernel<1>, grid, dim3(256), 0, mvk_stream(stream), dy, n, H, W, Cq, dx);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// Measured ceiling of the matrix pipes (bench.py `roofline_mfma.measured_*`): a register-only loop of
// v_mfma_f32_32x32x16_bf16, one wave per SIMD, 256 workgroups, 4 accumulator chains, 64 iters MFMAs per chain, with a different
// operand pair of hashed values per MFMA (random = 1) or two constants (random = 0).  The pipe is 100 % busy either way; with
// realistic operand data the chip sustains 1.4-1.9 GHz instead of 2.1-2.4 (power): THAT is the ceiling a real MFMA-bound
// kernel is measured against (profiles/r03_mfma_sustained.txt, tools/ubench/mfma_bf16_sustained.hip).
// ---------------------------------------------------------------------------------------------------------
namespace {
typedef float probe_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 probe_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int probe_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void mfma_probe_kernel(float* out, int iters, int rnd, unsigned seed) {
  probe_f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  probe_bf16x8 av[8], bv[8];
  // (every loop over the operand arrays is unrolled: a run-time index puts them in scratch memory, whose set-up was a fixed
  // ~36 us per launch — the 80-us launches bench.py times read 1.09 PFLOP/s where the loop itself runs at 1.85)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    probe_u32x4 ua, ub;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned h = (seed + threadIdx.x * 2654435761u + j * 40503u + e * 9176u + blockIdx.x * 7919u) * 2246822519u;
      h ^= h >> 15;
      h *= 3266489917u;
      h ^= h >> 13;
      const unsigned lo = 0x3f80u | (h & 0x807fu), hi = 0x3f80u | ((h >> 16) & 0x807fu);  // +-[1, 2)
      ua[e] = rnd ? (lo | (hi << 16)) : 0x3f803f80u;
      ub[e] = rnd ? ((hi ^ 0x8000u) | (lo << 16)) : 0x3f003f00u;
    }
    av[j] = __builtin_bit_cast(probe_bf16x8, ua);
    bv[j] = __builtin_bit_cast(probe_bf16x8, ub);
  }
  if (rnd == 2) {  // the same loop on v_mfma_f32_32x32x16_f16 (the scaled-fp16 kernels' instruction): the hashed bits read as fp16
    typedef _Float16 probe_f16x8 __attribute__((ext_vector_type(8)));
    probe_f16x8 ah[8], bh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      probe_u32x4 ua = __builtin_bit_cast(probe_u32x4, av[j]), ub = __builtin_bit_cast(probe_u32x4, bv[j]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {  // +-[1, 2) with all ten mantissa bits hashed (a low piece has no quiet bits either)
        ua[e] = (ua[e] & 0x83ff83ffu) | 0x3c003c00u | ((ua[e] >> 3) & 0x03800380u);
        ub[e] = (ub[e] & 0x83ff83ffu) | 0x3c003c00u | ((ub[e] >> 3) & 0x03800380u);
      }
      ah[j] = __builtin_bit_cast(probe_f16x8, ua);
      bh[j] = __builtin_bit_cast(probe_f16x8, ub);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[(u + i) & 7], bh[(u * 3 + i) & 7], acc[i], 0, 0, 0);
    }
  } else
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[(u + i) & 7], bv[(u * 3 + i) & 7], acc[i], 0, 0, 0);
  }
  float s_ = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s_ += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s_;
}
}  // namespace

// out: 65536 floats (written, meaningless); one launch = 256 workgroups x 4 waves x iters x 64 MFMAs of 32768 FLOP
// random_operands: 0 = two constants, 1 = hashed bf16 values, 2 = hashed fp16 values on v_mfma_f32_32x32x16_f16
extern "C" int mvk_probe_mfma_bf16(float* out, int iters, int random_operands, void* stream) {
  if (!out || iters <= 0) return MVK_EINVAL;
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(256), dim3(256), 0, mvk_stream(stream), out, iters, random_operands, 12345u);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

// The measured HBM denominator of bench.py's roofline ("also report against a measured device-copy
// bandwidth"): a float4 streaming copy — 16 bytes per lane, four independent nontemporal loads in flight per thread before the
// first store, a grid of 8 workgroups per CU walking the buffer in grid-sized strides (every wave's four loads are four
// consecutive 1-KB lines of one 4-KB run).  MI355X_MICROARCH.md quotes 6.29 TB/s (read + write bytes) for this shape of
// kernel; torch's copy_ — the probe of rounds 1-4 — reached 4.7-5.5.
namespace {
__global__ __launch_bounds__(256) void stream_copy_kernel(const mvk::f32x4* __restrict__ src, mvk::f32x4* __restrict__ dst, long long n4) {
  const long long stride = (long long)gridDim.x * 1024;  // float4 per grid sweep (4 per thread)
  for (long long base = (long long)blockIdx.x * 1024 + threadIdx.x; base < n4; base += stride) {
    mvk::f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (base + u * 256 < n4) v[u] = __builtin_nontemporal_load(src + base + u * 256);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (base + u * 256 < n4) __builtin_nontemporal_store(v[u], dst + base + u * 256);
  }
}
}  // namespace

// dst, src: 16-byte aligned, n a multiple of 4 floats.  One launch copies the whole buffer.
extern "C" int mvk_probe_stream_copy(float* dst, const float* src, int64_t n, void* stream) {
  if (!dst || !src || n <= 0 || (n & 3) || !mvk_aligned16(dst) || !mvk_aligned16(src)) return MVK_EINVAL;
  const long long n4 = n / 4;
  const int grid = (int)std::min<long long>((n4 + 1023) / 1024, 2048);
  hipLaunchKernelGGL(stream_copy_kernel, dim3(grid), dim3(256), 0, mvk_stream(stream), reinterpret_cast<const mvk::f32x4*>(src),
                     reinterpret_cast<mvk::f32x4*>(dst), n4);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Up to MVK_COPY_MAX device-to-device copies in ONE launch: the modalities of a batch into the captured step's input buffers
// (trainers/graph.py; the reference's DataLoader hands the step one collated dict per batch, trainers/base/base_trainer.py:682-700).
// Two hipMemcpyAsync blits of 1.6 + 6.3 MB were 6 + 6 us back to back in front of every replay.
// ---------------------------------------------------------------------------------------------------------
namespace {
struct CopyTable {
  int n;
  unsigned tile0[MVK_COPY_MAX + 1];  // first 16-KB tile of copy i (tile0[n] = all tiles)
  mvk_copy_desc d[MVK_COPY_MAX];
};
__global__ __launch_bounds__(256) void copy_batch_kernel(const CopyTable T) {
  for (unsigned t = blockIdx.x; t < T.tile0[T.n]; t += gridDim.x) {
    int i = 0;
    for (int j = 1; j < T.n; ++j)
      if (t >= T.tile0[j]) i = j;
    const long long n4 = T.d[i].bytes / 16, base = (long long)(t - T.tile0[i]) * 1024 + threadIdx.x;
    const mvk::f32x4* src = reinterpret_cast<const mvk::f32x4*>(T.d[i].src);
    mvk::f32x4* dst = reinterpret_cast<mvk::f32x4*>(T.d[i].dst);
    mvk::f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (base + 256 * u < n4) v[u] = src[base + 256 * u];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (base + 256 * u < n4) dst[base + 256 * u] = v[u];
  }
}
}  // namespace

extern "C" int mvk_copy_batch(const mvk_copy_desc* descs, int n, void* stream) {
  if (n == 0) return MVK_OK;
  if (!descs || n < 0 || n > MVK_COPY_MAX) return MVK_EINVAL;
  CopyTable T{};
  unsigned tiles = 0;
  for (int i = 0; i < n; ++i) {
    const mvk_copy_desc& d = descs[i];
    if (d.bytes == 0) continue;  // an empty tensor (torch hands out NULL for it)
    if (!d.dst || !d.src || d.bytes < 0 || (d.bytes & 15) || !mvk_aligned16(d.dst) || !mvk_aligned16(d.src)) return MVK_EINVAL;
    T.tile0[T.n] = tiles;
    T.d[T.n++] = d;
    tiles += (unsigned)((d.bytes + 16383) / 16384);
  }
  T.tile0[T.n] = tiles;
  if (tiles == 0) return MVK_OK;
  hipLaunchKernelGGL(copy_batch_kernel, dim3(tiles < 2048 ? tiles : 2048), dim3(256), 0, mvk_stream(stream), T);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

// ---------------------------------------------------------------------------------------------------------
// max |x| of a tensor: the operand scale of the scaled-fp16 kernels (csrc/bf3.hpp) for tensors whose producer does not
// publish it.  *out must hold 0 (or a lower bound that is to be kept) before the launch.
// ---------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long long n, float* out) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  const long long n4 = n / 4, stride = (long long)gridDim.x * 256;
  float m = 0.f;
  // a workgroup walks 16-KB tiles: four independent, neighbouring 16-byte loads in flight per thread
  const long long tiles = n4 / 1024;
  for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const v4* p = reinterpret_cast<const v4*>(x) + t * 1024 + threadIdx.x;
    const v4 a = p[0], b = p[256], c = p[512], d = p[768];
    const float ma = fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3])));
    const float mb = fmaxf(fmaxf(fabsf(b[0]), fabsf(b[1])), fmaxf(fabsf(b[2]), fabsf(b[3])));
    const float mc = fmaxf(fmaxf(fabsf(c[0]), fabsf(c[1])), fmaxf(fabsf(c[2]), fabsf(c[3])));
    const float md = fmaxf(fmaxf(fabsf(d[0]), fabsf(d[1])), fmaxf(fabsf(d[2]), fabsf(d[3])));
    m = fmaxf(m, fmaxf(fmaxf(ma, mb), fmaxf(mc, md)));
  }
  for (long long i = tiles * 1024 + (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const v4 v = reinterpret_cast<const v4*>(x)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) m = fmaxf(m, fabsf(x[4 * n4 + threadIdx.x]));
  __shared__ float red[4];
  mvk::amax_publish(m, out, red);
}
}  // namespace

extern "C" int mvk_amax(const float* x, int64_t n, float* out, void* stream) {
  if (!x || !out || n < 0 || !mvk_aligned16(x)) return MVK_EINVAL;
  if (n == 0) return MVK_OK;
  long long blocks = (n / 4 + 1023) / 1024;
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  hipLaunchKernelGGL(amax_kernel, dim3((unsigned)blocks), dim3(256), 0, mvk_stream(stream), x, (long long)n, out);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Batched transpose with a fused activation / activation derivative: the flatten of the ResNet encoders
// (`out.view(batch, nf0 * s0 * s0)` on an NCHW tensor, models/nn/cub.py:190-195, mmnist.py:300-306) and the `view(-1, nf0,
// s0, s0)` of the decoders (cub.py:232-240), whose convolutional stacks run NHWC here:
//   y[b][c][r] = act(x[b][r][c]) * dact'(msrc[b][r][c])      x, msrc: [batch][rows][cols], y: [batch][cols][rows]
// One pass (32 x 32 tiles through LDS, 128-byte rows on both sides) instead of an activation pass + a strided copy.
// ---------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void transpose_act_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int cols,
                                                            int act, const float* __restrict__ msrc, int dact) {
  __shared__ float tile[32][33];
  const long long base = (long long)blockIdx.z * rows * cols;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    float v = 0.f;
    if (r < rows && c < cols) {
      v = mvk_act(x[base + (long long)r * cols + c], act);
      if (msrc) v *= mvk_act_grad_from_out(msrc[base + (long long)r * cols + c], dact);
    }
    tile[ty + 8 * i][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (r < rows && c < cols) y[base + (long long)c * rows + r] = tile[tx][ty + 8 * i];
  }
}
}  // namespace

extern "C" int mvk_transpose_act(const float* x, float* y, int batch, int rows, int cols, int act, const float* msrc, int dact,
                                 void* stream) {
  if (!x || !y || batch < 0 || rows <= 0 || cols <= 0) return MVK_EINVAL;
  if (batch == 0) return MVK_OK;
  // gridDim.z <= 65535: larger batches (the likelihood evaluators decode 65 536 rows per chunk) go in slices
  const long long plane = (long long)rows * cols;
  for (int b0 = 0; b0 < batch; b0 += 65535) {
    const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
    hipLaunchKernelGGL(transpose_act_kernel, dim3((cols + 31) / 32, (rows + 31) / 32, nb), dim3(256), 0, mvk_stream(stream),
                       x + b0 * plane, y + b0 * plane, rows, cols, act, msrc ? msrc + b0 * plane : nullptr, dact);
    MVK_CHECK_LAUNCH();
  }
  return MVK_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Noise with the generator state in DEVICE memory: a hipGraph replay of a training step draws fresh noise without the
// two host-issued fill launches per replay that torch's graph-safe generator needs (seed / offset tensors).
// Philox4x32-10 (counter = offset + thread index, key = seed), 4 values per thread; normal: Box-Muller.
// state[0] = seed, state[1] = offset (in threads), state[2] = arrival ticket.  Every workgroup reads the offset, then takes
// a ticket; the last one to arrive advances the offset by the launch's thread count and resets the ticket — the values
// depend only on (seed, offset, index): deterministic, and consecutive launches on a stream never overlap.
// ---------------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
  const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
  const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
  c[0] = n0, c[1] = n1, c[2] = n2, c[3] = n3;
}

__global__ __launch_bounds__(256) void device_rng_kernel(float* __restrict__ out, long long n, unsigned long long* __restrict__ st,
                                                         int uniform, float lo, float hi) {
  const unsigned long long seed = st[0], off = st[1];
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const unsigned long long ctr = off + (unsigned long long)idx;
  unsigned c[4] = {(unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u};
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  float v[4];
  if (uniform) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = lo + (float)(c[e] >> 8) * (1.0f / 16777216.0f) * (hi - lo);  // [lo, hi)
  } else {
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
      const float u1 = (float)((c[e] >> 8) + 1u) * (1.0f / 16777216.0f);  // (0, 1]
      const float u2 = (float)(c[e + 1] >> 8) * (1.0f / 16777216.0f);     // [0, 1)
      const float r = sqrtf(-2.0f * logf(u1));
      float sn, cs;
      sincosf(6.28318530717958647692f * u2, &sn, &cs);
      v[e] = r * cs;
      v[e + 1] = r * sn;
    }
  }
  const long long o = idx * 4;
  if (o + 3 < n && mvk_dev_aligned16(out)) {
    *reinterpret_cast<float4*>(out + o) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (o + e < n) out[o + e] = v[e];
  }
  __syncthreads();  // every thread of this workgroup has read the offset
  if (threadIdx.x == 0) {
    __threadfence();
    unsigned* ticket = reinterpret_cast<unsigned*>(st + 2);
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
      st[1] = off + (unsigned long long)gridDim.x * 256ull;
      *ticket = 0u;
      __threadfence();
    }
  }
}
}  // namespace

extern "C" {

int mvk_device_rng(float* out, int64_t n, uint64_t* state, int uniform, float lo, float hi, void* stream) {
  if (!out || !state || n < 0) return MVK_EINVAL;
  if (n == 0) return MVK_OK;
  const long long threads = (n + 3) / 4;
  hipLaunchKernelGGL(device_rng_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, mvk_stream(stream), out,
                     (long long)n, reinterpret_cast<unsigned long long*>(state), uniform, lo, hi);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int mvk_axpby(const float* x, float a, const float* y, float b, int64_t n, int act, float* out, void* stream) {
  if (!out || n < 0 || (!x && !y)) return MVK_EINVAL;
  if (n == 0) return MVK_OK;
  if (n % 4 == 0 && mvk_aligned16(out) && (!x || mvk_aligned16(x)) && (!y || mvk_aligned16(y)))
    hipLaunchKernelGGL(axpby4_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, mvk_stream(stream),
                       reinterpret_cast<const float4*>(x), a, reinterpret_cast<const float4*>(y), b, (long long)(n / 4), act,
                       reinterpret_cast<float4*>(out));
  else
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, mvk_stream(stream), x, a, y, b,
                       (long long)n, act, out);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

}  // extern "C"

