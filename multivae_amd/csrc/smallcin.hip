// The image-consuming convolution: Conv2d(C <= 4, Cv, 4, 2, 1) on the NCHW network input (svhn.py:13-15), forward and
// backward-weight.  With K = 16 C <= 64 the implicit GEMM spends its time in scalar NCHW gathers (46 + 62 us at
// B = 512 for 0.4 GFLOP each); here every thread owns one output position, keeps its 16 C window values in registers
// and walks the output channels with the weight (forward) / the output gradient (backward) broadcast from LDS.
#include "common.hpp"

namespace mvk {

// igemm.hip: dWref[Cv][Cu][taps] += sum_z slab[z][(tap*Cu + cu)][cv] (ordered reduce into the reference weight layout)
int convref_reduce(const float* slab, int nz, int Cu, int Cv, int taps, float* dWref, hipStream_t s, bool deferred = false);

constexpr int SC_MAXC = 4;

template <int C>
__device__ __forceinline__ void load_window(const float* __restrict__ U, long long img, int H, int W, int i, int j,
                                            float (&x)[16 * C]) {
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int kh = 0; kh < 4; ++kh) {
      const int hh = 2 * i - 1 + kh;
      const float* row = U + ((img * C + c) * H + hh) * (long long)W;
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        const int ww = 2 * j - 1 + kw;
        const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
        x[(kh * 4 + kw) * C + c] = ok ? row[ww] : 0.f;  // k = tap * C + c, the order of the packed weight rows
      }
    }
}

// V[n,h,w,Cv] = act(conv(U[n,C,2h,2w]) + b);  Wdown[(tap*C + c)][cv].  One thread = one position, all output channels
// (measured: splitting the channels over 4 threads per position is slower: 52 vs 42 us at B = 512).
template <int C>
__global__ __launch_bounds__(256) void smallcin_fwd_kernel(const float* __restrict__ U, const float* __restrict__ Wdown,
                                                           const float* __restrict__ bias, float* __restrict__ V,
                                                           long long npos, int h, int w, int Cv, int act) {
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [16*C][Cv] + [Cv]
  const int K = 16 * C;
  for (int i = threadIdx.x; i < K * Cv; i += 256) sw[i] = Wdown[i];
  for (int i = threadIdx.x; i < Cv; i += 256) sw[K * Cv + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const long long pos = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pos >= npos) return;
  const int j = (int)(pos % w);
  const long long t = pos / w;
  const int i = (int)(t % h);
  const long long img = t / h;
  float x[16 * C];
  load_window<C>(U, img, 2 * h, 2 * w, i, j, x);
  float* out = V + pos * Cv;
  for (int c0 = 0; c0 < Cv; c0 += 4) {
    float a0 = sw[K * Cv + c0], a1 = sw[K * Cv + c0 + 1], a2 = sw[K * Cv + c0 + 2], a3 = sw[K * Cv + c0 + 3];
#pragma unroll
    for (int k = 0; k < 16 * C; ++k) {
      const float4 wv = *reinterpret_cast<const float4*>(sw + k * Cv + c0);  // same address in every lane: broadcast
      a0 = fmaf(x[k], wv.x, a0);
      a1 = fmaf(x[k], wv.y, a1);
      a2 = fmaf(x[k], wv.z, a2);
      a3 = fmaf(x[k], wv.w, a3);
    }
    *reinterpret_cast<float4*>(out + c0) = make_float4(mvk_act(a0, act), mvk_act(a1, act), mvk_act(a2, act), mvk_act(a3, act));
  }
}

// slab[block][(tap*C + c)][cv] = sum over the block's positions of window[k] * dV[pos][cv].
// The four waves split the staged positions; a lane owns a (2 C) x (Cv / 8) block of the gradient: 2 C + Cv / 8 floats read from
// LDS (vector reads, 8-byte aligned) feed 2 C Cv / 8 FMAs per position, and the waves' partial blocks are added in a fixed order at
// the end.  (Until round 5 a thread owned C x Cv / 16 entries and all 256 threads walked all positions: 5 scalar LDS reads per 6
// FMAs at Cu = 3, Cv = 32 — the launch, the last compute kernel of the headline step's critical chain, was bound by the LDS
// instruction rate: 26 us for 0.2 GFLOP.)
template <int C, int NCV>
__global__ __launch_bounds__(256) void smallcin_wgrad_kernel(const float* __restrict__ U, const float* __restrict__ dV,
                                                             float* __restrict__ slab, long long npos, int h, int w,
                                                             int pos_per_block) {
  constexpr int K = 16 * C, Cv = 16 * NCV;
  constexpr int PT = 64;                       // positions staged at a time
  constexpr int XS = K + 4;                    // row stride of the window matrix: 16-byte aligned rows
  constexpr int KQ = 2 * C, CQ = 2 * NCV;      // a lane's block: KQ window entries x CQ channels
  constexpr int STAGE = PT * XS + PT * Cv, RED = 3 * K * Cv;
  __shared__ __attribute__((aligned(16))) float smem[STAGE > RED ? STAGE : RED];
  float* const xs = smem;
  float* const ds = smem + PT * XS;
  const long long p0 = (long long)blockIdx.x * pos_per_block;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kq = lane >> 3, cq = lane & 7;
  float acc[KQ][CQ];
#pragma unroll
  for (int a = 0; a < KQ; ++a)
#pragma unroll
    for (int b = 0; b < CQ; ++b) acc[a][b] = 0.f;
  // Staging is a software pipeline: the global loads of tile t + 1 (a 4 x C window row per thread, PT Cv / 1024 gradient quads)
  // are in flight while tile t is multiplied — four dependent rounds of "load, barrier, multiply, barrier" per workgroup were the
  // launch's time (25 us for 0.2 GFLOP at the headline's encoder batch).
  constexpr int ND = PT * Cv / 4 / 256;
  static_assert(PT * Cv % 1024 == 0, "gradient quads divide over the workgroup");
  float xr[4 * C];
  float4 dr[ND];
  const int pl = threadIdx.x >> 2, kh = threadIdx.x & 3;
  auto fetch = [&](int s0) {
    const long long pos = p0 + s0 + pl;
    const bool pv = pos < npos && (s0 + pl) < pos_per_block;
    const int j = pv ? (int)(pos % w) : 0;
    const long long t = pv ? pos / w : 0;
    const int i = (int)(t % h);
    const long long img = t / h;
    const int H = 2 * h, W = 2 * w;
    const int hh = 2 * i - 1 + kh;
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        const int ww = 2 * j - 1 + kw;
        const bool ok = pv && hh >= 0 && hh < H && ww >= 0 && ww < W;
        xr[kw * C + c] = ok ? U[((img * C + c) * H + hh) * (long long)W + ww] : 0.f;
      }
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int e = threadIdx.x + u * 256, pp = e / (Cv / 4);
      const long long pos2 = p0 + s0 + pp;
      dr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pos2 < npos && (s0 + pp) < pos_per_block)
        dr[u] = *reinterpret_cast<const float4*>(dV + pos2 * Cv + (e - pp * (Cv / 4)) * 4);
    }
  };
  fetch(0);
  for (int s0 = 0; s0 < pos_per_block; s0 += PT) {
    __syncthreads();  // the previous tile's readers are done
#pragma unroll
    for (int q = 0; q < 4 * C; ++q) xs[pl * XS + kh * 4 * C + q] = xr[q];
#pragma unroll
    for (int u = 0; u < ND; ++u) *reinterpret_cast<float4*>(ds + (threadIdx.x + u * 256) * 4) = dr[u];
    __syncthreads();
    if (s0 + PT < pos_per_block) fetch(s0 + PT);
#pragma unroll 4
    for (int pp = wave; pp < PT; pp += 4) {
      float xv[KQ], dv[CQ];
      typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int a = 0; a < C; ++a) {
        const f2 t = *reinterpret_cast<const f2*>(xs + pp * XS + kq * KQ + 2 * a);
        xv[2 * a] = t[0], xv[2 * a + 1] = t[1];
      }
#pragma unroll
      for (int b = 0; b < NCV; ++b) {
        const f2 t = *reinterpret_cast<const f2*>(ds + pp * Cv + cq * CQ + 2 * b);
        dv[2 * b] = t[0], dv[2 * b + 1] = t[1];
      }
#pragma unroll
      for (int a = 0; a < KQ; ++a)
#pragma unroll
        for (int b = 0; b < CQ; ++b) acc[a][b] = fmaf(xv[a], dv[b], acc[a][b]);
    }
  }
  // the four waves' partial blocks in a fixed order: waves 1..3 through LDS, wave 0 adds and stores
  __syncthreads();
  if (wave > 0) {
#pragma unroll
    for (int a = 0; a < KQ; ++a)
#pragma unroll
      for (int b = 0; b < CQ; ++b) smem[(wave - 1) * (K * Cv) + (kq * KQ + a) * Cv + cq * CQ + b] = acc[a][b];
  }
  __syncthreads();
  if (wave == 0) {
    float* out = slab + (long long)blockIdx.x * (K * Cv);
#pragma unroll
    for (int a = 0; a < KQ; ++a)
#pragma unroll
      for (int b = 0; b < CQ; ++b) {
        const int i = (kq * KQ + a) * Cv + cq * CQ + b;
        out[i] = ((acc[a][b] + smem[i]) + smem[K * Cv + i]) + smem[2 * K * Cv + i];
      }
  }
}

bool smallcin_supported(int Cu, int Cv) { return Cu >= 1 && Cu <= SC_MAXC && (Cv == 16 || Cv == 32 || Cv == 64); }

int smallcin_fwd(const float* U, const float* Wdown, const float* bias, float* V, int n, int h, int w, int Cu, int Cv,
                 int act, hipStream_t s) {
  const long long npos = (long long)n * h * w;
  if (npos == 0) return MVK_OK;
  const size_t lds = (size_t)(16 * Cu * Cv + Cv) * sizeof(float);
  const dim3 grid((unsigned)((npos + 255) / 256));
  switch (Cu) {
    case 1: hipLaunchKernelGGL(smallcin_fwd_kernel<1>, grid, dim3(256), lds, s, U, Wdown, bias, V, npos, h, w, Cv, act); break;
    case 2: hipLaunchKernelGGL(smallcin_fwd_kernel<2>, grid, dim3(256), lds, s, U, Wdown, bias, V, npos, h, w, Cv, act); break;
    case 3: hipLaunchKernelGGL(smallcin_fwd_kernel<3>, grid, dim3(256), lds, s, U, Wdown, bias, V, npos, h, w, Cv, act); break;
    default: hipLaunchKernelGGL(smallcin_fwd_kernel<4>, grid, dim3(256), lds, s, U, Wdown, bias, V, npos, h, w, Cv, act); break;
  }
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

// dWref[Cv][Cu][4][4] += ...; needs blocks * 16*Cu*Cv floats of scratch; returns 1 when the scratch is too small
int smallcin_wgrad(const float* U, const float* dV, float* dWref, int n, int h, int w, int Cu, int Cv, float* ws,
                   long long ws_floats, hipStream_t s) {
  const long long npos = (long long)n * h * w;
  if (npos == 0) return MVK_OK;
  const int nout = 16 * Cu * Cv;
  int blocks = (int)((npos + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  float* dslab = defer_scratch(dWref, (long long)blocks * nout, s);
  if (dslab) ws = dslab;
  else if (!ws || (long long)blocks * nout > ws_floats) return 1;
  int ppb = (int)((npos + blocks - 1) / blocks);
  ppb = (ppb + 63) / 64 * 64;
  blocks = (int)((npos + ppb - 1) / ppb);
#define MVK_SCW(CC, NN) hipLaunchKernelGGL((smallcin_wgrad_kernel<CC, NN>), dim3(blocks), dim3(256), 0, s, U, dV, ws, npos, h, w, ppb)
#define MVK_SCW_C(CC)              \
  switch (Cv / 16) {               \
    case 1: MVK_SCW(CC, 1); break; \
    case 2: MVK_SCW(CC, 2); break; \
    default: MVK_SCW(CC, 4); break; \
  }
  switch (Cu) {
    case 1: MVK_SCW_C(1) break;
    case 2: MVK_SCW_C(2) break;
    case 3: MVK_SCW_C(3) break;
    default: MVK_SCW_C(4) break;
  }
#undef MVK_SCW_C
#undef MVK_SCW
  MVK_CHECK_LAUNCH();
  return convref_reduce(ws, blocks, Cu, Cv, 16, dWref, s, dslab != nullptr);
}

}  // namespace mvk
