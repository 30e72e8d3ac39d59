// 3x3 / stride 1 / pad 1 convolutions whose input OR output is an image (<= 4 channels): the first layer of the ResNet
// encoders and the last layer of the ResNet decoders (reference: models/nn/mmnist.py:254-366 `conv_img`, models/nn/cub.py:
// 144-246 `conv_img`), forward, backward-data and backward-weight.  On the tiled GEMM engine these shapes have K = 27 or
// N = 3: no 16-channel k-tiles, no 16-byte operand loads — they ran on the generic scalar-gather kernel (430 us per launch
// at the JMVAE 64x64 batch, 23 % of that step).  They are streaming problems (one pass over the 64-channel tensor) with
// 27 FMAs per output element, so plain fp32 FMA chains on the vector ALU are enough:
//   * smallcin:  Y[n,H,W,Cout] = act(conv(X[n,H,W,Cs<=4]) + b) (* act'(mask)): a workgroup takes a band of 8 image rows, the
//     3-channel tile with halo sits in LDS (broadcast reads), a thread keeps W[:, 4 couts] in registers (27 float4) and
//     writes 16 bytes per position.  Also the backward-data pass of the image-producing layer (same shape, flipped pack).
//   * smallcout: Y[n,H,W,Cs<=4] = act(conv(X[n,H,W,Cin]) + b): the Cin-channel tile with halo and the weights in LDS, two
//     threads per position split the taps and combine through a wave shuffle.
//   * wgrad:     dW[(tap, cs)][cb] = sum_pos S[pos + tap][cs] B[pos][cb] for either role of the small tensor; per-workgroup
//     slabs, ordered (deferred) finish into the reference layout.
#include <cstdlib>

#include "bf3.hpp"  // amax_publish (the amax protocol), f32x4

namespace {

using mvk::amax_publish;
using mvk::f32x4;

struct C3Args {
  const float* X;
  const float* Wp;   // [(tap * Cin + ci)][Cout]
  const float* bias;
  float* Y;
  const float* mask_src;  // optional tensor of Y's shape: the result is multiplied by mask_act'(mask_src)
  int n, H, W, Cin, Cout, act, mask_act;
  float* y_amax;          // smallcin only, optional: receives max |Y| (atomic max; amax protocol)
};

constexpr int C3_RB = 8;  // image rows per workgroup

// ---- input image -> many channels ----------------------------------------------------------------------------------------
template <int CS>
__global__ __launch_bounds__(256) void conv3_smallcin_kernel(const C3Args g) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [(RB + 2)][(W + 2)][CS], zero halo
  constexpr int KT = 9 * CS;
  const int W = g.W, H = g.H, Cout = g.Cout;
  const int TW = (W + 2) * CS;
  const int bands = (H + C3_RB - 1) / C3_RB;
  const int img = blockIdx.x / bands, y0 = (blockIdx.x % bands) * C3_RB;
  const int rows = min(C3_RB, H - y0);
  const int CG = Cout / 4, PL = 256 / CG;
  const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
  const bool active = pl < PL;
  // weights of this thread's 4 output channels
  f32x4 w[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) w[k] = *reinterpret_cast<const f32x4*>(g.Wp + (long long)k * Cout + 4 * cg);
  f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
  if (g.bias) b4 = *reinterpret_cast<const f32x4*>(g.bias + 4 * cg);
  // stage the tile (rows y0-1 .. y0+rows, columns -1 .. W), zero outside the image
  const float* ximg = g.X + (long long)img * H * W * CS;
  const int tile = (rows + 2) * TW;
  for (int i = threadIdx.x; i < tile; i += 256) {
    const int r = i / TW, rem = i - r * TW;
    const int xcol = rem / CS - 1, c = rem % CS, yy = y0 + r - 1;
    xs[i] = (yy >= 0 && yy < H && xcol >= 0 && xcol < W) ? ximg[((long long)yy * W + xcol) * CS + c] : 0.f;
  }
  __syncthreads();
  if (!active && !g.y_amax) return;
  const int npos = active ? rows * W : 0;
  float* yimg = g.Y + ((long long)img * H + y0) * W * Cout;
  const float* mimg = g.mask_src ? g.mask_src + ((long long)img * H + y0) * W * Cout : nullptr;
  float ymax = 0.f;
  for (int p = pl; p < npos; p += PL) {
    const int r = p / W, x = p - r * W;
    f32x4 acc = b4;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const float* row = xs + (r + dy) * TW + x * CS;  // columns x-1 .. x+1 of tile row r+dy: 3 * CS contiguous floats
#pragma unroll
      for (int q = 0; q < 3 * CS; ++q) {
        const float v = row[q];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(v, w[dy * 3 * CS + q][j], acc[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = mvk_act(acc[j], g.act);
    if (mimg) {
      const f32x4 ms = *reinterpret_cast<const f32x4*>(mimg + (long long)p * Cout + 4 * cg);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] *= mvk_act_grad_from_out(ms[j], g.mask_act);
    }
    *reinterpret_cast<f32x4*>(yimg + (long long)p * Cout + 4 * cg) = acc;
    ymax = fmaxf(fmaxf(ymax, fmaxf(fabsf(acc[0]), fabsf(acc[1]))), fmaxf(fabsf(acc[2]), fabsf(acc[3])));
  }
  if (g.y_amax) {  // every thread of the workgroup (amax_publish synchronises); the staging tile is dead: its head is the scratch
    __syncthreads();
    amax_publish(ymax, g.y_amax, xs);
  }
}


// ---- many channels -> output image ----------------------------------------------------------------------------------------
// Tile = 8 rows x 16 columns of output positions; the Cin-channel input tile with halo and the weights sit in LDS; a wave owns
// two tile rows (16 positions each) as MFMA row tiles: D[pos][co] += X[pos + tap][c] W[(tap, c)][co] on
// v_mfma_f32_16x16x4_f32 (exact fp32; only Cs <= 4 of the 16 columns are used — the layer is still bound by streaming X
// once).  Within a 16-channel block the k index is permuted so that a lane reads its four channels with one 16-byte load.
constexpr int C3_TR = 8, C3_TC = 16;

template <int CS>
__global__ __launch_bounds__(256) void conv3_smallcout_kernel(const C3Args g) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Cin = g.Cin, H = g.H, W = g.W;
  const int PS = Cin + 4;                        // floats per tile pixel (16-byte shifted rows)
  const int TWp = (C3_TC + 2) * PS;              // floats per tile row
  float* xt = sm;                                // [(TR + 2)][(TC + 2)][PS]
  float* wt = sm + (C3_TR + 2) * TWp;            // [9 * Cin][4]  (columns >= CS zero)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  const int tcols = (W + C3_TC - 1) / C3_TC, trows = (H + C3_TR - 1) / C3_TR;
  const int img = blockIdx.x / (tcols * trows), trem = blockIdx.x % (tcols * trows);
  const int y0 = (trem / tcols) * C3_TR, x0 = (trem % tcols) * C3_TC;
  for (int i = tid; i < 9 * Cin * 4; i += 256) {
    const int k = i >> 2, co = i & 3;
    wt[i] = co < CS ? g.Wp[(long long)k * CS + co] : 0.f;
  }
  const float* ximg = g.X + (long long)img * H * W * Cin;
  const int c4n = Cin / 4;
  for (int i = tid; i < (C3_TR + 2) * (C3_TC + 2) * c4n; i += 256) {
    const int pix = i / c4n, q = i - pix * c4n;
    const int r = pix / (C3_TC + 2), cx = pix - r * (C3_TC + 2);
    const int yy = y0 + r - 1, xx = x0 + cx - 1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = *reinterpret_cast<const f32x4*>(ximg + ((long long)yy * W + xx) * Cin + 4 * q);
    *reinterpret_cast<f32x4*>(xt + r * TWp + cx * PS + 4 * q) = v;
  }
  __syncthreads();
  const float bias = (g.bias && l15 < CS) ? g.bias[l15] : 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int r = wave * 2 + mt;  // tile row of this MFMA tile: positions (y0 + r, x0 + 0..15)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - dy * 3;
      const float* arow = xt + (r + dy) * TWp + (l15 + dx) * PS + 4 * lq;  // A: position l15, channels cb*16 + 4 lq .. +3
      const float* brow = wt + (tap * Cin + 4 * lq) * 4 + (l15 & 3);       // B: k = cb*16 + 4 lq + t, column l15 (& 3)
      for (int cb = 0; cb < Cin; cb += 16) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + cb);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[t], brow[(cb + t) * 4], acc, 0, 0, 0);
      }
    }
    // D[i = 4 lq + rr][j = l15]: position x0 + 4 lq + rr of row y0 + r, output channel l15
    const int yy = y0 + r;
    if (l15 < CS && yy < H) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int xx = x0 + 4 * lq + rr;
        if (xx < W) {
          const long long o = (((long long)img * H + yy) * W + xx) * CS + l15;
          float v = mvk_act(acc[rr] + bias, g.act);
          if (g.mask_src) v *= mvk_act_grad_from_out(g.mask_src[o], g.mask_act);
          g.Y[o] = v;
        }
      }
    }
  }
}


// ---- many channels -> output image, on the vector ALU (end of round 3) ------------------------------------------------------
// The MFMA version above uses 3 of the 16 columns of every v_mfma_f32_16x16x4_f32 and reads one B value from LDS per
// instruction: 1101 us per launch at cfg4's decoder batch (1600 images of 28x28, 321 MB of input whose HBM time is 64 us), 164 us
// at cfg5's.  Here a thread owns ONE output position and its CS <= 4 channels: 9 Cin CS FMAs in exact fp32, the input tile
// (8 x 32 positions + halo) staged 16 channels at a time as [pixel][16 + 4 pad floats] (80-byte stride: conflict-free 16-byte
// reads), the chunk's weights [tap][co][16] read as LDS broadcasts.  27 KB of LDS and ~40 registers: 5 workgroups per CU hide
// the staging of one behind the FMAs of the others.
constexpr int C3V_TR = 8, C3V_TC = 32, C3V_PS = 20;

template <int CS>
__global__ __launch_bounds__(256) void conv3_smallcout_v_kernel(const C3Args g) {
  __shared__ __attribute__((aligned(16))) float xt[(C3V_TR + 2) * (C3V_TC + 2) * C3V_PS];
  const int Cin = g.Cin, H = g.H, W = g.W;
  const int tid = threadIdx.x, r = tid >> 5, c = tid & 31;
  const int tcols = (W + C3V_TC - 1) / C3V_TC, trows = (H + C3V_TR - 1) / C3V_TR;
  const int img = blockIdx.x / (tcols * trows), trem = blockIdx.x % (tcols * trows);
  const int y0 = (trem / tcols) * C3V_TR, x0 = (trem % tcols) * C3V_TC;
  const float* ximg = g.X + (long long)img * H * W * Cin;
  float acc[CS];
#pragma unroll
  for (int co = 0; co < CS; ++co) acc[co] = 0.f;
  for (int cb = 0; cb < Cin; cb += 16) {
    if (cb) __syncthreads();  // the previous chunk's reads are done
    for (int i = tid; i < (C3V_TR + 2) * (C3V_TC + 2) * 4; i += 256) {
      const int pix = i >> 2, q = i & 3;
      const int pr = pix / (C3V_TC + 2), px = pix - pr * (C3V_TC + 2);
      const int yy = y0 + pr - 1, xx = x0 + px - 1;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = *reinterpret_cast<const f32x4*>(ximg + ((long long)yy * W + xx) * Cin + cb + 4 * q);
      *reinterpret_cast<f32x4*>(xt + pix * C3V_PS + 4 * q) = v;
    }
    __syncthreads();
    // The chunk's weights are the same for every lane: SCALAR loads (16 CS consecutive floats of the pack per tap, a uniform
    // address) feeding the FMAs as scalar operands.  As LDS broadcasts (round 3) they were 12 of the 16 ds_read_b128 per tap:
    // 128 LDS clocks per wave and tap beside 192 clocks of FMAs, four SIMDs on one LDS — the kernel ran at a third of its FMA
    // time.  Same sums in the same order.
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - dy * 3;
      const float* xp = xt + ((r + dy) * (C3V_TC + 2) + c + dx) * C3V_PS;
      const float* __restrict__ wp = g.Wp + (long long)(tap * Cin + cb) * CS;  // [16 channels][CS]
      f32x4 x[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = *reinterpret_cast<const f32x4*>(xp + 4 * j);
#pragma unroll
      for (int co = 0; co < CS; ++co)
#pragma unroll
        for (int ch = 0; ch < 16; ++ch) acc[co] = fmaf(x[ch >> 2][ch & 3], wp[ch * CS + co], acc[co]);
    }
  }
  const int yy = y0 + r, xx = x0 + c;
  if (yy < H && xx < W) {
    const long long o = (((long long)img * H + yy) * W + xx) * CS;
#pragma unroll
    for (int co = 0; co < CS; ++co) {
      float v = mvk_act(acc[co] + (g.bias ? g.bias[co] : 0.f), g.act);
      if (g.mask_src) v *= mvk_act_grad_from_out(g.mask_src[o + co], g.mask_act);
      g.Y[o + co] = v;
    }
  }
}

// ---- weight gradient with an image on one side ---------------------------------------------------------------------------
// slab[block][index in dWref] = sum over the block's positions of S[pos + sgn * off(tap)][cs] * B[pos][cb]:
//   the image is the INPUT  (Cin = CS):  S = X,  B = dY, sgn = +1, dWref[cb][cs][tap]
//   the image is the OUTPUT (Cout = CS): S = dY, B = X,  sgn = -1, dWref[cs][cb][tap]   (the same sum re-indexed by the input position)
constexpr int C3_WGRAD_SLABS = 1024;  // most workgroups (= slab rows) of the weight-gradient kernel
struct C3WArgs {
  const float* S;
  const float* B;
  float* slab;
  int n, H, W, CB, sgn, image_is_input;
};

// Thread = (4 channels of B: cg) x (one of 4 groups of (tap, cs) pairs: sub) x (one of PG = 64 / (CB / 4) position
// groups: pg); NP = ceil(9 CS / 4) accumulator quads per thread.  Per position: one 16-byte load of B (the lanes of a
// wave that share cg read the same address), NP broadcast LDS reads of S, 4 NP FMAs; 8 loads in flight per lane.
// The PG partial sums are added in a fixed order through LDS, the workgroups' results go to slab rows.
template <int CS>
__global__ __launch_bounds__(256) void conv3_small_wgrad_kernel(const C3WArgs g) {
  constexpr int NPAIR = 9 * CS, NP = (NPAIR + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) float st[];  // [(RB + 2)][(W + 2)][CS] zero halo, then red[PG][NPAIR * CB]
  const int W = g.W, H = g.H, CB = g.CB;
  const int TW = (W + 2) * CS;
  const int CG = CB / 4, PG = 64 / CG;
  const int cg = threadIdx.x % CG, sub = (threadIdx.x / CG) & 3, pg = threadIdx.x / (4 * CG);
  float* red = st + (((C3_RB + 2) * TW + 3) & ~3);
  f32x4 acc[NP];
  int soff[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int q = min(sub + 4 * j, NPAIR - 1);  // a padding pair repeats the last one (never stored)
    const int tap = q / CS, cs = q % CS;
    soff[j] = ((1 + g.sgn * (tap / 3 - 1)) * (W + 2) + (1 + g.sgn * (tap % 3 - 1))) * CS + cs;
  }
  const int bands = (H + C3_RB - 1) / C3_RB;
  for (int item = blockIdx.x; item < g.n * bands; item += gridDim.x) {
    const int img = item / bands, y0 = (item % bands) * C3_RB;
    const int rows = min(C3_RB, H - y0);
    __syncthreads();
    const float* simg = g.S + (long long)img * H * W * CS;
    for (int i = threadIdx.x; i < (rows + 2) * TW; i += 256) {
      const int r = i / TW, rem = i - r * TW;
      const int xcol = rem / CS - 1, c = rem % CS, yy = y0 + r - 1;
      st[i] = (yy >= 0 && yy < H && xcol >= 0 && xcol < W) ? simg[((long long)yy * W + xcol) * CS + c] : 0.f;
    }
    __syncthreads();
    const float* bimg = g.B + (((long long)img * H + y0) * W) * CB + 4 * cg;
    if (W >= PG) {
      // ONE loop over the band's rows x W positions (they are contiguous in B): with a loop per row a 28-wide map gives each position
      // group 7 iterations — the 8-fold unrolled body never runs, the remainder loop keeps ONE 16-byte load in flight, and the kernel
      // waited for memory latency (262 us at cfg4's decoder batch for 63 us of FMAs).  The row / column of a position are carried
      // along (no division); the same sums in the same order when W is a multiple of the position groups.
      const int npos = rows * W;
      int x = pg, soffr = 0;  // soffr = r * TW (the staged rows carry a halo column on each side)
#pragma unroll 8
      for (int idx = pg; idx < npos; idx += PG) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bimg + (long long)idx * CB);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          const float sv = st[soffr + x * CS + soff[j]];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[j][e] = fmaf(sv, b4[e], acc[j][e]);
        }
        x += PG;
        const bool wrap = x >= W;
        x = wrap ? x - W : x;
        soffr = wrap ? soffr + TW : soffr;
      }
    } else
    for (int r = 0; r < rows; ++r) {
      const float* srow = st + r * TW;
      const float* brow = bimg + (long long)r * W * CB;
#pragma unroll 8
      for (int x = pg; x < W; x += PG) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(brow + x * CB);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          const float sv = srow[x * CS + soff[j]];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[j][e] = fmaf(sv, b4[e], acc[j][e]);
        }
      }
    }
  }
  __syncthreads();
  const int total = NPAIR * CB;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int q = sub + 4 * j;
    if (q < NPAIR) *reinterpret_cast<f32x4*>(red + pg * total + q * CB + 4 * cg) = acc[j];
  }
  __syncthreads();
  float* slab = g.slab + (long long)blockIdx.x * total;
  for (int i = threadIdx.x; i < total; i += 256) {
    float v = red[i];
    for (int p = 1; p < PG; ++p) v += red[p * total + i];
    const int q = i / CB, cb = i - q * CB, tap = q / CS, cs = q - tap * CS;
    slab[g.image_is_input ? ((cb * CS + cs) * 9 + tap) : ((cs * CB + cb) * 9 + tap)] = v;
  }
}

}  // namespace

namespace mvk {

// 1: shape not covered (the caller continues with the GEMM engine)
int conv3_smallcin(const float* X, const float* Wp, const float* bias, float* Y, int n, int H, int W, int Cin, int Cout, int act,
                   const float* mask_src, int mask_act, hipStream_t s, float* y_amax) {
  static const int off = mvk_tune("MVK_CONV3SMALL") ? atoi(mvk_tune("MVK_CONV3SMALL")) == 0 : 0;
  if (off || Cin < 1 || Cin > 4 || Cout % 4 != 0 || Cout < 4 || Cout > 1024 || W > 256 || n < 1) return 1;
  if (!mvk_aligned16(Wp) || !mvk_aligned16(Y) || (bias && !mvk_aligned16(bias)) || (mask_src && !mvk_aligned16(mask_src))) return 1;
  C3Args a{X, Wp, bias, Y, mask_src, n, H, W, Cin, Cout, act, mask_act, y_amax};
  const int bands = (H + C3_RB - 1) / C3_RB;
  const size_t lds = (size_t)(C3_RB + 2) * (W + 2) * Cin * sizeof(float);
  const dim3 grid((unsigned)(n * bands));
  switch (Cin) {
    case 1: hipLaunchKernelGGL(conv3_smallcin_kernel<1>, grid, dim3(256), lds, s, a); break;
    case 2: hipLaunchKernelGGL(conv3_smallcin_kernel<2>, grid, dim3(256), lds, s, a); break;
    case 3: hipLaunchKernelGGL(conv3_smallcin_kernel<3>, grid, dim3(256), lds, s, a); break;
    default: hipLaunchKernelGGL(conv3_smallcin_kernel<4>, grid, dim3(256), lds, s, a); break;
  }
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

// 1: shape not covered
int conv3_smallcout(const float* X, const float* Wp, const float* bias, float* Y, int n, int H, int W, int Cin, int Cout, int act,
                    const float* mask_src, int mask_act, hipStream_t s) {
  static const int off = mvk_tune("MVK_CONV3SMALL") ? atoi(mvk_tune("MVK_CONV3SMALL")) == 0 : 0;
  if (off || Cout < 1 || Cout > 4 || Cin % 16 != 0 || Cin < 16 || Cin > 256 || n < 1 || !mvk_aligned16(X)) return 1;
  C3Args a{X, Wp, bias, Y, mask_src, n, H, W, Cin, Cout, act, mask_act, nullptr};
  static const int mfma = mvk_tune("MVK_SMALLCOUT_MFMA") ? atoi(mvk_tune("MVK_SMALLCOUT_MFMA")) : 0;  // A/B: the MFMA kernel
  if (!mfma) {
    const dim3 vgrid((unsigned)((long long)n * ((W + C3V_TC - 1) / C3V_TC) * ((H + C3V_TR - 1) / C3V_TR)));
    switch (Cout) {
      case 1: hipLaunchKernelGGL(conv3_smallcout_v_kernel<1>, vgrid, dim3(256), 0, s, a); break;
      case 2: hipLaunchKernelGGL(conv3_smallcout_v_kernel<2>, vgrid, dim3(256), 0, s, a); break;
      case 3: hipLaunchKernelGGL(conv3_smallcout_v_kernel<3>, vgrid, dim3(256), 0, s, a); break;
      default: hipLaunchKernelGGL(conv3_smallcout_v_kernel<4>, vgrid, dim3(256), 0, s, a); break;
    }
    MVK_CHECK_LAUNCH();
    return MVK_OK;
  }
  const int tcols = (W + C3_TC - 1) / C3_TC, trows = (H + C3_TR - 1) / C3_TR;
  const size_t lds = ((size_t)(C3_TR + 2) * (C3_TC + 2) * (Cin + 4) + (size_t)9 * Cin * 4) * sizeof(float);
  const dim3 grid((unsigned)(n * tcols * trows));
#define MVK_C3SC(CS_)                                                                                                   \
  {                                                                                                                     \
    if (lds > 64 * 1024)                                                                                                \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_smallcout_kernel<CS_>),                             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                  \
    hipLaunchKernelGGL(conv3_smallcout_kernel<CS_>, grid, dim3(256), lds, s, a);                                        \
  }
  switch (Cout) {
    case 1: MVK_C3SC(1) break;
    case 2: MVK_C3SC(2) break;
    case 3: MVK_C3SC(3) break;
    default: MVK_C3SC(4) break;
  }
#undef MVK_C3SC
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

// slab: [*nz][9 * Cin * Cout] partial gradients already in dWref order; 1: shape not covered
int conv3_small_wgrad(const float* X, const float* dY, float* slab, long long slab_floats, int n, int H, int W, int Cin, int Cout,
                      int* nz, hipStream_t s) {
  static const int off = mvk_tune("MVK_CONV3SMALL") ? atoi(mvk_tune("MVK_CONV3SMALL")) == 0 : 0;
  const bool in_small = Cin <= 4, out_small = Cout <= 4;
  if (off || n < 1 || (!in_small && !out_small) || W > 256) return 1;
  const int CS = in_small ? Cin : Cout, CB = in_small ? Cout : Cin;
  const int CG = CB / 4;
  if (CB % 4 != 0 || CG < 1 || CG > 64 || (64 % CG) != 0 || !slab) return 1;  // CB in {4, 8, 16, 32, 64, 128, 256}
  const float* S = in_small ? X : dY;
  const float* B = in_small ? dY : X;
  if (!mvk_aligned16(B) || !mvk_aligned16(slab)) return 1;
  const long long total = 9ll * Cin * Cout;
  const int bands = (H + C3_RB - 1) / C3_RB;
  long long grid = (long long)n * bands;
  if (grid > C3_WGRAD_SLABS) grid = C3_WGRAD_SLABS;
  if (grid * total > slab_floats) grid = slab_floats / total;
  if (grid < 1) return 1;
  C3WArgs a{S, B, slab, n, H, W, CB, in_small ? 1 : -1, in_small ? 1 : 0};
  const size_t tile = ((size_t)(C3_RB + 2) * (W + 2) * CS + 3) & ~(size_t)3;
  const size_t lds = (tile + (size_t)(64 / CG) * total) * sizeof(float);
  if (lds > 64 * 1024) return 1;
  switch (CS) {
    case 1: hipLaunchKernelGGL(conv3_small_wgrad_kernel<1>, dim3((unsigned)grid), dim3(256), lds, s, a); break;
    case 2: hipLaunchKernelGGL(conv3_small_wgrad_kernel<2>, dim3((unsigned)grid), dim3(256), lds, s, a); break;
    case 3: hipLaunchKernelGGL(conv3_small_wgrad_kernel<3>, dim3((unsigned)grid), dim3(256), lds, s, a); break;
    default: hipLaunchKernelGGL(conv3_small_wgrad_kernel<4>, dim3((unsigned)grid), dim3(256), lds, s, a); break;
  }
  MVK_CHECK_LAUNCH();
  *nz = (int)grid;
  return MVK_OK;
}

}  // namespace mvk
