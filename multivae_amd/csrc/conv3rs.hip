// Register-stationary-weight kernels for the 3x3 / stride-1 / pad-1 convolutions of the ResNet blocks
// (reference: models/nn/mmnist.py:214-252 ResnetBlock, models/nn/cub.py:250-293 ResnetBlock; the layers of
// EncoderResnetMMNIST / DecoderResnetMMNIST / CUB_Resnet_Encoder / CUB_Resnet_Decoder with 64 or 128 input channels).
//
// Same data flow as imgconv.hip, re-derived for a 3x3 window on maps of ANY size:
//
//   * one workgroup per CU, 4 waves, one wave per SIMD, up to 512 registers per lane; every wave keeps an
//     [18 k-steps x 16 k][32 n] slice of the layer's GEMM weight W[(tap, ci)][co] as three bf16 pieces in 216 registers of
//     the accumulator half (AGPRs) for the whole launch: 64 input channels -> 2 waves share a 32-column tile (9 taps x 4
//     channel chunks = 36 k-steps), 128 input channels -> 4 waves; wider outputs take more workgroup types;
//   * THE IMAGES ARE ONE STREAM OF POSITIONS.  Every image is laid out with one zero column on the right of each row and
//     one zero row below it: position p = (img (H+1) + y) (W+1) + x.  The zero column is the right halo of its row AND the
//     left halo of the next one, the zero row the bottom halo of its image AND the top halo of the next — so tap (dy, dx) of
//     output position p is input position p + dy (W+1) + dx for every p, with no validity test, and a 32-row GEMM tile is
//     32 consecutive positions whatever W is (the outputs computed at the zero positions, 1.5 % of a 64x64 map, 23 % of a
//     7x7 one, are dropped).  A workgroup walks its share of the stream once: the positions live in an LDS ring
//     (slot = p mod RING, one slot = the three bf16 piece rows of all channels + 16 bytes of padding: conflict-free
//     ds_read_b128 fragments), chunk T+D+1 is converted while tile T is multiplied;
//   * a table of 32 pixel indices per chunk (or -1 for the zero positions), computed incrementally by 32 lanes and
//     published through LDS, drives the staging loads, the mask / residual loads and the stores, so nothing in the main
//     loop divides or branches per element;
//   * the main loop is the two-tile-latency pipeline of imgconv_kernel: tile T carries the accumulator sum + exchange of
//     tile T-1, the epilogue of tile T-2, the conversion of chunk T+D+1 and ONE barrier; the epilogue fuses bias,
//     LeakyReLU / ReLU, the activation derivative of the layer the result lands in, the residual sum of the block
//     (res + alpha * result) and the bias-gradient column sums.
#include <cstdlib>
#include <type_traits>

#include "bf3.hpp"

#ifdef MVK_NO_RFL
#define MVK_RFL(x) (x)
#else
#define MVK_RFL(x) __builtin_amdgcn_readfirstlane(x)
#endif

#ifndef MVK_C3_SCHED
#define MVK_C3_SCHED 4  // "others" per MFMA of the scheduling pipeline (0 = hipcc's own order)
#endif
#ifndef MVK_C3_SCHED2
#define MVK_C3_SCHED2 2  // multiplier of MVK_C3_SCHED in the scaled-fp16 form (3 MFMAs per product instead of 6)
#endif
#ifndef MVK_C3_F16ACC
// accumulators of the scaled-fp16 form: 2 = main + cross (dependent MFMAs on the same accumulator issue back to back),
// 3 = one main + a cross per k-step of a pair, 4 = main + cross per k-step.  A/B (64 -> 64 @64x64, n = 128): 153 / 162 / 165 us
#define MVK_C3_F16ACC 2
#endif
#ifndef MVK_C3_DIST
// scaled-fp16 form: pairs of k-steps the A fragments are read ahead of their MFMAs (1 or 2).  A/B: 151 vs 167 us — the LDS
// latency was not what the loop waits for, and 2 moves the barrier one pair up
#define MVK_C3_DIST 1
#endif
#ifndef MVK_C3_NW2
// waves per workgroup of the scaled-fp16 form: 4 = one per SIMD, 8 = two per SIMD with 72 weight registers each (twice the
// tap split: 4 / 8 waves share a column tile).  A/B (64 -> 64 @64x64, n = 128, masked form): 157 vs 158 us, plain 129 vs 139 us
// — what the second wave covers, the wider exchange and the 8-wave barrier (19 % of the cycles) take back.
#define MVK_C3_NW2 4
#endif
#ifndef MVK_C3_BAR
#define MVK_C3_BAR 7    // the pair behind which the tile's barrier sits (6: one more pair of cover for what follows it)
#endif

namespace mvk {

#ifdef MVK_C3PROF  // tools/conv3_phase.py: per-wave cycle counters (total, waiting at the tile barrier)
__device__ unsigned long long* g_c3_dbg = nullptr;
#endif

struct C3Args {
  const float* X;        // input [n][H][W][CIN] (NHWC)
  const float* Wp;       // fp32 GEMM pack W[(tap * CIN + ci)][COUT] (mvk_pack_weights kind c3)
  const float* bias;     // [COUT] or null
  float* Y;              // output [n][H][W][COUT]
  const float* act_src;  // tensor of the output's shape whose activation derivative multiplies the result, or null
  const float* res;      // residual of the output's shape: Y = res + res_alpha * result, or null
  float* colsum_part;    // [workers][COUT] per-workgroup column sums of the result (before the residual), or null
  unsigned xbytes, ybytes;  // sizes of X and of Y / act_src / res in bytes (< 2^32 - 4096): buffer bounds
  int n, H, W;
  float aslope, mslope;  // negative-side slopes of the output activation (1 = none) and of the mask (src_act)
  float res_alpha;
  float islope;          // negative-side slope of the activation applied to X while it is staged (1 = none)
  float pre_scale;       // the convolution sum is multiplied by this before the bias (1 = none)
  int ring;              // LDS ring length in positions (multiple of 32, >= 32 (2 D + 2))
  int D;                 // reach of the window in 32-position chunks: ceil((W + 2) / 32)
  int tiles;             // ceil(n (H+1) (W+1) / 32)
  mvk_prof_slot* prof;
  // scaled-fp16 form (NP = 2, bf3.hpp): upper bounds of max |X| and max |Wp| (device scalars, both required)
  const float* x_amax;
  const float* w_amax;
  float* y_amax;         // either form, optional: max |Y| is published here (atomic max; must hold 0 before the launch)
  float* Y2;             // scaled-fp16 residual form, optional: act(conv + bias), the value before res + res_alpha * (.)
  // a launch over a SLICE of the input channels (a 256-channel layer as two 128-channel launches, mvk_conv3x3_s_part): X points at
  // the slice's first channel, Wp at its first row of tap 0
  unsigned xpix_bytes;   // bytes between consecutive pixels of X (4 CIN for a whole tensor)
  int wtap_rows;         // rows of Wp per tap (CIN for a whole tensor)
  float res_pre;         // 1: the residual is added BEFORE the activation (Y = act(conv + bias + res): the partial sum of the
                         // other slice), res_alpha unused; 0: Y = res + res_alpha * act(conv + bias)
};

// NP = pieces per operand element: 3 bf16 pieces (6 MFMAs per product) or 2 scaled fp16 pieces (3 MFMAs per product)
template <int CIN, int COUT, int NP = 3>
struct C3Cfg {
  static constexpr int CHUNKS = CIN / 16;            // 16-channel k-steps per tap
  static constexpr int KALL = 9 * CHUNKS;            // k-steps of one output element
  // The bf16 form needs 216 weight registers per wave: ONE wave per SIMD, every latency covered by software pipelining.  The
  // fp16 form has half the MFMAs (1730 cycles per tile) to hide the same staging / exchange / epilogue work behind, and that
  // work alone takes 3260 cycles per tile (tools/conv3_phase.py, subtraction builds): the loop is bound by it, not by the
  // matrix pipe.  MVK_C3_NW2 = 8 runs two waves per SIMD with 72 weight registers each; it measured no faster (see above).
  static constexpr int NW = NP == 2 ? MVK_C3_NW2 : 4;  // waves per workgroup
  static constexpr int KPW = 72 / NW;                // k-steps per wave: 18 = 216 (NP = 3) / 144 (NP = 2) weight registers, 9 = 72
  static constexpr int HPP = KPW / 9;                // k-steps per loop iteration ("pair")
  static constexpr int KSPLIT = KALL / KPW;          // waves sharing one 32-column tile
  static constexpr int NCT = COUT / 32;
  static constexpr int ROLES = NCT * KSPLIT;
  static constexpr int WG_TYPES = ROLES / NW;
  static constexpr int S = 2 * NP * CIN + 16;        // bytes per ring slot: NP pieces x CIN halves + pad ((S / 16) odd)
  static constexpr int NF4 = CIN / (8 * NW);         // float4 staging units per thread and chunk
  static constexpr int OWN = 16 / KSPLIT;            // accumulator registers (output rows per lane) a wave finishes
  static constexpr int XWAVE = (KSPLIT - 1) * OWN * 64 * 4;
  static constexpr int XBUF = NW * XWAVE;
  static constexpr int PTAB_INTS = 16 * 32;
  // largest window reach (chunks) whose ring fits the LDS: 3 = W <= 94, 2 = W <= 62, 1 = W <= 30
  static constexpr int MAXD = CIN == 64 ? 3 : (NP == 2 ? 2 : 1);
  static_assert(KALL % KPW == 0 && ROLES % NW == 0 && NW % KSPLIT == 0 && KPW % 9 == 0 && NF4 >= 1, "roles");
  static_assert((S / 16) % 2 == 1, "odd 16-byte stride: conflict-free fragments");
  __host__ __device__ static constexpr int lds_bytes(int ring) { return ring * S + 2 * XBUF + PTAB_INTS * 4 + NW * 32 * 4; }
};

__device__ __forceinline__ bf16x8 c3_pack8(const unsigned (&d)[4]) {
  u32x4 v = {d[0], d[1], d[2], d[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int CIN, int COUT, bool HAS_SRC, bool HAS_RES, int NP, bool DUAL = false>
__device__ __forceinline__ void c3rs_body(const C3Args& g) {
  using T = C3Cfg<CIN, COUT, NP>;
  using frag = std::conditional_t<NP == 3, bf16x8, f16x8>;
  // scaled-fp16 form: the operands are multiplied by sx / sw on their way into pieces, the result by 1 / (sx sw)
  const float sx = NP == 2 ? f16_scale_of(*g.x_amax) : 1.f, sw = NP == 2 ? f16_scale_of(*g.w_amax) : 1.f;
  const float inv_sx = f16_inv_scale(sx), inv_sw = f16_inv_scale(sw);
  extern __shared__ __attribute__((aligned(16))) char lds[];
  mvk_prof_begin(g.prof);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, kg = lane >> 5;
  // workgroup types of one worker read the same positions: all of them on ONE XCD (block b runs on XCD b % 8; imgconv.hip)
  int wgtype = blockIdx.x % T::WG_TYPES;
  int worker = blockIdx.x / T::WG_TYPES;
  const int workers = gridDim.x / T::WG_TYPES;
#ifndef MVK_NO_XCDMAP
  if (T::WG_TYPES > 1 && gridDim.x % (8 * T::WG_TYPES) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    wgtype = slot % T::WG_TYPES;
    worker = xcd * (gridDim.x / (8 * T::WG_TYPES)) + slot / T::WG_TYPES;
  }
#endif
  const int role = wgtype * T::NW + wave;
  const int ct = role / T::KSPLIT, ks = role % T::KSPLIT;
  const int ncol = ct * 32 + col;

  // ---- weights: 18 k-steps x NP pieces, resident for the whole launch -----------------------------------------------------
  frag Bw[T::KPW][NP];
#pragma unroll
  for (int i = 0; i < T::KPW; ++i) {
    const int gk = ks * T::KPW + i;
    const long long rowbase = (long long)(gk / T::CHUNKS) * g.wtap_rows + (gk % T::CHUNKS) * 16 + kg * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = g.Wp[(rowbase + e) * COUT + ncol];
    unsigned p[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if constexpr (NP == 3) bf3_split(v[2 * e], v[2 * e + 1], p[0][e], p[1][e], p[2][e]);
      else f16_split(v[2 * e] * sw, v[2 * e + 1] * sw, p[0][e], p[1][e]);
    }
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) Bw[i][pc] = __builtin_bit_cast(frag, u32x4{p[pc][0], p[pc][1], p[pc][2], p[pc][3]});
  }
  // AGPR citizens (see imgconv_kernel): MFMA reads its B operand from a[...] directly, no v_accvgpr_read per use
#pragma unroll
  for (int i = 0; i < T::KPW; ++i)
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) {
      frag t = Bw[i][pc];
      asm volatile("" : "=a"(Bw[i][pc]) : "0"(t));
    }

  // ---- LDS ---------------------------------------------------------------------------------------------------------------
  const int RING = g.ring, NCH = RING / 32;
  char* const ringp = lds;
  char* const xbase = lds + RING * T::S;
  int* const ptab = reinterpret_cast<int*>(xbase + 2 * T::XBUF);
  float* const csred = reinterpret_cast<float*>(ptab + T::PTAB_INTS);

  const int H = g.H, W = g.W, W1 = W + 1, H1 = H + 1, PB = W1 * H1;
  const int dx32 = 32 % W1, dv32 = 32 / W1;
  const int D = g.D;
  const int T0 = MVK_RFL((int)((long long)g.tiles * worker / workers));  // uniform trip count (see imgconv.hip)
  const int T1 = MVK_RFL((int)((long long)g.tiles * (worker + 1) / workers));
  const int NT = T1 - T0;

  // pixel-index table: entry [c & 15][l] = pixel index of position 32 c + l, or -1 (zero column / zero row / outside)
  int pimg, pv, px;
  {
    const int p0 = 32 * (T0 - D) + col + PB;  // >= 0: PB >= 32 D for every supported shape (checked by the launcher)
    pimg = p0 / PB - 1;
    const int rem = p0 % PB;
    pv = rem / W1;
    px = rem % W1;
  }
  int cnext = T0 - D;
  auto ptab_store = [&]() {
    const bool ok = pimg >= 0 && pimg < g.n && pv < H && px < W;
    const int val = ok ? (pimg * H + pv) * W + px : -1;
    if (wave == 0 && kg == 0) ptab[(cnext & 15) * 32 + col] = val;
    ++cnext;
    px += dx32;
    pv += dv32;
    if (px >= W1) {
      px -= W1;
      pv += 1;
    }
    if (pv >= H1) {
      pv -= H1;
      pimg += 1;
    }
  };

  // staging: float4 unit f = tid + k * 256 of a chunk's [32 positions][CIN] block
  int soff[T::NF4], spos[T::NF4], sc4[T::NF4];
#pragma unroll
  for (int k = 0; k < T::NF4; ++k) {
    const int f = tid + k * (T::NW * 64);
    spos[k] = f / (CIN / 4);
    sc4[k] = (f % (CIN / 4)) * 4;
    soff[k] = spos[k] * T::S + (f % (CIN / 4)) * 8;
  }
  // A loaded value is first touched ONE TILE after its load was issued (the zero positions are selected at conversion
  // time, not at load time): with one wave per SIMD a wait on a fresh load stalls the matrix pipe for a memory latency.
  // Every global access is a raw buffer operation with a 32-bit byte offset: pixel index -1 (a zero position, or a row that
  // must not be stored) gives an offset beyond the buffer's size, where loads return 0 and stores are dropped — no
  // selects, no 64-bit address arithmetic and no branches in the loop.
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)g.X, 0, (int)g.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)g.Y, 0, (int)g.ybytes, 0x00020000);
  // DUAL: the activations BEFORE the residual sum go to Y2 as well (a post-activation ResNet block keeps them for its backward)
  static_assert(!DUAL || (HAS_RES && !HAS_SRC), "the second store exists for the forward residual form");
  const __amdgpu_buffer_rsrc_t rsY2 = __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL ? g.Y2 : g.Y), 0, (int)g.ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsM =
      __builtin_amdgcn_make_buffer_rsrc((void*)(HAS_SRC ? g.act_src : g.Y), 0, (int)g.ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)(HAS_RES ? g.res : g.Y), 0, (int)g.ybytes, 0x00020000);
  const unsigned xpb = g.xpix_bytes;
  f32x4 raw[T::NF4];
  auto unit_pix = [&](int c, int k) { return ptab[(c & 15) * 32 + spos[k]]; };
  auto load_unit_at = [&](int pix, int k) {
    raw[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (unsigned)pix * xpb + sc4[k] * 4u, 0, 0));
  };
  auto load_unit = [&](int c, int k) { load_unit_at(unit_pix(c, k), k); };
  const float islope = g.islope, sxn = sx * g.islope;
  auto write_vals = [&](int wbase, int k, const f32x4& v) {  // wbase = byte offset of the chunk's first slot
    // the activation of the layer that produced X, applied on the way into LDS (X is then stored before its activation)
    char* d = ringp + wbase + soff[k];
    if constexpr (NP == 3) {
      const float r0 = v[0] > 0.f ? v[0] : v[0] * islope, r1 = v[1] > 0.f ? v[1] : v[1] * islope;
      const float r2 = v[2] > 0.f ? v[2] : v[2] * islope, r3 = v[3] > 0.f ? v[3] : v[3] * islope;
      unsigned a0, a1, a2, b0, b1, b2;
      bf3_split(r0, r1, a0, a1, a2);
      bf3_split(r2, r3, b0, b1, b2);
      *reinterpret_cast<u32x2*>(d) = u32x2{a0, b0};
      *reinterpret_cast<u32x2*>(d + 2 * CIN) = u32x2{a1, b1};
      *reinterpret_cast<u32x2*>(d + 4 * CIN) = u32x2{a2, b2};
    } else {  // the operand scale rides on the activation's two slopes
      unsigned a0, a1, b0, b1;
      f16_split_sv(v[0], v[1], v[0] > 0.f ? sx : sxn, v[1] > 0.f ? sx : sxn, a0, a1);
      f16_split_sv(v[2], v[3], v[2] > 0.f ? sx : sxn, v[3] > 0.f ? sx : sxn, b0, b1);
      *reinterpret_cast<u32x2*>(d) = u32x2{a0, b0};
      *reinterpret_cast<u32x2*>(d + 2 * CIN) = u32x2{a1, b1};
    }
  };
  auto write_unit = [&](int wbase, int k) { write_vals(wbase, k, raw[k]); };
  auto chunk_slot = [&](int c) { return (((c % NCH) + NCH) % NCH) * 32 * T::S; };

  // ---- prologue: table entries T0-D .. T0+D+2, chunks T0-D .. T0+D in the ring, chunk T0+D+1 in registers ---------------
  if (NT > 0) {
    for (int c = T0 - D; c <= T0 + D + 2; ++c) ptab_store();
  }
  __syncthreads();
  if (NT > 0) {  // all loads of the 2 D + 1 start-up chunks in flight at once (one memory latency, not 2 D + 1)
    constexpr int MAXCH = T::MAXD * 2 + 1;
    f32x4 praw[MAXCH][T::NF4];
#pragma unroll
    for (int u = 0; u < MAXCH; ++u)
      if (u < 2 * D + 1) {
#pragma unroll
        for (int k = 0; k < T::NF4; ++k)
          praw[u][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                     rsX, (unsigned)unit_pix(T0 - D + u, k) * xpb + sc4[k] * 4u, 0, 0));
      }
#pragma unroll
    for (int k = 0; k < T::NF4; ++k) load_unit(T0 + D + 1, k);
#pragma unroll
    for (int u = 0; u < MAXCH; ++u)
      if (u < 2 * D + 1) {
        const int wb = chunk_slot(T0 - D + u);
#pragma unroll
        for (int k = 0; k < T::NF4; ++k) write_vals(wb, k, praw[u][k]);
      }
  }
  __syncthreads();

  // column sums (the bias gradient of the layer below) exist in the backward-data form only: mask, no residual
  constexpr bool WITH_CSUM = HAS_SRC && !HAS_RES;
  const float bias = g.bias ? g.bias[ncol] : 0.f;
#ifdef MVK_C3PROF
  const unsigned long long c3_t0 = __builtin_readcyclecounter();
  unsigned long long c3_bar = 0;
#endif
  // residual before / behind the activation without a branch: v = a act(v0 + wpre rr) + wpost rr
  const float aslope = g.aslope, mslope = g.mslope, alpha = g.res_pre != 0.f ? 1.f : g.res_alpha;
  const float wpre = g.res_pre != 0.f ? 1.f : 0.f, wpost = 1.f - wpre;
  const float pre_scale = NP == 2 ? g.pre_scale * inv_sw * inv_sx : g.pre_scale;
  float csum = 0.f, amax_l = 0.f;

  // ---- main loop, instantiated per tap-split rank (every "is this my slice" test is a compile-time fact) ----------------
  auto run = [&](auto ks_tag) {
    constexpr int KSC = decltype(ks_tag)::value;
    constexpr int TAP_LO = (KSC * T::KPW) / T::CHUNKS, TAP_HI = (KSC * T::KPW + T::KPW - 1) / T::CHUNKS;
    constexpr int NTAPW = TAP_HI - TAP_LO + 1;
    int delta[NTAPW];
#pragma unroll
    for (int j = 0; j < NTAPW; ++j) {
      const int tap = TAP_LO + j;
      delta[j] = ((tap / 3 - 1) * W1 + (tap % 3 - 1) + RING) * T::S;  // bytes: slot * S is never multiplied in the loop
    }
    const unsigned RINGB = (unsigned)RING * T::S;
    const int colB = col * T::S;
    auto frag_base = [&](int (&ab)[NTAPW], int tb) {  // byte offsets of this lane's A rows of the tile whose first slot is at byte tb
#pragma unroll
      for (int j = 0; j < NTAPW; ++j) {
        unsigned s = (unsigned)(tb + colB + delta[j]);
        s = min(s, s - RINGB);
        s = min(s, s - RINGB);
        ab[j] = (int)s + kg * 16;
      }
    };
    auto read_pair = [&](frag (&dst)[T::HPP][NP], const int (&ab)[NTAPW], int pr) {
#pragma unroll
      for (int h = 0; h < T::HPP; ++h) {
        const int gk = KSC * T::KPW + T::HPP * pr + h;
        const int j = gk / T::CHUNKS - TAP_LO, c = gk % T::CHUNKS;
#pragma unroll
        for (int pc = 0; pc < NP; ++pc)
          dst[h][pc] = *reinterpret_cast<const frag*>(ringp + ab[j] + pc * 2 * CIN + c * 32);
      }
    };
    // rows of the 32 x 32 tile this lane finishes: accumulator registers [KSC * OWN, KSC * OWN + OWN)
    int rrow[T::OWN];
#pragma unroll
    for (int o = 0; o < T::OWN; ++o) {
      const int r = KSC * T::OWN + o;
      rrow[o] = (r & 3) + 8 * (r >> 2) + 4 * kg;
    }
    const int wgrp = (wave / T::KSPLIT) * T::KSPLIT;  // first wave of this wave's column-tile group

    // accumulators of a tile: NP = 3: one per k-step of a pair; NP = 2: main (hi hi') and cross (hi lo' + lo hi') per k-step
    constexpr int NACC = NP == 3 ? 2 : MVK_C3_F16ACC;
    f32x16 pend[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) pend[i] = f32x16{0};
    // msk / rr / pix of a tile are fetched (global loads) while the NEXT tile is multiplied and consumed one tile after
    // that: two register sets indexed by the compile-time parity of the tile, no copy between them (hipcc hoists a copy
    // "prev = cur" to the last use of prev, i.e. right behind the load, and waits there for the memory latency)
    float own[T::OWN], res_next[T::OWN], res_prev[T::OWN], msk[2][T::OWN], rr[2][T::OWN];
    int pix[2][T::OWN];
#pragma unroll
    for (int o = 0; o < T::OWN; ++o) {
      own[o] = res_next[o] = res_prev[o] = 0.f;
      msk[0][o] = msk[1][o] = rr[0][o] = rr[1][o] = 0.f;
      pix[0][o] = pix[1][o] = -1;
    }
    auto finish_pending = [&](char* xb) {
      f32x16 sum;
      if constexpr (NP == 3) {
        sum = pend[0] + pend[1];
      } else {
        f32x16 cross;
        if constexpr (NACC == 4) {
          cross = pend[2] + pend[3];
          sum = pend[0] + pend[1];
        } else if constexpr (NACC == 3) {
          cross = pend[1] + pend[2];
          sum = pend[0];
        } else {
          cross = pend[1];
          sum = pend[0];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sum[r] = fmaf(cross[r], 1.f / 2048.f, sum[r]);
      }
#pragma unroll
      for (int r = 0; r < T::KSPLIT; ++r) {
        if (r == KSC) continue;
        const int slot = KSC < r ? KSC : KSC - 1;
        float* dst = reinterpret_cast<float*>(xb + (wgrp + r) * T::XWAVE + slot * T::OWN * 256);
#pragma unroll
        for (int o = 0; o < T::OWN; ++o) dst[o * 64 + lane] = sum[r * T::OWN + o];
      }
#pragma unroll
      for (int o = 0; o < T::OWN; ++o) own[o] = sum[KSC * T::OWN + o];
    };
    auto gather_result = [&](const char* xb) {  // ordered sum over the ranks: deterministic
      const float* src = reinterpret_cast<const float*>(xb + wave * T::XWAVE);
#pragma unroll
      for (int o = 0; o < T::OWN; ++o) {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < T::KSPLIT; ++r) {
          if (r == KSC) {
            v += own[o];
          } else {
            const int slot = r < KSC ? r : r - 1;
            v += src[(slot * T::OWN + o) * 64 + lane];
          }
        }
        res_next[o] = v;
      }
    };
    auto fetch_pix = [&](int P, int tabrow) {  // pixel indices of this lane's rows of the tile whose table row is tabrow
#pragma unroll
      for (int o = 0; o < T::OWN; ++o) pix[P][o] = ptab[tabrow * 32 + rrow[o]];
    };
    auto fetch_row = [&](int P, int o) {  // mask / residual of row o (issued one tile before they are consumed)
      const unsigned off = (unsigned)pix[P][o] * (COUT * 4u) + ncol * 4u;
      if (HAS_SRC) msk[P][o] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsM, off, 0, 0));
      if (HAS_RES) rr[P][o] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, off, 0, 0));
    };
    auto epilogue_row = [&](int Q, int o, bool valid) {  // Q: the parity whose fetch belongs to the tile in res_prev
      float v = fmaf(res_prev[o], pre_scale, bias);   // pre_scale carries 1 / (sx sw) in the scaled-fp16 form
      // (the other channel slice's partial sum; selected, not multiplied by 0, where it does not enter: a non-finite residual must
      // not turn into NaN inside the activation's input — ADVICE r5; wpre / wpost are wave-uniform)
      if (HAS_RES && wpre != 0.f) v = fmaf(wpre, rr[Q][o], v);
      v = fmaxf(v, v * aslope);                        // (leaky) ReLU / identity: 0 <= slope <= 1
      if (HAS_SRC) v = msk[Q][o] > 0.f ? v : v * mslope;
      const int px_ = valid ? pix[Q][o] : -1;
      if (WITH_CSUM) csum += px_ < 0 ? 0.f : v;
      if (DUAL) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsY2, (unsigned)px_ * (COUT * 4u) + ncol * 4u, 0, 0);
      if (HAS_RES) v = wpost != 0.f ? fmaf(alpha, v, wpost * rr[Q][o]) : alpha * v;
      amax_l = fmaxf(amax_l, px_ < 0 ? 0.f : fabsf(v));
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsY, (unsigned)px_ * (COUT * 4u) + ncol * 4u, 0, 0);
    };
    auto rotate = [&]() {
#pragma unroll
      for (int o = 0; o < T::OWN; ++o) res_prev[o] = res_next[o];
    };

    int tb = (int)(((long long)32 * T0) % RING) * T::S;  // first ring slot of tile T, in bytes
    int wch = (((T0 + D + 1) % NCH) + NCH) % NCH;       // ring chunk that chunk T+D+1 goes to
    int ab_cur[NTAPW], ab_nxt[NTAPW];
    // A fragments are read DIST pairs ahead of their MFMAs (the bf16 form's 12 MFMAs per pair cover one burst of LDS reads of
    // the four waves; the 6 of the fp16 form do not): set (pr + 9 P) % NSET holds pair pr of a tile of parity P
    constexpr int DIST = NP == 2 ? MVK_C3_DIST : 1, NSET = DIST + 1;
    // the barrier that publishes chunk T+D+1 must sit in front of the first read of tile T+1's fragments
    constexpr int BAR = MVK_C3_BAR < 9 - DIST ? MVK_C3_BAR : 8 - DIST;
    static_assert(18 % NSET == 0 && DIST >= 1, "the set of a pair must depend on the tile's parity only");
    frag a_q[NSET][T::HPP][NP];
    if (NT > 0) {
      frag_base(ab_cur, tb);
#pragma unroll
      for (int dd = 0; dd < DIST; ++dd) read_pair(a_q[dd], ab_cur, dd);
    }
    auto tile = [&](auto par_tag, int t) {
      constexpr int P = decltype(par_tag)::value;
      const int Tt = T0 + t;
      const bool valid2 = t >= 2;
      char* const xb = xbase + P * T::XBUF;
      const int wbase = wch * 32 * T::S;
      int tbn = tb + 32 * T::S;
      tbn = tbn >= (int)RINGB ? tbn - (int)RINGB : tbn;
      f32x16 acc[NACC];
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = f32x16{0};
      int upix[T::NF4];
#pragma unroll
      for (int pr = 0; pr < 9; ++pr) {
#ifndef MVK_C3X_NOFRAG  // MVK_C3X_*: subtraction experiments of tools/conv3_phase.py (timing only: the results are wrong)
        if (pr + DIST <= 8) read_pair(a_q[(pr + DIST + 9 * P) % NSET], ab_cur, pr + DIST);
        else read_pair(a_q[(pr + DIST + 9 * P) % NSET], ab_nxt, pr + DIST - 9);  // behind the barrier: the next tile's first pairs
#endif
        frag(&a_cur)[T::HPP][NP] = a_q[(pr + 9 * P) % NSET];
        if (pr == 0) {
#ifndef MVK_C3X_NOXCHG
          finish_pending(xb);
#endif
          fetch_pix(P, (Tt - 1) & 15);            // table rows published by the previous barrier: ONE wait per tile
#pragma unroll
          for (int k = 0; k < T::NF4; ++k) upix[k] = unit_pix(Tt + D + 2, k);
        }
        if (pr == 1) ptab_store();               // table row of chunk Tt + D + 3 (staging loads of the next tile)
        if (pr == 2) frag_base(ab_nxt, tbn);
        // conversion of chunk Tt+D+1 (loaded one tile ago) + the loads of chunk Tt+D+2: done before the barrier behind pair 7
        {
          constexpr int STEP = T::NF4 == 2 ? 3 : 2;
          constexpr int FIRST = BAR == 7 ? 1 : 0;  // the last conversion sits in front of the barrier
#ifndef MVK_C3X_NOSTAGE
          if (pr >= FIRST && pr <= BAR && (pr - FIRST) % STEP == 0 && (pr - FIRST) / STEP < T::NF4) {
            write_unit(wbase, (pr - FIRST) / STEP);
            load_unit_at(upix[(pr - FIRST) / STEP], (pr - FIRST) / STEP);
          }
#endif
        }
#pragma unroll
        for (int o = 0; o < T::OWN; ++o) {
          if (o * 8 / T::OWN + 1 != pr) continue;
#ifndef MVK_C3X_NOEPI
          fetch_row(P, o);                        // tile Tt-1: consumed two tiles later
          epilogue_row(P ^ 1, o, valid2);         // tile Tt-2
#endif
        }
#ifndef MVK_C3X_NOMFMA
        if constexpr (NP == 3) {
          constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB_[6] = {2, 1, 0, 1, 0, 0};  // small terms first
#pragma unroll
          for (int m = 0; m < 6; ++m) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[0][PA[m]], Bw[2 * pr][PB_[m]], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[1][PA[m]], Bw[2 * pr + 1][PB_[m]], acc[1], 0, 0, 0);
          }
        } else if constexpr (T::HPP == 1) {  // one k-step per iteration (two waves per SIMD): main, cross
          static_assert(NACC == 2, "two accumulators");
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[0][0], Bw[pr][1], acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[0][0], Bw[pr][0], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[0][1], Bw[pr][0], acc[1], 0, 0, 0);
        } else {  // cross terms (pieces 0 x 1, 1 x 0) and the main term in accumulators of their own
          constexpr int M0 = 0, M1 = NACC == 4 ? 1 : 0, C0 = NACC == 4 ? 2 : 1, C1 = NACC == 4 ? 3 : (NACC == 3 ? 2 : 1);
          acc[C0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[0][0], Bw[2 * pr][1], acc[C0], 0, 0, 0);
          acc[M0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[0][0], Bw[2 * pr][0], acc[M0], 0, 0, 0);
          acc[C1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[1][0], Bw[2 * pr + 1][1], acc[C1], 0, 0, 0);
          acc[M1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[1][0], Bw[2 * pr + 1][0], acc[M1], 0, 0, 0);
          acc[C0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[0][1], Bw[2 * pr][0], acc[C0], 0, 0, 0);
          acc[C1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[1][1], Bw[2 * pr + 1][0], acc[C1], 0, 0, 0);
        }
#else
        acc[0][pr] += __builtin_bit_cast(f32x4, a_cur[0][0])[0] + __builtin_bit_cast(f32x4, a_cur[T::HPP - 1][NP - 1])[1];
#endif
#ifndef MVK_C3X_NOXCHG
        if (pr == BAR + 1) gather_result(xb);  // behind the barrier: result of tile Tt-1
#endif
        if (MVK_C3_SCHED > 0 && T::NW == 4) {  // the same "other" work per pair behind half as many MFMAs in the fp16 form
          constexpr int OTHERS = (MVK_C3_SCHED > 0 ? MVK_C3_SCHED : 1) * (NP == 3 ? 1 : MVK_C3_SCHED2);
#pragma unroll
          for (int m = 0; m < 4 * NP; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x496, OTHERS, 0);
          }
        }
#ifdef MVK_C3PROF
        if (pr == BAR) {
          const unsigned long long b0 = __builtin_readcyclecounter();
          __syncthreads();
          c3_bar += __builtin_readcyclecounter() - b0;
        }
#else
        if (pr == BAR) __syncthreads();
#endif
      }
#pragma unroll
      for (int i = 0; i < NACC; ++i) pend[i] = acc[i];
      rotate();
#pragma unroll
      for (int j = 0; j < NTAPW; ++j) ab_cur[j] = ab_nxt[j];
      tb = tbn;
      wch = wch + 1 == NCH ? 0 : wch + 1;
    };
    auto drain = [&](auto par_tag) {
      constexpr int P = decltype(par_tag)::value;  // parity of the (virtual) tile NT
      // drain 1: finish the last tile, epilogue of the one before it
      char* const xb = xbase + P * T::XBUF;
      finish_pending(xb);
      fetch_pix(P, (T1 - 1) & 15);
#pragma unroll
      for (int o = 0; o < T::OWN; ++o) {
        fetch_row(P, o);
        epilogue_row(P ^ 1, o, NT >= 2);
      }
      __syncthreads();
      gather_result(xb);
      rotate();
      // drain 2: epilogue of the last tile
#pragma unroll
      for (int o = 0; o < T::OWN; ++o) epilogue_row(P, o, true);
    };
    std::integral_constant<int, 0> par0;
    std::integral_constant<int, 1> par1;
    for (int t = 0; t < NT; t += 2) {
      tile(par0, t);
      if (t + 1 < NT) tile(par1, t + 1);
    }
    if (NT > 0) {
      if (NT & 1) drain(par1);
      else drain(par0);
    }
  };
  const int ks_u = __builtin_amdgcn_readfirstlane(ks);
  if (ks_u == 0) run(std::integral_constant<int, 0>{});
  else if (ks_u == 1) run(std::integral_constant<int, 1>{});
  else if (T::KSPLIT > 2 && ks_u == 2) run(std::integral_constant<int, (T::KSPLIT > 2 ? 2 : 0)>{});
  else if (T::KSPLIT > 3 && ks_u == 3) run(std::integral_constant<int, (T::KSPLIT > 3 ? 3 : 0)>{});
  else if (T::KSPLIT > 4 && ks_u == 4) run(std::integral_constant<int, (T::KSPLIT > 4 ? 4 : 0)>{});
  else if (T::KSPLIT > 5 && ks_u == 5) run(std::integral_constant<int, (T::KSPLIT > 5 ? 5 : 0)>{});
  else if (T::KSPLIT > 6 && ks_u == 6) run(std::integral_constant<int, (T::KSPLIT > 6 ? 6 : 0)>{});
  else if (T::KSPLIT > 7) run(std::integral_constant<int, (T::KSPLIT > 7 ? 7 : 0)>{});

#ifdef MVK_C3PROF
  if (g_c3_dbg && lane == 0) {
    unsigned long long* o = g_c3_dbg + (blockIdx.x * 4 + wave) * 2;
    o[0] = __builtin_readcyclecounter() - c3_t0;
    o[1] = c3_bar;
  }
#endif
  if (g.y_amax) amax_publish(amax_l, g.y_amax, csred);  // uniform branch; csred is free until the column sums below
  if (WITH_CSUM && g.colsum_part) {  // fixed-order sum over the lanes / waves that share a column
    csum += __shfl_xor(csum, 32, 64);
    __syncthreads();
    if (kg == 0) csred[wave * 32 + col] = csum;
    __syncthreads();
    constexpr int CPW = (T::NW / T::KSPLIT) * 32;  // columns this workgroup covers
    if (tid < CPW) {
      const int grp = tid / 32, c_l = tid % 32;
      float s = 0.f;
      for (int r = 0; r < T::KSPLIT; ++r) s += csred[(grp * T::KSPLIT + r) * 32 + c_l];
      g.colsum_part[(long long)worker * COUT + wgtype * CPW + tid] = s;
    }
  }
  mvk_prof_end(g.prof);
}

// one wave per SIMD (up to 512 registers per lane) / two waves per SIMD (256)
template <int CIN, int COUT, bool HAS_SRC, bool HAS_RES, int NP, bool DUAL = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void c3rs_kernel(const C3Args g) {
  static_assert(C3Cfg<CIN, COUT, NP>::NW == 4, "four waves");
  c3rs_body<CIN, COUT, HAS_SRC, HAS_RES, NP, DUAL>(g);
}
template <int CIN, int COUT, bool HAS_SRC, bool HAS_RES, int NP, bool DUAL = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void c3rs_kernel8(const C3Args g) {
  static_assert(C3Cfg<CIN, COUT, NP>::NW == 8, "eight waves");
  c3rs_body<CIN, COUT, HAS_SRC, HAS_RES, NP, DUAL>(g);
}
template <int CIN, int COUT, bool HAS_SRC, bool HAS_RES, int NP, bool DUAL = false>
static const void* c3rs_entry() {
  if constexpr (C3Cfg<CIN, COUT, NP>::NW == 8) return reinterpret_cast<const void*>(c3rs_kernel8<CIN, COUT, HAS_SRC, HAS_RES, NP, DUAL>);
  else return reinterpret_cast<const void*>(c3rs_kernel<CIN, COUT, HAS_SRC, HAS_RES, NP, DUAL>);
}

template <int CIN, int COUT, int NP>
static int c3rs_launch(const C3Args& a, int* part_rows, hipStream_t s) {
  using T = C3Cfg<CIN, COUT, NP>;
  const int lds = T::lds_bytes(a.ring);
  if (lds > 160 * 1024) return 1;
  const void* all[5] = {c3rs_entry<CIN, COUT, false, false, NP>(), c3rs_entry<CIN, COUT, true, false, NP>(),
                        c3rs_entry<CIN, COUT, false, true, NP>(), c3rs_entry<CIN, COUT, true, true, NP>(),
                        c3rs_entry<CIN, COUT, false, true, NP, NP == 2>()};  // [4]: + the second store (scaled-fp16 form only)
  if (a.Y2 && (NP != 2 || a.act_src || !a.res)) return 1;
  static int attr_bytes = 0;
  if (attr_bytes < lds) {
    for (const void* f : all)
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return MVK_ELAUNCH;
    attr_bytes = 160 * 1024;
  }
  const int grid = 256;
  if (part_rows) *part_rows = grid / T::WG_TYPES;
  C3Args ap = a;
  ap.prof = prof_next(7, 2.0 * a.n * a.H * a.W * 9.0 * CIN * COUT);
  const int which = a.Y2 ? 4 : (a.act_src ? 1 : 0) + (a.res ? 2 : 0);
  void* kargs[1] = {&ap};
  if (hipLaunchKernel(all[which], dim3(grid), dim3(T::NW * 64), kargs, lds, s) != hipSuccess) return MVK_ELAUNCH;
  MVK_CHECK_LAUNCH();
  prof_fold(ap.prof, s);
  return MVK_OK;
}

static float c3_slope(int act) { return act == MVK_ACT_RELU ? 0.f : act == MVK_ACT_LEAKY02 ? 0.2f : 1.f; }

// =====================================================================================================================
// Weight gradient of the same layers:  dW[co][ci][kh][kw] += sum_p dY[p][co] X[p + (kh-1)(W+1) + (kw-1)][ci]
// =====================================================================================================================
// OUTPUT-stationary, on the same position stream: the reduction index of this GEMM is the position, and with the zero
// column / zero row in the stream (dY is staged with zeros there too) every tap is again a plain shift.  A workgroup type
// owns a [9 taps][64 ci][64 co] block of the gradient, wave (h, c) of it the nine 32 x 32 tiles [tap][32 ci][32 co] in 144
// accumulator registers for the whole launch; Cin / 64 x Cout / 64 workgroup types cover a layer, each staging only its 64
// input and 64 output channels.  X lives in the ring of the forward kernel (three bf16 piece rows per slot), dY in two
// 32-position buffers; both operands are position-major in LDS, i.e. transposed for the matrix cores, and are read with
// ds_read_b64_tr_b16 (imgwgrad_kernel) at per-lane slot addresses.  Per 32 positions a wave issues 2 k-steps x 3 kernel
// rows x 18 MFMAs (three accumulator chains interleaved); the A fragments of the next kernel row are read one row ahead.
// Every workgroup writes its block as one slab; the slabs are added in a fixed order (convref_reduce, deferred).
struct C3WArgs {
  const float* X;   // [n][H][W][Cin]
  const float* dY;  // [n][H][W][Cout]
  float* slab;      // [workers][9 Cin][Cout]
  float* dbpart;    // [workers][Cout] column sums of dY (x dy_scale): the bias gradient, or null
  float islope;     // negative-side slope of the activation applied to X while it is staged (1 = none)
  float dy_scale;   // both results are multiplied by this (the 0.1 of a ResNet block's residual branch)
  int n, H, W, Cin, Cout;
  int ring, D, tiles;
  unsigned xbytes, ybytes;  // sizes of X / dY in bytes (< 2^32 - 8192)
  mvk_prof_slot* prof;
  const float* x_amax;  // scaled-fp16 form (NP = 2): upper bounds of max |X| and max |dY| (device scalars)
  const float* y_amax;
};

#ifndef MVK_C3W_PAD
#define MVK_C3W_PAD 16
#endif
#ifndef MVK_C3W_SCHED
#define MVK_C3W_SCHED 4
#endif
constexpr int C3W_S = 6 * 64 + MVK_C3W_PAD;  // bytes per slot: 3 pieces x 64 channels bf16 + pad
static int c3w_lds_bytes(int ring) { return (ring + 64) * C3W_S + 16 * 32 * 4; }

typedef __bf16 c3_bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 c3_f16x4 __attribute__((ext_vector_type(4)));
template <typename F>
__device__ __forceinline__ F c3_tr_pair(const char* p0, const char* p1) {
  if constexpr (std::is_same<F, bf16x8>::value) {
    typedef __attribute__((address_space(3))) c3_bf16x4* lp;
    const c3_bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(p0));
    const c3_bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(p1));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  } else {
    typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
    typedef __attribute__((address_space(3))) h4* lp;
    const c3_f16x4 lo = __builtin_bit_cast(c3_f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(p0)));
    const c3_f16x4 hi = __builtin_bit_cast(c3_f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(p1)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

// NP = 3: bf16 pieces, 6 MFMAs per product.  NP = 2: scaled fp16 (bf3.hpp) with ONE accumulator per tile — the nine tap tiles
// leave no registers for a second one, so the 2^11 between the main and the cross terms sits in the operands instead:
//   X  (scale sx, max in [2^13, 2^14)):  xh = fp16(x sx), xl = fp16((x sx - xh) 2^11)
//   dY (scale sy, max in [2^3, 2^4)):    yH = fp16(y sy 2^11), yl = fp16(y sy 2^11 - yH), yh = fp16(y sy)
//   acc += xh yH + xh yl + xl yh  =  2^11 sx sy x y (1 + O(2^-22))                       (three fp16 MFMAs, small terms first)
// dY keeps full precision down to 2^-29 of its maximum (yH and yl are normal fp16 numbers there), X down to 2^-28.
template <int NP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void c3wg_kernel(const C3WArgs g) {
  constexpr int S = C3W_S;
  constexpr int NPA = NP, NPB = 3;  // planes of a slot: X has NP pieces, dY three in both forms
  using frag = std::conditional_t<NP == 3, bf16x8, f16x8>;
  const float sx = NP == 2 ? f16_scale_of(*g.x_amax) : 1.f;
  const float sy = NP == 2 ? f16_scale_of(*g.y_amax) * (1.f / 1024.f) : 1.f;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  mvk_prof_begin(g.prof);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, kg = lane >> 5;
  const int cit = g.Cin / 64, types = cit * (g.Cout / 64);
  int wgtype = blockIdx.x % types;
  int worker = blockIdx.x / types;
  const int workers = gridDim.x / types;
#ifndef MVK_NO_XCDMAP
  if (types > 1 && gridDim.x % (8 * types) == 0) {  // the types of a worker on one XCD (imgconv.hip)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    wgtype = slot % types;
    worker = xcd * (gridDim.x / (8 * types)) + slot / types;
  }
#endif
  const int ci0 = (wgtype % cit) * 64, co0 = (wgtype / cit) * 64;
  const int h = wave & 1, c = wave >> 1;  // this wave's 32-channel tiles of the block

  const int RING = g.ring, NCH = RING / 32;
  char* const ringp = lds;
  char* const dyp = lds + RING * S;  // two 32-slot buffers
  int* const ptab = reinterpret_cast<int*>(dyp + 64 * S);

  const int H = g.H, W = g.W, W1 = W + 1, H1 = H + 1, PB = W1 * H1;
  const int dx32 = 32 % W1, dv32 = 32 / W1;
  const int D = g.D;
  const int T0 = MVK_RFL((int)((long long)g.tiles * worker / workers));  // uniform trip count (see imgconv.hip)
  const int T1 = MVK_RFL((int)((long long)g.tiles * (worker + 1) / workers));
  const int NT = T1 - T0;

  int pimg, pv, px;
  {
    const int p0 = 32 * (T0 - D) + col + PB;
    pimg = p0 / PB - 1;
    const int rem = p0 % PB;
    pv = rem / W1;
    px = rem % W1;
  }
  int cnext = T0 - D;
  auto ptab_store = [&]() {
    const bool ok = pimg >= 0 && pimg < g.n && pv < H && px < W;
    const int val = ok ? (pimg * H + pv) * W + px : -1;
    if (wave == 0 && kg == 0) ptab[(cnext & 15) * 32 + col] = val;
    ++cnext;
    px += dx32;
    pv += dv32;
    if (px >= W1) {
      px -= W1;
      pv += 1;
    }
    if (pv >= H1) {
      pv -= H1;
      pimg += 1;
    }
  };

  // staging: units 0,1 = the X chunk, units 2,3 = the dY chunk; float4 unit f = tid + k * 256 of a [32 positions][64] block
  int soff[2], spos[2], sc4[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int f = tid + k * 256;
    spos[k] = f / 16;
    sc4[k] = (f % 16) * 4;
    soff[k] = spos[k] * S + (f % 16) * 8;
  }
  // raw buffer loads with 32-bit byte offsets: pixel index -1 (a zero position) is out of range and reads as 0
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)g.X, 0, (int)g.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)g.dY, 0, (int)g.ybytes, 0x00020000);
  const unsigned xrow = g.Cin * 4u, yrow = g.Cout * 4u;
  const unsigned xcol[2] = {(ci0 + sc4[0]) * 4u, (ci0 + sc4[1]) * 4u}, ycol[2] = {(co0 + sc4[0]) * 4u, (co0 + sc4[1]) * 4u};
  f32x4 raw[4];
  auto unit_pix = [&](int chunk, int k) { return ptab[(chunk & 15) * 32 + spos[k & 1]]; };
  auto load_x = [&](int pix, int k) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (unsigned)pix * xrow + xcol[k & 1], 0, 0));
  };
  auto load_unit_at = [&](int pix, int k) {
    raw[k] = k < 2 ? load_x(pix, k)
                   : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsD, (unsigned)pix * yrow + ycol[k & 1], 0, 0));
  };
  const float islope = g.islope, sxn = sx * g.islope, sy11 = sy * 2048.f;
  f32x4 dysum[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};  // this thread's 4 channels of dY, summed over its positions
  auto write_vals = [&](char* base, int k, const f32x4& v, float mine = 1.f) {  // base: first slot of the chunk (ring) / buffer (dY)
    float r0 = v[0], r1 = v[1], r2 = v[2], r3 = v[3];
    char* d = base + soff[k & 1];
    if (k >= 2) {  // mine = 0: the tile behind this workgroup's range (staged by the last iteration, summed by its owner)
#pragma unroll
      for (int e = 0; e < 4; ++e) dysum[k & 1][e] = fmaf(v[e], mine, dysum[k & 1][e]);
    }
    if constexpr (NP == 3) {
      if (k < 2) {  // X: the activation of the layer that produced it, applied on the way into LDS
        r0 = r0 > 0.f ? r0 : r0 * islope;
        r1 = r1 > 0.f ? r1 : r1 * islope;
        r2 = r2 > 0.f ? r2 : r2 * islope;
        r3 = r3 > 0.f ? r3 : r3 * islope;
      }
      unsigned a0, a1, a2, b0, b1, b2;
      bf3_split(r0, r1, a0, a1, a2);
      bf3_split(r2, r3, b0, b1, b2);
      *reinterpret_cast<u32x2*>(d) = u32x2{a0, b0};
      *reinterpret_cast<u32x2*>(d + 128) = u32x2{a1, b1};
      *reinterpret_cast<u32x2*>(d + 256) = u32x2{a2, b2};
    } else if (k < 2) {  // X: (xh, xl), the scale on the activation's two slopes
      unsigned a0, a1, b0, b1;
      f16_split_sv(r0, r1, r0 > 0.f ? sx : sxn, r1 > 0.f ? sx : sxn, a0, a1);
      f16_split_sv(r2, r3, r2 > 0.f ? sx : sxn, r3 > 0.f ? sx : sxn, b0, b1);
      *reinterpret_cast<u32x2*>(d) = u32x2{a0, b0};
      *reinterpret_cast<u32x2*>(d + 128) = u32x2{a1, b1};
    } else {  // dY: planes (yh, yH, yl)
      unsigned h01, H01, l01, h23, H23, l23;
      f16_split3_su(r0, r1, sy, sy11, h01, H01, l01);
      f16_split3_su(r2, r3, sy, sy11, h23, H23, l23);
      *reinterpret_cast<u32x2*>(d) = u32x2{h01, h23};
      *reinterpret_cast<u32x2*>(d + 128) = u32x2{H01, H23};
      *reinterpret_cast<u32x2*>(d + 256) = u32x2{l01, l23};
    }
  };
  auto write_unit = [&](char* base, int k, float mine = 1.f) { write_vals(base, k, raw[k], mine); };
  auto chunk_slot = [&](int ch) { return (((ch % NCH) + NCH) % NCH) * 32 * S; };

  // ---- prologue -------------------------------------------------------------------------------------------------------------
  if (NT > 0) {
    for (int ch = T0 - D; ch <= T0 + D + 2; ++ch) ptab_store();
  }
  __syncthreads();
  if (NT > 0) {  // all start-up loads in flight at once
    f32x4 praw[7][2];
#pragma unroll
    for (int u = 0; u < 7; ++u)
      if (u < 2 * D + 1) {
#pragma unroll
        for (int k = 0; k < 2; ++k) praw[u][k] = load_x(unit_pix(T0 - D + u, k), k);
      }
#pragma unroll
    for (int k = 2; k < 4; ++k) load_unit_at(unit_pix(T0, k), k);
#pragma unroll
    for (int u = 0; u < 7; ++u)
      if (u < 2 * D + 1) {
#pragma unroll
        for (int k = 0; k < 2; ++k) write_vals(ringp + chunk_slot(T0 - D + u), k, praw[u][k]);
      }
#pragma unroll
    for (int k = 2; k < 4; ++k) write_unit(dyp, k);  // tile T0 -> buffer 0
#pragma unroll
    for (int k = 0; k < 2; ++k) load_unit_at(unit_pix(T0 + D + 1, k), k);
#pragma unroll
    for (int k = 2; k < 4; ++k) load_unit_at(unit_pix(T0 + 1, k), k);
  }
  __syncthreads();

  // transposing-read geometry (imgwgrad_kernel): 16-lane group gq reads [4 positions][16 channels]; lane lp supplies position
  // (lp >> 2), channels 4 (lp & 3) .. +3 and receives channel lp of the block, positions 0..3
  const int gq = lane >> 4, lp = lane & 15;
  const int kbase = 8 * (gq >> 1) + (lp >> 2);
  const int cch = 16 * (gq & 1) + 4 * (lp & 3);
  const int a_ch = (h * 32 + cch) * 2, b_ch = (c * 32 + cch) * 2;
  int boff[2][2];  // dY fragment addresses inside a buffer: [k-step][4-position half]
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t) boff[s][t] = (s * 16 + kbase + 4 * t) * S + b_ch;

  f32x16 acc[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) acc[j] = f32x16{0};

  int tb = (int)(((long long)32 * T0) % RING) * S;  // first slot of tile T, in bytes
  int wch = (((T0 + D + 1) % NCH) + NCH) % NCH;

  // fragment addresses of one kernel row (dy) of k-step s of the tile whose first slot is tbv: [dx][4-position half]
  // (all in BYTES: slot * S never multiplied in the loop)
  const unsigned RINGB = (unsigned)RING * S;
  const int laneB = kbase * S + (int)RINGB;
  auto row_addr = [&](int (&ad)[3][2], int tbv, int s, int dy) {  // tbv: first slot of the tile, in bytes
    unsigned q = (unsigned)(tbv + laneB + (s * 16 + dy * W1 - 1) * S);
    q = min(q, q - RINGB);
    q = min(q, q - RINGB);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        unsigned r = q + (dx + 4 * t) * S;
        r = min(r, r - RINGB);
        ad[dx][t] = (int)r + a_ch;
      }
  };
  auto read_row = [&](frag (&A)[3][NPA], const int (&ad)[3][2]) {
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
      for (int pc = 0; pc < NPA; ++pc) A[dx][pc] = c3_tr_pair<frag>(ringp + ad[dx][0] + pc * 128, ringp + ad[dx][1] + pc * 128);
  };
  auto read_b = [&](frag (&B)[2][NPB], const char* buf) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int pc = 0; pc < NPB; ++pc) B[s][pc] = c3_tr_pair<frag>(buf + boff[s][0] + pc * 128, buf + boff[s][1] + pc * 128);
  };

  frag A_cur[3][NPA], Bf[2][NPB];
  int ad[3][2];
  if (NT > 0) {
    row_addr(ad, tb, 0, -1);
    read_row(A_cur, ad);
    read_b(Bf, dyp);
  }
  for (int t = 0; t < NT; ++t) {
    const int Tt = T0 + t;
    const char* const nbuf_r = dyp + ((t + 1) & 1) * 32 * S;
    char* const nbuf_w = dyp + ((t + 1) & 1) * 32 * S;
    char* const xw = ringp + wch * 32 * S;
    int tbn = tb + 32 * S;
    tbn = tbn >= (int)RINGB ? tbn - (int)RINGB : tbn;
    int upix[4];
#pragma unroll
    for (int step = 0; step < 6; ++step) {  // (k-step, kernel row)
      const int s = step / 3, grp = step % 3;
      frag A_nxt[3][NPA], B_nxt[2][NPB];
      if (step < 5) {
        row_addr(ad, tb, (step + 1) / 3, (step + 1) % 3 - 1);
        read_row(A_nxt, ad);
      } else {  // behind the barrier (end of step 4): the first fragments of the NEXT tile, covered by this step's 18 MFMAs
        row_addr(ad, tbn, 0, -1);
        read_row(A_nxt, ad);
        read_b(B_nxt, nbuf_r);
      }
      if (step == 0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) upix[k] = unit_pix(Tt + D + 2, k);
#pragma unroll
        for (int k = 2; k < 4; ++k) upix[k] = unit_pix(Tt + 2, k);
      }
      if (step == 1) ptab_store();  // table row of chunk Tt + D + 3
      if (step >= 1 && step <= 4) {  // conversion of X chunk Tt+D+1 / dY tile Tt+1 (loaded one tile ago), next loads
        const int k = step - 1;
        write_unit(k < 2 ? xw : nbuf_w, k, t + 1 < NT ? 1.f : 0.f);
        load_unit_at(upix[k], k);
      }
      if constexpr (NP == 3) {
        constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB_[6] = {2, 1, 0, 1, 0, 0};  // small terms first
#pragma unroll
        for (int m = 0; m < 6; ++m)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
            acc[grp * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_cur[dx][PA[m]], Bf[s][PB_[m]], acc[grp * 3 + dx], 0, 0, 0);
      } else {
        constexpr int PA[3] = {1, 0, 0}, PB_[3] = {0, 2, 1};  // xl yh, xh yl, xh yH
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
            acc[grp * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_cur[dx][PA[m]], Bf[s][PB_[m]], acc[grp * 3 + dx], 0, 0, 0);
      }
      // the "1 MFMA, N others" directive pays in the bf16 form only: the fp16 form (half the MFMAs per step) measured 174-182 us
      // with it and 158-166 us with hipcc's own order (64 x 64 @64x64, n = 128, incl. the finish); the bf16 form 214 vs 231
      if (MVK_C3W_SCHED > 0 && NP == 3) {
#pragma unroll
        for (int m = 0; m < 6 * NP; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x496, MVK_C3W_SCHED > 0 ? MVK_C3W_SCHED : 1, 0);
        }
      }
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int pc = 0; pc < NPA; ++pc) A_cur[dx][pc] = A_nxt[dx][pc];
      if (step == 4) __syncthreads();  // the conversions of this tile (steps 1-4) are published; ONE barrier per tile
      if (step == 5) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int pc = 0; pc < NPB; ++pc) Bf[s2][pc] = B_nxt[s2][pc];
      }
    }
    tb = tbn;
    wch = wch + 1 == NCH ? 0 : wch + 1;
  }

  // partial gradient of this workgroup: slab[worker][(tap * Cin + ci)][co]
  float* const slab = g.slab + (long long)worker * 9 * g.Cin * g.Cout;
  const float out_a = NP == 2 ? f16_inv_scale(sx) : 1.f, out_b = NP == 2 ? f16_inv_scale(sy) * (1.f / 2048.f) * g.dy_scale : g.dy_scale;
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = ci0 + h * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
      slab[((long long)j * g.Cin + ci) * g.Cout + co0 + c * 32 + col] = NP == 2 ? acc[j][r] * out_a * out_b : acc[j][r] * out_b;
    }
  if (g.dbpart && ci0 == 0) {  // bias gradient: column sums of dY, added in a fixed order (thread groups of equal channels)
    float* const red = reinterpret_cast<float*>(lds);  // the ring is dead behind the last barrier of the loop
    __syncthreads();
    *reinterpret_cast<f32x4*>(red + tid * 4) = dysum[0] + dysum[1];
    __syncthreads();
    if (tid < 64) {
      float sum = 0.f;
      for (int j = 0; j < 16; ++j) sum += red[(j * 16 + (tid >> 2)) * 4 + (tid & 3)];
      g.dbpart[(long long)worker * g.Cout + co0 + tid] = sum * g.dy_scale;
    }
  }
  mvk_prof_end(g.prof);
}

static int c3_ring(int W);
bool c3rs_wgrad_ok(int n, int H, int W, int Cin, int Cout) {
  if (n <= 0 || H < 4 || W < 4 || W > 96 || Cin % 64 != 0 || Cout % 64 != 0 || Cin > 256 || Cout > 256) return false;
  const int types = (Cin / 64) * (Cout / 64);
  if (types > 16) return false;
  const long long total = (long long)n * (H + 1) * (W + 1);
  if (total >= (1ll << 31) - 64 || (long long)n * H * W * (Cin > Cout ? Cin : Cout) * 4 >= (1ll << 32) - 8192) return false;
  const int D = (W + 2 + 31) / 32;
  if ((H + 1) * (W + 1) < 32 * D || 32 / (W + 1) + 1 > H + 1 || D > 3) return false;
  return c3w_lds_bytes(c3_ring(W)) <= 160 * 1024;
}

// slab: [*nz][9 Cin][Cout] partial gradients (needs (256 / types) * 9 Cin Cout floats); 1: not covered
int c3rs_wgrad(const float* X, const float* dY, float* slab, long long slab_floats, float* dbpart, int x_act, float dy_scale,
               int n, int H, int W, int Cin, int Cout, int* nz, const float* x_amax, const float* y_amax, hipStream_t s) {
  if (x_act == MVK_ACT_SIGMOID) return 1;
  if (!c3rs_wgrad_ok(n, H, W, Cin, Cout) || !mvk_aligned16(X) || !mvk_aligned16(dY)) return 1;
  const int types = (Cin / 64) * (Cout / 64);
  const int grid = 256 / types * types, workers = grid / types;
  if (slab_floats < (long long)workers * 9 * Cin * Cout) return 1;
  const int ring = c3_ring(W), lds = c3w_lds_bytes(ring);
  static bool attr_done = false;
  if (!attr_done) {
    for (const void* f : {reinterpret_cast<const void*>(c3wg_kernel<3>), reinterpret_cast<const void*>(c3wg_kernel<2>)})
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return MVK_ELAUNCH;
    attr_done = true;
  }
  const long long total = (long long)n * (H + 1) * (W + 1);
  C3WArgs a{X, dY, slab, dbpart, c3_slope(x_act), dy_scale, n, H, W, Cin, Cout, ring, (W + 2 + 31) / 32, (int)((total + 31) / 32),
            (unsigned)((long long)n * H * W * Cin * 4), (unsigned)((long long)n * H * W * Cout * 4), nullptr, x_amax, y_amax};
  a.prof = prof_next(8, 2.0 * n * H * W * 9.0 * Cin * Cout);
  *nz = workers;
  if (x_amax && y_amax) hipLaunchKernelGGL(c3wg_kernel<2>, dim3(grid), dim3(256), lds, s, a);
  else hipLaunchKernelGGL(c3wg_kernel<3>, dim3(grid), dim3(256), lds, s, a);
  MVK_CHECK_LAUNCH();
  prof_fold(a.prof, s);
  return MVK_OK;
}

static int c3_lds_bytes(int Cin, int Cout, int ring, int np) {
  if (np == 2) {
    if (Cin == 64 && Cout == 64) return C3Cfg<64, 64, 2>::lds_bytes(ring);
    if (Cin == 64 && Cout == 128) return C3Cfg<64, 128, 2>::lds_bytes(ring);
    if (Cin == 128 && Cout == 64) return C3Cfg<128, 64, 2>::lds_bytes(ring);
    if (Cin == 128 && Cout == 128) return C3Cfg<128, 128, 2>::lds_bytes(ring);
    if (Cin == 128 && Cout == 256) return C3Cfg<128, 256, 2>::lds_bytes(ring);
    return 1 << 30;
  }
  if (Cin == 64 && Cout == 64) return C3Cfg<64, 64>::lds_bytes(ring);
  if (Cin == 64 && Cout == 128) return C3Cfg<64, 128>::lds_bytes(ring);
  if (Cin == 128 && Cout == 64) return C3Cfg<128, 64>::lds_bytes(ring);
  if (Cin == 128 && Cout == 128) return C3Cfg<128, 128>::lds_bytes(ring);
  if (Cin == 128 && Cout == 256) return C3Cfg<128, 256>::lds_bytes(ring);
  return 1 << 30;
}

static int c3_ring(int W) {
  const int D = (W + 2 + 31) / 32;
  int ring = 32 * (2 * D + 2);
  while (ring <= 32 * D + 63 + W + 2) ring += 32;  // chunk T+D+1 must not land on a slot tile T still reads
  return ring;
}

// every condition under which c3rs_conv takes a problem (the caller reserves arena space only behind this test)
// np: 3 = bf16 pieces, 2 = scaled fp16 pieces (smaller ring slots: 128 input channels reach maps up to 62 wide)
bool c3rs_shape_ok(int n, int H, int W, int Cin, int Cout, int np) {
  if (n <= 0 || H < 4 || W < 4 || W > 96) return false;
  const long long total = (long long)n * (H + 1) * (W + 1);
  // 32-bit byte offsets into X / Y; offset (unsigned)(-1) * 4 C must stay out of range
  if (total >= (1ll << 31) - 64 || (long long)n * H * W * (Cin > Cout ? Cin : Cout) * 4 >= (1ll << 32) - 8192) return false;
  const int D = (W + 2 + 31) / 32;
  if ((H + 1) * (W + 1) < 32 * D) return false;   // the table's start-up shift by one image block
  if (D > (Cin == 64 ? 3 : (np == 2 ? 2 : 1))) return false;  // C3Cfg::MAXD
  if (32 / (W + 1) + 1 > H + 1) return false;     // one wrap per 32-position step of the incremental (image, row, column)
  return c3_lds_bytes(Cin, Cout, c3_ring(W), np) <= 160 * 1024;
}

// 1: not covered (the caller falls back to the implicit-GEMM engine).  colsum_part: [256 / types][Cout] floats.
int c3rs_conv(const float* X, const float* Wp, const float* bias, float* Y, int n, int H, int W, int Cin, int Cout, int act,
              const float* act_src, int src_act, const float* res, float res_alpha, float* colsum_part, int* part_rows,
              int x_act, float pre_scale, const float* x_amax, const float* w_amax, float* y_amax, hipStream_t s, float* y_pre,
              int x_channels, int w_channels, int res_pre) {
  // x_channels / w_channels (0 = Cin): channels per pixel of the tensor X is a slice of / input channels of the layer Wp belongs to
  if (act == MVK_ACT_SIGMOID || (act_src && src_act == MVK_ACT_SIGMOID) || x_act == MVK_ACT_SIGMOID || !mvk_aligned16(X)) return 1;
  if (colsum_part && (!act_src || res)) return 1;  // column sums: backward-data form only (WITH_CSUM)
  const bool f16 = x_amax && w_amax;
  if (!c3rs_shape_ok(n, H, W, Cin, Cout, f16 ? 2 : 3)) return 1;
  const long long total = (long long)n * (H + 1) * (W + 1);
  const int xc = x_channels > 0 ? x_channels : Cin, wc = w_channels > 0 ? w_channels : Cin;
  if (xc < Cin || wc < Cin || (res_pre && !res) || (long long)n * H * W * xc * 4 >= (1ll << 32) - 8192) return 1;
  C3Args a{X, Wp, bias, Y, act_src, res, colsum_part, (unsigned)(((long long)n * H * W - 1) * xc * 4 + (long long)Cin * 4),
           (unsigned)((long long)n * H * W * Cout * 4), n, H, W, c3_slope(act), c3_slope(act_src ? src_act : MVK_ACT_NONE),
           res_alpha, c3_slope(x_act), pre_scale, c3_ring(W), (W + 2 + 31) / 32, (int)((total + 31) / 32), nullptr,
           x_amax, w_amax, y_amax, y_pre, (unsigned)(xc * 4), wc, res_pre ? 1.f : 0.f};
  if (y_pre && !f16) return 1;
  if (f16) {
    if (Cin == 64 && Cout == 64) return c3rs_launch<64, 64, 2>(a, part_rows, s);
#ifndef MVK_C3_PROBE_ONLY  // variant builds of tools/conv3_variants.sh: one instantiation, short compiles
    if (Cin == 64 && Cout == 128) return c3rs_launch<64, 128, 2>(a, part_rows, s);
    if (Cin == 128 && Cout == 64) return c3rs_launch<128, 64, 2>(a, part_rows, s);
    if (Cin == 128 && Cout == 128) return c3rs_launch<128, 128, 2>(a, part_rows, s);
    if (Cin == 128 && Cout == 256) return c3rs_launch<128, 256, 2>(a, part_rows, s);
#endif
    return 1;
  }
  if (Cin == 64 && Cout == 64) return c3rs_launch<64, 64, 3>(a, part_rows, s);
#ifndef MVK_C3_PROBE_ONLY
  if (Cin == 64 && Cout == 128) return c3rs_launch<64, 128, 3>(a, part_rows, s);
  if (Cin == 128 && Cout == 64) return c3rs_launch<128, 64, 3>(a, part_rows, s);
  if (Cin == 128 && Cout == 128) return c3rs_launch<128, 128, 3>(a, part_rows, s);
  if (Cin == 128 && Cout == 256) return c3rs_launch<128, 256, 3>(a, part_rows, s);
#endif
  return 1;
}

#ifdef MVK_C3PROF
extern "C" int mvk_c3_debug_buffer(unsigned long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_c3_dbg), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

}  // namespace mvk
