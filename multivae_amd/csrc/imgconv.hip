// Register-stationary-weight convolution kernels for the 4x4 / stride-2 / pad-1 layer pairs of the SVHN networks
// (reference: models/nn/svhn.py:19-28 Conv2d stack, :52-60 ConvTranspose2d stack).
//
// The implicit-GEMM engine (igemm_bf.hpp) re-gathers every activation element 16 times (once per kernel tap) from
// L2 and re-splits it into bf16 pieces each time: its big launches are bound by the vector-memory path, not by the
// matrix cores.  These kernels turn the data flow round for gfx950's 512-register unified
// VGPR/AGPR file:
//
//   * one workgroup per CU, 4 waves, ONE wave per SIMD, up to 512 registers per lane;
//   * WEIGHTS ARE STATIONARY IN REGISTERS: every wave loads its [256 k][32 n] slice of the layer's GEMM weight
//     once per launch, splits it into the three bf16 pieces (x = x0 + x1 + x2, igemm_bf.hpp) and keeps the
//     16 k-steps x 3 pieces x 4 registers = 192 VGPRs as MFMA B operands for the whole launch.  4 waves x 48 KB
//     hold the whole weight of the 64<->32 channel layer pair (196 KB as pieces: more than the 160 KB of LDS); the
//     128<->64 pair (786 KB) takes 4 workgroup types;
//   * IMAGES STREAM THROUGH LDS: each activation image is read from HBM/L2 once per workgroup type with coalesced
//     16-byte loads, split once, and written as three bf16 planes [pixel][channel] (padded rows, one zero row for
//     the halo); every (tap, class) use is a ds_read_b128 A fragment from LDS.  Double buffered; the conversion of
//     the next unit is spread over the k-steps of the current one;
//   * a product is the 6 piece products of order <= 2 on v_mfma_f32_32x32x16_bf16, alternating between two
//     accumulators (two independent chains per wave);
//   * waves that split the reduction dimension exchange accumulator slices through LDS (each wave finishes a
//     slice of the 32 x 32 tile): one s_barrier per tile;
//   * the epilogue adds the bias, applies the activation / the activation derivative of the layer the result
//     lands in, stores whole 128-byte rows and keeps per-column sums (bias gradient) in registers; per-workgroup
//     partials are reduced in a fixed order (deterministic).
//
// Per 32-row tile a wave issues 96 MFMAs (3072 matrix-pipe cycles) against 48 ds_read_b128, <= 8 global loads and
// ~100-200 VALU: the loop is MFMA-bound by construction.
#include <cstdlib>
#include <type_traits>

#include "bf3.hpp"

#ifndef MVK_IC_ABL
#define MVK_IC_ABL 0  // subtraction builds (tools/ab_probe.sh): 1 = every input load from the worker's first unit (L2-resident), 2 = every store and mask load on the first unit
#endif
#ifdef MVK_NO_RFL
#define MVK_RFL(x) (x)
#else
#define MVK_RFL(x) __builtin_amdgcn_readfirstlane(x)
#endif

#ifndef MVK_IC_SCHED
#define MVK_IC_SCHED (HS == 8 ? 5 : 4)  // "others" per MFMA of the scheduling pipeline in imgconv_kernel (0 = hipcc's own order)
#endif

#ifndef MVK_IC_SCHED2
#define MVK_IC_SCHED2 2  // multiplier of MVK_IC_SCHED in the scaled-fp16 form (half the MFMAs per pair of k-steps)
#endif

namespace mvk {

#ifdef MVK_ICPROF  // tools/imgconv_phase.py: per-wave cycle counters (total, waiting at the tile barrier, k-loops)
__device__ unsigned long long* g_ic_dbg = nullptr;
#define IC_CLK() __builtin_readcyclecounter()
#endif

enum { IC_UP = 0, IC_DOWN = 1 };

struct ImgConvArgs {
  const float* A;        // input images [n][APIX][CIN] (NHWC)
  const float* Wp;       // fp32 GEMM pack: Wup[class][(a,b,cv)][cu] or Wdown[(kh,kw,cu)][cv]
  const void* wfrag;     // the same weights as bf16-piece fragments [role][k-step][piece][lane][8] (mvk_pack_weights), or null
  const float* bias;     // [COUT] or null
  float* out;            // output images (NHWC)
  const float* act_src;  // tensor of the output's shape whose activation derivative multiplies the result, or null
  float* colsum_part;    // [rows][COUT] per-workgroup column sums of the stored result, or null
  int n;                 // images
  int act, src_act;
  mvk_prof_slot* prof;   // device-timestamp record of this launch (null: profiler off)
  // scaled-fp16 form (NP = 2, bf3.hpp): device scalars bounding max |A| and max |Wp| (both required); either form: max |out|
  // is published into y_amax (atomic max; must hold 0 before the launch) when it is given
  const float* x_amax;
  const float* w_amax;
  float* y_amax;
};

// NP = pieces per operand element: 3 bf16 pieces (6 MFMAs per product) or 2 scaled fp16 pieces (3 MFMAs per product)
template <int KIND, int HS, int CIN, int COUT, int NP = 3>
struct ICfg {
  static constexpr int PIX = HS * HS;                          // output rows per image (per parity class for UP)
  static constexpr int APIX = KIND == IC_UP ? PIX : 4 * PIX;   // input pixels per image
  static constexpr int AW = KIND == IC_UP ? HS : 2 * HS;       // input image width
  static constexpr int CHUNKS = CIN / 16;                      // 16-channel k-steps per tap
  static constexpr int NTAPS = 16 / CHUNKS;                    // taps per wave (16 k-steps = 192 B registers)
  static constexpr int TAPS_ALL = KIND == IC_UP ? 4 : 16;      // taps of one output element
  static constexpr int KSPLIT = TAPS_ALL / NTAPS;              // waves sharing one output tile
  static constexpr int NCT = COUT / 32;                        // 32-column tiles
  static constexpr int ROLES = (KIND == IC_UP ? 4 : 1) * NCT * KSPLIT;
  static constexpr int WG_TYPES = ROLES / 4;
  static constexpr int SU = PIX >= 32 ? 1 : 32 / PIX;          // images per stage unit
  static constexpr int TPU = PIX >= 32 ? PIX / 32 : 1;         // 32-row tiles per stage unit (and class)
  static constexpr int AROWS = SU * APIX;                      // LDS rows per unit
  static constexpr int S = CIN * 2 + 16;                       // bytes per LDS row (bf16 channels + pad)
  // Pixel (img, y, x) of a unit sits at img * IMGB + y * LINEB + xrow(x) * S.  An A fragment (ds_read_b128, four groups of
  // 16 lanes, each reading 256 bytes = 16 slots of 16 B per LDS cycle) takes the pixels of 32 consecutive outputs at one
  // tap: for UP that is a stride of one pixel per output and plain rows are conflict-free; for DOWN the stride is two
  // pixels (2 S = 160 or 288 bytes: only even slots / every other slot pair are hit -> 2-way for 16 -> 8 output columns,
  // 4-way for 8 -> 4).  Storing the even and the odd input columns of a line in separate half lines (16 -> 8) makes the
  // stride one row again, and LPAD / IPAD shift the lines (and the second image of a 2-image unit) onto the slots the
  // first line of a lane group leaves free: every group then covers 16 distinct slots (checked exhaustively offline).
  static constexpr bool XPERM = KIND == IC_DOWN && HS == 8;
  static constexpr int LPAD = KIND == IC_DOWN ? 64 : 0;
  static constexpr int IPAD = (KIND == IC_DOWN && HS == 4) ? 16 : 0;
  static constexpr int LINEB = AW * S + LPAD;
  static constexpr int IMGB = AW * LINEB + IPAD;
  static constexpr int ZOFF = SU * IMGB;                       // the zero row (halo)
  static constexpr int PLANE = SU * IMGB + S;
  __host__ __device__ static constexpr int pix_off(int img, int y, int x) {
    return img * IMGB + y * LINEB + (XPERM ? (x & 1) * (AW / 2) + (x >> 1) : x) * S;
  }
  static constexpr int BUF = NP * PLANE;
  static constexpr int NF4 = AROWS * CIN / 4 / 256;            // float4 units per thread per stage unit
  static constexpr int OWN = 16 / KSPLIT;                      // accumulator registers a wave finishes
  static constexpr int XWAVE = (KSPLIT - 1) * OWN * 64 * 4;    // exchange bytes received per wave
  static constexpr int XBUF = KSPLIT > 1 ? 4 * XWAVE : 0;
  static constexpr int LDS_BYTES = 2 * BUF + 2 * XBUF + 4 * 32 * 4;
  static_assert(CIN % 16 == 0 && 16 % CHUNKS == 0 && TAPS_ALL % NTAPS == 0, "k split");
  static_assert(ROLES % 4 == 0 && COUT % 32 == 0, "roles");
  static_assert((AROWS * CIN / 4) % 256 == 0, "stage unit must split evenly over 256 threads");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

__device__ __forceinline__ bf16x8 ic_pack8(const unsigned (&d)[4]) {
  u32x4 v = {d[0], d[1], d[2], d[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int KIND, int HS, int CIN, int COUT, bool HAS_SRC, bool PIPE, int NP = 3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void imgconv_kernel(const ImgConvArgs g) {
  using T = ICfg<KIND, HS, CIN, COUT, NP>;
  static_assert(NP == 3 || PIPE, "the scaled-fp16 form exists for the two-tile-latency loop only");
  using frag = std::conditional_t<NP == 3, bf16x8, f16x8>;
  // scaled-fp16 form: the operands are multiplied by sx / sw on their way into pieces, the result by 1 / (sx sw)
  const float sx = NP == 2 ? f16_scale_of(*g.x_amax) : 1.f, sw = NP == 2 ? f16_scale_of(*g.w_amax) : 1.f;
  const float inv_s = NP == 2 ? f16_inv_scale(sx) * f16_inv_scale(sw) : 1.f;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  mvk_prof_begin(g.prof);
#ifdef MVK_ICPROF
  const unsigned long long ic_t0 = IC_CLK();
  unsigned long long ic_bar = 0, ic_k = 0, ic_pre = 0, ic_post = 0, ic_tend = 0;
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, kg = lane >> 5;
  // Workgroup types of one worker read the SAME images (each holds another slice of the weights).  Block b runs on XCD b % 8
  // (observed placement; a speed choice only), and every XCD has its own L2: with type = b % WG_TYPES the four readers of an
  // image sat on four XCDs and the input crossed the fabric 2.0-2.5 times (profiles/r04_pmc_hbm.md) in launches that move
  // 250-420 MB in 60-80 us, i.e. run at the HBM rate.  Types of a worker on ONE XCD: slot = b / 8 on XCD b % 8, type = slot %
  // WG_TYPES, so the four workgroups stream the same lines through one L2 at the same time.  Same unit ranges per worker.
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

  // ---- role of this wave ----------------------------------------------------------------------------------
  int cls = 0, ct = 0, ks = 0;
  if (KIND == IC_UP) {
    const int role = wgtype * 4 + wave;  // (class, ks, ct)
    cls = role / (T::KSPLIT * T::NCT);
    ks = (role / T::NCT) % T::KSPLIT;
    ct = role % T::NCT;
  } else if (T::WG_TYPES == 1) {
    ks = wave / T::NCT;
    ct = wave % T::NCT;
  } else {  // one column tile per workgroup type, the waves split the taps
    ct = wgtype;
    ks = wave;
  }
  const int py = cls >> 1, px = cls & 1;
  const int ncol = ct * 32 + col;

  // ---- weights: 16 k-steps x 3 pieces, resident for the whole launch ---------------------------------------------
  frag Bw[T::NTAPS][T::CHUNKS][NP];
  if (NP == 3 && g.wfrag) {  // 48 coalesced 16-byte loads per lane
    const bf16x8* f = reinterpret_cast<const bf16x8*>(g.wfrag) + (long long)(wgtype * 4 + wave) * 16 * 3 * 64 + lane;
#pragma unroll
    for (int q = 0; q < T::NTAPS; ++q)
#pragma unroll
      for (int c = 0; c < T::CHUNKS; ++c)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) Bw[q][c][pc < NP ? pc : 0] = __builtin_bit_cast(frag, f[((q * T::CHUNKS + c) * 3 + pc) * 64]);
  } else
#pragma unroll
  for (int q = 0; q < T::NTAPS; ++q) {
    const int tap = ks * T::NTAPS + q;
    const long long rowbase = KIND == IC_UP ? (long long)(cls * 4 + tap) * CIN : (long long)tap * CIN;
#pragma unroll
    for (int c = 0; c < T::CHUNKS; ++c) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = g.Wp[(rowbase + c * 16 + kg * 8 + e) * COUT + ncol];
      unsigned p[3][4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (NP == 3) bf3_split(v[2 * e], v[2 * e + 1], p[0][e], p[1][e], p[2][e]);
        else f16_split(v[2 * e] * sw, v[2 * e + 1] * sw, p[0][e], p[1][e]);
      }
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) Bw[q][c][pc] = __builtin_bit_cast(frag, u32x4{p[pc][0], p[pc][1], p[pc][2], p[pc][3]});
    }
  }

  // The 192 weight registers live in the ACCUMULATOR half of the register file: gfx950's MFMA takes its A / B operands from
  // AGPRs as well as VGPRs, and an empty asm with an "=a" output makes the value an AGPR citizen for hipcc (its MFMA
  // builtin then reads a[...] directly).  Without this the kernel needs ~330 architectural VGPRs, hipcc parks ~100 of
  // the weight registers in AGPRs as SPILLS and pays one v_accvgpr_read per MFMA to bring them back (203 reads per 192
  // MFMAs in the 32 -> 64 channel kernel) — in a loop whose budget is ~5 non-MFMA instructions per MFMA.
#pragma unroll
  for (int q = 0; q < T::NTAPS; ++q)
#pragma unroll
    for (int c = 0; c < T::CHUNKS; ++c)
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) {
        frag t = Bw[q][c][pc];
        asm volatile("" : "=a"(Bw[q][c][pc]) : "0"(t));
      }

  // ---- LDS geometry -----------------------------------------------------------------------------------------------
  char* const xbase = lds + 2 * T::BUF;
  float* const csred = reinterpret_cast<float*>(lds + 2 * T::BUF + 2 * T::XBUF);
  // zero rows (at ZOFF of every plane of both buffers)
  for (int i = tid; i < 2 * NP * (T::S / 4); i += 256) {
    const int pl = i / (T::S / 4), w = i % (T::S / 4);
    *reinterpret_cast<unsigned*>(lds + (pl / NP) * T::BUF + (pl % NP) * T::PLANE + T::ZOFF + w * 4) = 0u;
  }
  // A-fragment byte offsets of this lane, per (tile of the unit, tap of this wave): row * S + kg * 16
  int aoff[T::TPU][T::NTAPS];
#pragma unroll
  for (int tt = 0; tt < T::TPU; ++tt) {
    const int gr = tt * 32 + col;
    const int img = gr / T::PIX, p = gr % T::PIX, i = p / HS, j = p % HS;
#pragma unroll
    for (int q = 0; q < T::NTAPS; ++q) {
      const int tap = ks * T::NTAPS + q;
      int y, x;
      if (KIND == IC_UP) {
        y = i + py - (tap >> 1);
        x = j + px - (tap & 1);
      } else {
        y = 2 * i - 1 + (tap >> 2);
        x = 2 * j - 1 + (tap & 3);
      }
      const bool ok = y >= 0 && y < T::AW && x >= 0 && x < T::AW;
      aoff[tt][q] = (ok ? T::pix_off(img, y, x) : T::ZOFF) + kg * 16;
    }
  }
  // staging: float4 unit f = tid + u*256 of the unit's [AROWS][CIN] matrix -> row f / (CIN/4), 4 channels
  int soff[T::NF4];
#pragma unroll
  for (int u = 0; u < T::NF4; ++u) {
    const int f = tid + u * 256;
    const int r = f / (CIN / 4), img = r / T::APIX, p = r % T::APIX;
    soff[u] = T::pix_off(img, p / T::AW, p % T::AW) + (f % (CIN / 4)) * 8;
  }
  // output rows this wave finishes: accumulator registers [ks*OWN, ks*OWN+OWN).  The element offset of row
  // R = (r & 3) + 8 (r >> 2) + 4 kg of tile tt inside the unit's output block is additive over (tt, r, kg) (no
  // carries between the bit fields for the supported shapes): lane part + compile-time part.
  auto out_off = [&](int tt, int r, int kgv) -> int {
    const int R = (r & 3) + 8 * (r >> 2) + 4 * kgv;
    const int gr = tt * 32 + R;
    const int img = gr / T::PIX, p = gr % T::PIX, i = p / HS, j = p % HS;
    if (KIND == IC_UP) return ((img * 2 * HS + 2 * i) * 2 * HS + 2 * j) * COUT;
    return (img * T::PIX + p) * COUT;
  };
  const int obase = out_off(0, ks * T::OWN, kg) + (KIND == IC_UP ? (py * 2 * HS + px) * COUT : 0) + ncol;
  constexpr long long OUT_UNIT = (long long)T::SU * (KIND == IC_UP ? 4 * T::PIX : T::PIX) * COUT;
  constexpr long long IN_UNIT = (long long)T::AROWS * CIN;

  const float bias = g.bias ? g.bias[ncol] : 0.f;
  const float act_lo = g.act == MVK_ACT_RELU ? 0.f : -__builtin_inff();
  const float src_lo = g.src_act == MVK_ACT_RELU ? 0.f : -__builtin_inff();
  float csum = 0.f, amax_l = 0.f;

  const long long units = g.n / T::SU;
  // readfirstlane: the 64-bit division is emitted on the vector ALU, which makes the trip count of the unit loop 'divergent'
  // for hipcc (exec-masked loop, every live-out accumulator read back per iteration: 32 v_accvgpr_read + a drain of the matrix pipe per tile)
  const long long u0 = MVK_RFL((int)(units * worker / workers));
  const long long u1 = MVK_RFL((int)(units * (worker + 1) / workers));

  f32x4 raw[T::NF4];
  auto unit_src = [&](long long u) {  // clamped: the tail re-reads the last unit instead of branching
    const long long uc = (MVK_IC_ABL & 1) ? u0 : (u < u1 ? u : u1 - 1);
    return reinterpret_cast<const f32x4*>(g.A + uc * IN_UNIT) + tid;
  };
  auto write_f4 = [&](char* buf, int k) {
    if constexpr (NP == 3) {
      unsigned a0, a1, a2, b0, b1, b2;
      bf3_split(raw[k][0], raw[k][1], a0, a1, a2);
      bf3_split(raw[k][2], raw[k][3], b0, b1, b2);
      *reinterpret_cast<u32x2*>(buf + soff[k]) = u32x2{a0, b0};
      *reinterpret_cast<u32x2*>(buf + T::PLANE + soff[k]) = u32x2{a1, b1};
      *reinterpret_cast<u32x2*>(buf + 2 * T::PLANE + soff[k]) = u32x2{a2, b2};
    } else {
      unsigned a0, a1, b0, b1;
      f16_split_su(raw[k][0], raw[k][1], sx, a0, a1);
      f16_split_su(raw[k][2], raw[k][3], sx, b0, b1);
      *reinterpret_cast<u32x2*>(buf + soff[k]) = u32x2{a0, b0};
      *reinterpret_cast<u32x2*>(buf + T::PLANE + soff[k]) = u32x2{a1, b1};
    }
  };

  if (u0 < u1) {
    const f32x4* s0 = unit_src(u0);
#pragma unroll
    for (int k = 0; k < T::NF4; ++k) raw[k] = s0[k * 256];
#pragma unroll
    for (int k = 0; k < T::NF4; ++k) write_f4(lds, k);
    const f32x4* s1 = unit_src(u0 + 1);
#pragma unroll
    for (int k = 0; k < T::NF4; ++k) raw[k] = s1[k * 256];
  }
  __syncthreads();

  // ---- main loop: a software pipeline over 32-row tiles -----------------------------------------------------------
  // Inside the k-loop of tile t (8 pairs of k-steps, 12 MFMAs each) run, in slices between the MFMA groups:
  //   * the A fragments of the NEXT pair (ds_read_b128, one pair ahead),
  //   * the conversion of the next unit into the other LDS buffer + the reload of its registers (first tile of a unit),
  //   * the loads of the activation-derivative source of tile t (consumed one tile later),
  //   * the epilogue of tile t-1 (bias, activation, mask, store, column sums).
  // After the k-loop: accumulator exchange (waves that split the taps), ONE barrier, first fragments of tile t+1.
  auto read_pair = [&](frag (&dst)[2][NP], const char* buf, int tt, int pr) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int q = (2 * pr + h) / T::CHUNKS, c = (2 * pr + h) % T::CHUNKS;
#pragma unroll
      for (int pc = 0; pc < NP; ++pc)
        dst[h][pc] = *reinterpret_cast<const frag*>(buf + aoff[tt][q] + pc * T::PLANE + c * 32);
    }
  };
  float res_prev[T::OWN], msk_prev[T::OWN], msk_cur[T::OWN];
#pragma unroll
  for (int o = 0; o < T::OWN; ++o) res_prev[o] = msk_prev[o] = msk_cur[o] = 0.f;
  auto epilogue_row = [&](float* outp, int ptt, int o, float validf) {
    const int off = obase + out_off(ptt, o, 0);
    float v = fmaxf((NP == 2 ? res_prev[o] * inv_s : res_prev[o]) + bias, act_lo);          // NONE / RELU
    if (HAS_SRC) v = (msk_prev[o] > src_lo) ? v : 0.f;    // x ReLU'(y) of the layer the result lands in (or x 1)
    outp[off] = v;
    csum = fmaf(v, validf, csum);
    amax_l = fmaxf(amax_l, fabsf(v) * validf);
  };
  // A fragments are read ahead of their MFMAs: one pair of k-steps in the bf16 form (12 MFMAs = 384 matrix-pipe cycles cover a
  // ds_read_b128 under load), TWO pairs in the fp16 form, whose pair is 6 MFMAs = 192 cycles — less than the latency of an LDS
  // read when the four waves of the workgroup issue their reads together (the LDS array is busy 8 cycles per b128 wave read,
  // 16 reads per pair).  Subtraction builds (MVK_IC_ABL: all loads and stores on one L2-resident unit) run at 0.9 of the full
  // kernel's time: the loop waits on neither HBM nor the instruction count (r04: -10 % instructions, same time).
#ifndef MVK_IC_DEPTH2
#define MVK_IC_DEPTH2 0  // measured: no gain alone (65-80 us either way), +3 % on the step
#endif
  constexpr bool D2 = NP == 2 && PIPE && MVK_IC_DEPTH2;
  // MVK_IC_DEPTH2 = 2: depth 2 AND the tile barrier one pair earlier (behind pair 5: every fragment of the tile has been read by
  // then), so that the first fragments of the next tile are read behind the barrier with two pairs of MFMAs (384 cycles) in
  // front of their first use instead of one; the conversion of the next unit is spread over pairs 0-5.
  constexpr bool D2E = D2 && MVK_IC_DEPTH2 == 2;
  constexpr int BP = D2E ? 5 : 6;  // the pair whose end holds the barrier
  frag a_cur[2][NP], a_n1[2][NP];
  if (u0 < u1) {
    read_pair(a_cur, lds, 0, 0);
    if (D2) read_pair(a_n1, lds, 0, 1);
  }
  if constexpr (PIPE) {
    // ---- main loop, two-tile latency (default) --------------------------------------------------------------------
    // With one barrier at the END of every tile the waves spend a third of their cycles between k-loops (measured with
    // the in-kernel counters of tools/imgconv_phase.sh: accumulator read-out + exchange writes before the barrier, first
    // fragments + exchange reads behind it, the matrix pipe idle meanwhile).  Here tile T carries, inside its k-loop:
    //   pair 0    : the sum of tile T-1's two accumulator chains and (tap-split kernels) its exchange slices -> LDS
    //   pairs 0-7 : the epilogue slices of tile T-2 and the mask loads of tile T-1
    //   pairs 0-6 : the conversion of the next unit (first tile of a unit)
    //   after pair 6: THE barrier (publishes the exchange slices and the converted unit; the other buffer is free)
    //   pair 7    : the first fragments of tile T+1 and the exchange read of tile T-1 (result of T-1, consumed by T+1)
    // so nothing but the barrier itself stands between the MFMAs of consecutive tiles.  Same sums in the same order as the
    // one-tile-latency loop below (bit-identical results).
    // The main loop is instantiated once per tap-split rank of the wave (KSC): every "is this my slice" test of the
    // exchange is then a compile-time fact.  With a run-time rank hipcc emits exec-masked branches for them (it cannot
    // prove the rank wave-uniform): ~40 tiny basic blocks per tile that no scheduling pipeline crosses.
    auto run = [&](auto ks_tag) {
    constexpr int KSC = decltype(ks_tag)::value;
    // accumulator chains of tile T-1 (NP = 3: one per k-step of a pair; NP = 2: main hi hi' and cross hi lo' + lo hi')
    f32x16 pend0 = {0}, pend1 = {0};
    float own[T::OWN], res_next[T::OWN];
#pragma unroll
    for (int o = 0; o < T::OWN; ++o) own[o] = res_next[o] = 0.f;
    int tcount = 0, xpar = 0;
    constexpr int DU1 = 1, DU2 = T::TPU == 1 ? 2 : 1;  // units back of tiles T-1 (when tt == 0) and T-2
    auto finish_pending = [&](char* xb) {  // pair 0: sum of the chains; exchange slices out, own slice kept
      f32x16 sum;
      if constexpr (NP == 3) {
        sum = pend0 + pend1;
      } else {
        sum = pend0;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum[r] = fmaf(pend1[r], 1.f / 2048.f, sum[r]);
      }
      if (T::KSPLIT > 1) {
#pragma unroll
        for (int r = 0; r < T::KSPLIT; ++r) {
          if (r == KSC) continue;
          const int dwave = T::WG_TYPES == 1 || KIND == IC_UP ? r * T::NCT + ct : r;
          const int slot = KSC < r ? KSC : KSC - 1;
          float* dst = reinterpret_cast<float*>(xb + dwave * T::XWAVE + slot * T::OWN * 256);
#pragma unroll
          for (int o = 0; o < T::OWN; ++o) dst[o * 64 + lane] = sum[r * T::OWN + o];
        }
      }
#pragma unroll
      for (int o = 0; o < T::OWN; ++o) own[o] = sum[KSC * T::OWN + o];
    };
    auto gather_result = [&](const char* xb) {  // pair 7 (behind the barrier): ordered sum over the ranks
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
    for (long long u = u0; u < u1; ++u) {
      const int cur = (int)((u - u0) & 1);
      const char* const abuf = lds + cur * T::BUF;
      char* const nbuf = lds + (cur ^ 1) * T::BUF;
      const f32x4* const src2 = unit_src(u + 2);
      float* const outp_cur = g.out + ((MVK_IC_ABL & 2) ? u0 + 2 : u) * OUT_UNIT;
      const float* const srcp_cur = HAS_SRC ? g.act_src + ((MVK_IC_ABL & 2) ? u0 + 2 : u) * OUT_UNIT : nullptr;
#pragma unroll
      for (int tt = 0; tt < T::TPU; ++tt) {
        const int ptt1 = (tt + T::TPU - 1) % T::TPU;          // tile-in-unit index of tile T-1
        const int ptt2 = (tt + 2 * T::TPU - 2) % T::TPU;      // ... of tile T-2 (== tt for TPU 1 and 2)
        // rows of tile T-2; before there is one: the rows of THIS tile, which its real epilogue overwrites two tiles later
        // (same lane, same address, program order), with weight 0 in the column sums
        const bool valid2 = tcount >= 2;
        float* const outp_e = valid2 ? outp_cur - DU2 * OUT_UNIT : outp_cur;
        const float validf = valid2 ? 1.f : 0.f;
        // mask of tile T-1 (consumed during tile T+1); before there is one: any readable rows
        const float* const srcp_1 = !HAS_SRC ? nullptr : ((tt > 0 || tcount == 0) ? srcp_cur : srcp_cur - DU1 * OUT_UNIT);
        char* const xb = xbase + xpar * T::XBUF;
        const char* const next_buf = (tt + 1 < T::TPU) ? abuf : nbuf;
        const int ntt = (tt + 1) % T::TPU;
        f32x16 acc0 = {0}, acc1 = {0};
#ifdef MVK_ICPROF
        const unsigned long long ic_k0 = IC_CLK();
#endif
#pragma unroll
        for (int pr = 0; pr < 8; ++pr) {
          frag a_nxt[2][NP];  // depth 1: pair pr + 1; depth 2: pair pr + 2 (a_n1 holds pair pr + 1)
          if (D2) {
            if (pr < 6) read_pair(a_nxt, abuf, tt, pr + 2);
            else if (D2E) read_pair(a_nxt, next_buf, ntt, pr - 6);  // behind the (earlier) barrier: pairs 0, 1 of the next tile
          } else if (pr < 7) {
            read_pair(a_nxt, abuf, tt, pr + 1);
          }
          if (pr == 0) finish_pending(xb);
          if (D2E) {
            if (tt == 0) {  // conversion of the next unit over pairs 0-5
#pragma unroll
              for (int k = 0; k < T::NF4; ++k) {
                if (k * 6 / T::NF4 != pr) continue;
                write_f4(nbuf, k);
                raw[k] = src2[k * 256];
              }
            }
          } else
          if (tt == 0 && pr < 7) {  // conversion of the next unit: complete before the barrier behind pair 6
            if (T::NF4 >= 8) {
              constexpr int PER = T::NF4 / 8;
#pragma unroll
              for (int k = 0; k < (pr == 6 ? 2 * PER : PER); ++k) {
                write_f4(nbuf, pr * PER + k);
                raw[pr * PER + k] = src2[(pr * PER + k) * 256];
              }
            } else if (pr % (8 / T::NF4) == 0) {
              write_f4(nbuf, pr / (8 / T::NF4));
              raw[pr / (8 / T::NF4)] = src2[(pr / (8 / T::NF4)) * 256];
            }
          }
#pragma unroll
          for (int o = 0; o < T::OWN; ++o) {
            if (o * 8 / T::OWN != pr) continue;
            if (HAS_SRC) msk_cur[o] = srcp_1[obase + out_off(ptt1, o, 0)];
            epilogue_row(outp_e, ptt2, o, validf);
          }
          const int q0 = (2 * pr) / T::CHUNKS, c0 = (2 * pr) % T::CHUNKS;
          const int q1 = (2 * pr + 1) / T::CHUNKS, c1 = (2 * pr + 1) % T::CHUNKS;
          if constexpr (NP == 3) {
            constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {2, 1, 0, 1, 0, 0};  // small terms first
#pragma unroll
            for (int m = 0; m < 6; ++m) {
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[0][PA[m]], Bw[q0][c0][PB[m]], acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[1][PA[m]], Bw[q1][c1][PB[m]], acc1, 0, 0, 0);
            }
          } else {  // acc0 = main, acc1 = cross (dependent MFMAs on one accumulator issue back to back)
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[0][0], Bw[q0][c0][1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[0][0], Bw[q0][c0][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[1][0], Bw[q1][c1][1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[1][0], Bw[q1][c1][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[0][1], Bw[q0][c0][0], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[1][1], Bw[q1][c1][0], acc1, 0, 0, 0);
          }
          if (pr == 7) {  // behind the barrier: first fragments of the next tile, result of tile T-1
            if (D2E) {
            } else if (D2) {
              read_pair(a_n1, next_buf, ntt, 0);
              read_pair(a_nxt, next_buf, ntt, 1);
            } else {
              read_pair(a_nxt, next_buf, ntt, 0);
            }
            gather_result(xb);
          }
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) {
              if (D2) {
                a_cur[h][pc] = a_n1[h][pc];
                if (pr != 6 || D2E) a_n1[h][pc] = a_nxt[h][pc];  // depth 2, late barrier: pair 6 reads nothing (pair 8 is the next tile's)
              } else {
                a_cur[h][pc] = a_nxt[h][pc];
              }
            }
          if (MVK_IC_SCHED > 0) {  // "1 MFMA, N others" (see the one-tile-latency loop)
#pragma unroll
            for (int m = 0; m < 4 * NP; ++m) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x496, (MVK_IC_SCHED > 0 ? MVK_IC_SCHED : 1) * (NP == 3 ? 1 : MVK_IC_SCHED2), 0);
            }
          }
#ifdef MVK_ICPROF
          if (pr == BP) {
            const unsigned long long ic_b0 = IC_CLK();
            ic_k += ic_b0 - ic_k0;
            if (T::KSPLIT > 1 || tt == T::TPU - 1) __syncthreads();
            ic_tend = IC_CLK();
            ic_bar += ic_tend - ic_b0;
          }
#else
          if (pr == BP && (T::KSPLIT > 1 || tt == T::TPU - 1)) __syncthreads();
#endif
        }
#ifdef MVK_ICPROF
        ic_post += IC_CLK() - ic_tend;  // pair 7
#endif
        pend0 = acc0;
        pend1 = acc1;
#pragma unroll
        for (int o = 0; o < T::OWN; ++o) {
          res_prev[o] = res_next[o];
          msk_prev[o] = msk_cur[o];
        }
        xpar ^= 1;
        ++tcount;
      }
    }
    if (u0 < u1) {
      // drain 1: finish the last tile (N-1), epilogue of tile N-2, masks of tile N-1
      const long long ul = u1 - 1;
      char* const xb = xbase + xpar * T::XBUF;
      finish_pending(xb);
      {
        constexpr int tl2 = T::TPU == 1 ? 0 : T::TPU - 2;  // tile-in-unit index of tile N-2
        const bool valid2 = tcount >= 2;
        const long long u2 = T::TPU == 1 ? ul - 1 : ul;
        float* const outp_e = g.out + (valid2 ? u2 : ul) * OUT_UNIT;
        const float* const srcp_1 = HAS_SRC ? g.act_src + ul * OUT_UNIT : nullptr;
#pragma unroll
        for (int o = 0; o < T::OWN; ++o) {
          if (HAS_SRC) msk_cur[o] = srcp_1[obase + out_off(T::TPU - 1, o, 0)];
          epilogue_row(outp_e, valid2 ? tl2 : T::TPU - 1, o, valid2 ? 1.f : 0.f);
        }
      }
      if (T::KSPLIT > 1) __syncthreads();
      gather_result(xb);
      // drain 2: epilogue of the last tile
#pragma unroll
      for (int o = 0; o < T::OWN; ++o) {
        res_prev[o] = res_next[o];
        msk_prev[o] = msk_cur[o];
      }
      float* const outp_last = g.out + ul * OUT_UNIT;
#pragma unroll
      for (int o = 0; o < T::OWN; ++o) epilogue_row(outp_last, T::TPU - 1, o, 1.f);
    }
    };
    const int ks_u = __builtin_amdgcn_readfirstlane(ks);
    if (T::KSPLIT == 1 || ks_u == 0) run(std::integral_constant<int, 0>{});
    else if (T::KSPLIT > 1 && ks_u == 1) run(std::integral_constant<int, (T::KSPLIT > 1 ? 1 : 0)>{});
    else if (T::KSPLIT > 2 && ks_u == 2) run(std::integral_constant<int, (T::KSPLIT > 2 ? 2 : 0)>{});
    else if (T::KSPLIT > 3) run(std::integral_constant<int, (T::KSPLIT > 3 ? 3 : 0)>{});
  } else {
  bool first = true;
  int xpar = 0;
  for (long long u = u0; u < u1; ++u) {
    const int cur = (int)((u - u0) & 1);
    const char* const abuf = lds + cur * T::BUF;
    char* const nbuf = lds + (cur ^ 1) * T::BUF;
    const f32x4* const src2 = unit_src(u + 2);
    float* const outp_cur = g.out + u * OUT_UNIT;
    const float* const srcp_cur = HAS_SRC ? g.act_src + u * OUT_UNIT : nullptr;
#pragma unroll
    for (int tt = 0; tt < T::TPU; ++tt) {
      const int ptt = (tt + T::TPU - 1) % T::TPU;  // tile whose epilogue runs inside this k-loop
      // the very first tile has no predecessor: its slices store act(bias) into rows of this unit that the real
      // epilogue of those rows overwrites later (same lane, same address, program order) and add 0 to the column sums
      float* const outp_prev = (tt > 0 || first) ? outp_cur : outp_cur - OUT_UNIT;
      const float validf = first ? 0.f : 1.f;
      f32x16 acc0 = {0}, acc1 = {0};
#ifdef MVK_ICPROF
      const unsigned long long ic_k0 = IC_CLK();
#endif
#pragma unroll
      for (int pr = 0; pr < 8; ++pr) {
        bf16x8 a_nxt[2][3];
        if (pr < 7) read_pair(a_nxt, abuf, tt, pr + 1);
        if (tt == 0) {
          constexpr int PER = T::NF4 >= 8 ? T::NF4 / 8 : 1;
          if (T::NF4 >= 8) {
#pragma unroll
            for (int k = 0; k < PER; ++k) {
              write_f4(nbuf, pr * PER + k);
              raw[pr * PER + k] = src2[(pr * PER + k) * 256];
            }
          } else if (pr % (8 / T::NF4) == 0) {
            write_f4(nbuf, pr / (8 / T::NF4));
            raw[pr / (8 / T::NF4)] = src2[(pr / (8 / T::NF4)) * 256];
          }
        }
        // rows o of this pair's slice: OWN rows over 8 pairs
#pragma unroll
        for (int o = 0; o < T::OWN; ++o) {
          if (o * 8 / T::OWN != pr) continue;
          if (HAS_SRC) msk_cur[o] = srcp_cur[obase + out_off(tt, o, 0)];
          epilogue_row(outp_prev, ptt, o, validf);
        }
        constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {2, 1, 0, 1, 0, 0};  // small terms first
        const int q0 = (2 * pr) / T::CHUNKS, c0 = (2 * pr) % T::CHUNKS;
        const int q1 = (2 * pr + 1) / T::CHUNKS, c1 = (2 * pr + 1) % T::CHUNKS;
#pragma unroll
        for (int m = 0; m < 6; ++m) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[0][PA[m]], Bw[q0][c0][PB[m]], acc0, 0, 0, 0);
#ifdef MVK_IC_ONECHAIN
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[1][PA[m]], Bw[q1][c1][PB[m]], acc0, 0, 0, 0);
#else
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[1][PA[m]], Bw[q1][c1][PB[m]], acc1, 0, 0, 0);
#endif
        }
        if (pr < 7) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) a_cur[h][pc] = a_nxt[h][pc];
        }
        // One wave per SIMD: the ~28 idle issue cycles behind an MFMA hide about five single-issue instructions, a cluster
        // of MFMAs hides none — and hipcc clusters the 12 MFMAs of a k-step pair (108 of 191 MFMA-to-MFMA gaps empty,
        // the VALU / LDS / memory work in bursts of 7-20 between the clusters).  Ask the scheduler for the pipeline
        // "1 MFMA, N others" instead (N = 4 ... 10 measured: 5-6 is best; cycles per wave at n = 5120:
        // down 8x8 254 k -> 214 k, down 4x4 262 k -> 222 k, up 4x4 218 k -> 197 k, up 8x8 174 k -> 168 k).
        if (MVK_IC_SCHED > 0) {
#pragma unroll
          for (int m = 0; m < 12; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // MFMA
            __builtin_amdgcn_sched_group_barrier(0x496, MVK_IC_SCHED > 0 ? MVK_IC_SCHED : 1, 0);  // VALU | SALU | VMEM | DS | TRANS
          }
        }
      }
#ifdef MVK_ICPROF
      ic_k += IC_CLK() - ic_k0;
      const unsigned long long ic_p0 = IC_CLK();
#endif
#ifndef MVK_IC_ONECHAIN
      acc0 += acc1;
#endif
      // next tile: same unit and buffer, or the first tile of the next unit in the other buffer (complete after the
      // barrier below; the last unit of the workgroup re-reads a converted copy of itself, unused)
      const char* const next_buf = (tt + 1 < T::TPU) ? abuf : nbuf;
      const int ntt = (tt + 1) % T::TPU;
      if (T::KSPLIT > 1) {
        // accumulator exchange: registers [r*OWN, r*OWN+OWN) go to the wave of rank r of this tile's group
        char* const xb = xbase + xpar * T::XBUF;
#pragma unroll
        for (int r = 0; r < T::KSPLIT; ++r) {
          if (r == ks) continue;
          const int dwave = T::WG_TYPES == 1 || KIND == IC_UP ? r * T::NCT + ct : r;
          const int slot = ks < r ? ks : ks - 1;
          float* dst = reinterpret_cast<float*>(xb + dwave * T::XWAVE + slot * T::OWN * 256);
#pragma unroll
          for (int o = 0; o < T::OWN; ++o) dst[o * 64 + lane] = acc0[r * T::OWN + o];
        }
#ifdef MVK_ICPROF
        const unsigned long long ic_b0 = IC_CLK();
        ic_pre += ic_b0 - ic_p0;
        __syncthreads();
        ic_tend = IC_CLK();
        ic_bar += ic_tend - ic_b0;
#else
        __syncthreads();
#endif
        read_pair(a_cur, next_buf, ntt, 0);
        const float* src = reinterpret_cast<const float*>(xb + wave * T::XWAVE);
#pragma unroll
        for (int o = 0; o < T::OWN; ++o) {
          float v = 0.f;
#pragma unroll
          for (int r = 0; r < T::KSPLIT; ++r) {  // fixed order over the ranks: deterministic
            if (r == ks) {
              float mine = 0.f;
#pragma unroll
              for (int rr = 0; rr < T::KSPLIT; ++rr)
                if (rr == ks) mine = acc0[rr * T::OWN + o];
              v += mine;
            } else {
              const int slot = r < ks ? r : r - 1;
              v += src[(slot * T::OWN + o) * 64 + lane];
            }
          }
          res_prev[o] = v;
        }
        xpar ^= 1;
      } else {
#pragma unroll
        for (int o = 0; o < T::OWN; ++o) res_prev[o] = acc0[o];
#ifdef MVK_ICPROF
        const unsigned long long ic_b0 = IC_CLK();
        ic_pre += ic_b0 - ic_p0;
        if (tt == T::TPU - 1) __syncthreads();
        ic_tend = IC_CLK();
        ic_bar += ic_tend - ic_b0;
#else
        if (tt == T::TPU - 1) __syncthreads();
#endif
        read_pair(a_cur, next_buf, ntt, 0);
      }
#pragma unroll
      for (int o = 0; o < T::OWN; ++o) msk_prev[o] = msk_cur[o];
      first = false;
#ifdef MVK_ICPROF
      ic_post += IC_CLK() - ic_tend;
#endif
    }
  }
  if (u0 < u1) {  // epilogue of the last tile
    float* const outp_last = g.out + (u1 - 1) * OUT_UNIT;
#pragma unroll
    for (int o = 0; o < T::OWN; ++o) epilogue_row(outp_last, T::TPU - 1, o, 1.f);
  }

  }
  if (g.y_amax) {  // uniform branch; csred is free until the column sums below
    amax_publish(amax_l, g.y_amax, csred);
    __syncthreads();
  }
  if (g.colsum_part) {  // fixed-order sum over the waves that share a column tile
    csum += __shfl_xor(csum, 32, 64);
    if (kg == 0) csred[wave * 32 + col] = csum;
    __syncthreads();
    if (KIND == IC_DOWN && T::WG_TYPES > 1) {  // all four waves hold column tile `wgtype`
      if (tid < 32) {
        float s = 0.f;
        for (int w = 0; w < 4; ++w) s += csred[w * 32 + tid];
        g.colsum_part[(long long)worker * COUT + wgtype * 32 + tid] = s;
      }
    } else if (tid < COUT) {
      const int c_t = tid / 32, c_l = tid % 32;
      float s = 0.f;
      for (int w = 0; w < 4; ++w) {
        const int wct = KIND == IC_UP ? (wgtype * 4 + w) % T::NCT : w % T::NCT;
        if (wct == c_t) s += csred[w * 32 + c_l];
      }
      g.colsum_part[(long long)(worker * T::WG_TYPES + wgtype) * COUT + tid] = s;  // row = (worker, type), whatever block ran it
    }
  }
#ifdef MVK_ICPROF
  if (g_ic_dbg && lane == 0) {
    unsigned long long* o = g_ic_dbg + (blockIdx.x * 4 + wave) * 4;
    o[0] = IC_CLK() - ic_t0;
    o[1] = ic_bar;
    o[2] = ic_k;
    o[3] = (ic_pre << 32) | (ic_post & 0xffffffffull);
  }
#endif
  mvk_prof_end(g.prof);
}

// scaled-fp16 form (two-tile-latency loop only)
template <int KIND, int HS, int CIN, int COUT>
static int imgconv_launch_f16(const ImgConvArgs& a, int* part_rows, hipStream_t s) {
  using T = ICfg<KIND, HS, CIN, COUT, 2>;
  static bool attr_done = false;
  auto kern = a.act_src ? imgconv_kernel<KIND, HS, CIN, COUT, true, true, 2> : imgconv_kernel<KIND, HS, CIN, COUT, false, true, 2>;
  if (!attr_done) {
    const void* all[2] = {reinterpret_cast<const void*>(imgconv_kernel<KIND, HS, CIN, COUT, true, true, 2>),
                          reinterpret_cast<const void*>(imgconv_kernel<KIND, HS, CIN, COUT, false, true, 2>)};
    for (const void* f : all)
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES) != hipSuccess) return MVK_ELAUNCH;
    attr_done = true;
  }
  const int grid = 256;
  if (part_rows) *part_rows = (KIND == IC_DOWN && T::WG_TYPES > 1) ? grid / T::WG_TYPES : grid;
  ImgConvArgs ap = a;
  ap.prof = prof_next(KIND == IC_UP ? 2 : 3, 2.0 * a.n * HS * HS * 16.0 * CIN * COUT);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), T::LDS_BYTES, s, ap);
  MVK_CHECK_LAUNCH();
  prof_fold(ap.prof, s);
  return MVK_OK;
}

template <int KIND, int HS, int CIN, int COUT>
static int imgconv_launch(const ImgConvArgs& a, int* part_rows, hipStream_t s) {
  using T = ICfg<KIND, HS, CIN, COUT>;
  if (a.x_amax && a.w_amax) return imgconv_launch_f16<KIND, HS, CIN, COUT>(a, part_rows, s);
  static bool attr_done = false;
  // MVK_IMGCONV_PIPE=0: the one-tile-latency main loop (A/B; results are bit-identical)
  static const bool pipe = !(mvk_tune("MVK_IMGCONV_PIPE") && atoi(mvk_tune("MVK_IMGCONV_PIPE")) == 0);
  auto kern = a.act_src ? (pipe ? imgconv_kernel<KIND, HS, CIN, COUT, true, true> : imgconv_kernel<KIND, HS, CIN, COUT, true, false>)
                        : (pipe ? imgconv_kernel<KIND, HS, CIN, COUT, false, true> : imgconv_kernel<KIND, HS, CIN, COUT, false, false>);
  if (!attr_done) {
    const void* all[4] = {reinterpret_cast<const void*>(imgconv_kernel<KIND, HS, CIN, COUT, true, true>),
                          reinterpret_cast<const void*>(imgconv_kernel<KIND, HS, CIN, COUT, true, false>),
                          reinterpret_cast<const void*>(imgconv_kernel<KIND, HS, CIN, COUT, false, true>),
                          reinterpret_cast<const void*>(imgconv_kernel<KIND, HS, CIN, COUT, false, false>)};
    for (const void* f : all)
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES) != hipSuccess) return MVK_ELAUNCH;
    attr_done = true;
  }
  const int grid = 256;
  if (part_rows) *part_rows = (KIND == IC_DOWN && T::WG_TYPES > 1) ? grid / T::WG_TYPES : grid;
  ImgConvArgs ap = a;
  ap.prof = prof_next(KIND == IC_UP ? 2 : 3, 2.0 * a.n * HS * HS * 16.0 * CIN * COUT);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), T::LDS_BYTES, s, ap);
  MVK_CHECK_LAUNCH();
  prof_fold(ap.prof, s);
  return MVK_OK;
}

// =====================================================================================================================
// Weight gradient of the same layer pairs:  dW[cv][cu][kh][kw] += sum_{n,i,j} V[n,i,j,cv] U[n,2i-1+kh,2j-1+kw,cu]
// =====================================================================================================================
// The reduction index of this GEMM is the pixel, and both operands are stored channel-contiguous ([pixel][channel]),
// i.e. transposed for the matrix cores.  gfx950's ds_read_b64_tr_b16 transposes on the way out of LDS (16 lanes read
// a [4 pixels][16 channels] block as 8-byte row pieces and receive 4 pixels of one channel each), so the images are
// staged exactly as in the kernels above (one coalesced pass, one split, three bf16 planes [pixel][channel]) and every
// (tap, channel tile) use is a pair of transposing reads at per-lane gathered pixel addresses (halo -> zero row).
//   * OUTPUT-stationary: a wave keeps 8 accumulator tiles (128 registers) for the whole launch: the 4 waves of a
//     workgroup hold the 16 taps x 32 x 64 gradient of the 64<->32 layer (one kernel row per wave); the 128<->64 layer
//     takes one kernel row per workgroup type, one tap per wave;
//   * a workgroup walks its share of the images once, one s_barrier per stage unit, and writes its partial gradient
//     as one slab; the slabs are summed in a fixed order (deterministic) into the reference layout.
// NP = 3: bf16 pieces (U and V three planes each, 6 MFMAs per product).  NP = 2: scaled fp16 with ONE accumulator per tile
// (the eight tiles leave no registers for a second one; see c3wg_kernel): U as (uh, ul) = 2 planes, V as (vh, vH, vl) = 3
// planes, acc += uh vH + uh vl + ul vh = 2^11 su sv u v.
template <int HS, int CU, int CV, int NP = 3>
struct WCfg {
  static constexpr int PIX = HS * HS, AW = 2 * HS;
  static constexpr bool SPLIT_KH = (CU * CV > 32 * 64);          // one kernel row per workgroup type
  static constexpr int WG_TYPES = SPLIT_KH ? 4 : 1;
  static constexpr int NKW = SPLIT_KH ? 1 : 4;                   // taps (kw) per wave
  static constexpr int MT = SPLIT_KH ? CU / 32 : 4;              // A fragments per k-step: cu tiles, or the 4 kw taps
  static constexpr int NT = CV / 32;
  static constexpr int SU = PIX >= 32 ? 1 : 2;                   // images per stage unit
  static constexpr int KS = SU * PIX / 16;                       // 16-pixel k-steps per unit
  static constexpr int UROWS_IMG = SPLIT_KH ? HS * AW : 4 * PIX; // staged input rows per image (one row per (i, x))
  static constexpr int UROWS = SU * UROWS_IMG, VROWS = SU * PIX;
#ifndef MVK_IW_PAD_A
#define MVK_IW_PAD_A 32  // tools/imgwgrad_pad.sh: 16 -> 112.8 us, 32 ... 72 -> 102-107 us (64 <-> 128 channels, n = 5120)
#endif
#ifndef MVK_IW_PAD_B
#define MVK_IW_PAD_B 0
#endif
  static constexpr int PAD = SPLIT_KH ? MVK_IW_PAD_A : MVK_IW_PAD_B;  // row padding (bytes) of the LDS images
  static constexpr int SUB = CU * 2 + PAD, SVB = CV * 2 + PAD;   // bytes per LDS row
  static constexpr int PLANE_U = (UROWS + 1) * SUB, PLANE_V = VROWS * SVB;
  static constexpr int OFF_V = NP * PLANE_U;
  static constexpr int BUF = NP * PLANE_U + 3 * PLANE_V;
  static constexpr int NFU = UROWS * CU / 4 / 256, NFV = VROWS * CV / 4 / 256;
  static constexpr int NF = NFU + NFV;
  static constexpr int LDS_BYTES = 2 * BUF;
  static_assert(MT * NT == 8, "8 accumulator tiles per wave");
  static_assert(SPLIT_KH || CU == 32, "the 4 taps of a kernel row are the 4 row tiles");
  static_assert(UROWS * CU / 4 % 256 == 0 && VROWS * CV / 4 % 256 == 0 && NF % KS == 0, "staging split");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

typedef _Float16 ic_f16x4 __attribute__((ext_vector_type(4)));
template <typename F>
__device__ __forceinline__ F ic_tr_pair(const char* p0, const char* p1) {
  if constexpr (std::is_same<F, bf16x8>::value) {
    typedef __attribute__((address_space(3))) bf16x4* lp;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(p0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(p1));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  } else {
    typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
    typedef __attribute__((address_space(3))) h4* lp;
    const ic_f16x4 lo = __builtin_bit_cast(ic_f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(p0)));
    const ic_f16x4 hi = __builtin_bit_cast(ic_f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(p1)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

struct ImgWgradArgs {
  const float* U;  // [n][2h][2w][CU]
  const float* V;  // [n][h][w][CV]
  float* slab;     // [workers][16 * CU][CV] partial gradients
  int n;
  mvk_prof_slot* prof;
  const float* u_amax;  // scaled-fp16 form: device scalars bounding max |U| and max |V|
  const float* v_amax;
};

template <int HS, int CU, int CV, int NP = 3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void imgwgrad_kernel(const ImgWgradArgs g) {
  using T = WCfg<HS, CU, CV, NP>;
  using frag = std::conditional_t<NP == 3, bf16x8, f16x8>;
  constexpr int NPU = NP, NPV = 3;  // planes of U / V
  const float su = NP == 2 ? f16_scale_of(*g.u_amax) : 1.f;
  const float sv = NP == 2 ? f16_scale_of(*g.v_amax) * (1.f / 1024.f) : 1.f, sv11 = sv * 2048.f;  // max |V| sv in [2^3, 2^4)
  extern __shared__ __attribute__((aligned(16))) char lds[];
  mvk_prof_begin(g.prof);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Workgroup types of one worker read the SAME images (each holds another slice of the weights).  Block b runs on XCD b % 8
  // (observed placement; a speed choice only), and every XCD has its own L2: with type = b % WG_TYPES the four readers of an
  // image sat on four XCDs and the input crossed the fabric 2.0-2.5 times (profiles/r04_pmc_hbm.md) in launches that move
  // 250-420 MB in 60-80 us, i.e. run at the HBM rate.  Types of a worker on ONE XCD: slot = b / 8 on XCD b % 8, type = slot %
  // WG_TYPES, so the four workgroups stream the same lines through one L2 at the same time.  Same unit ranges per worker.
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
  const int kh = T::SPLIT_KH ? wgtype : wave;

  // zero rows of the U planes of both buffers
  for (int i = tid; i < 2 * NPU * (T::SUB / 4); i += 256) {
    const int pl = i / (T::SUB / 4), w = i % (T::SUB / 4);
    *reinterpret_cast<unsigned*>(lds + (pl / NPU) * T::BUF + (pl % NPU) * T::PLANE_U + T::UROWS * T::SUB + w * 4) = 0u;
  }
  // transposing-read addresses: 16-lane group gq reads [4 pixels][16 channels]; lane lp supplies pixel (lp >> 2),
  // channels 4 (lp & 3) .. +3 and receives channel lp of the block, pixels 0..3
  const int gq = lane >> 4, lp = lane & 15;
  const int kbase = 8 * (gq >> 1) + (lp >> 2);
  const int cch = 16 * (gq & 1) + 4 * (lp & 3);
  int uaddr[T::KS][2][T::NKW], vaddr[T::KS][2];
#pragma unroll
  for (int s = 0; s < T::KS; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int p = s * 16 + kbase + 4 * t;
      const int img = p / T::PIX, pp = p % T::PIX, i = pp / HS, j = pp % HS;
      vaddr[s][t] = T::OFF_V + p * T::SVB + cch * 2;
      const int y = 2 * i - 1 + kh;
#pragma unroll
      for (int q = 0; q < T::NKW; ++q) {
        const int kw = T::SPLIT_KH ? wave : q;
        const int x = 2 * j - 1 + kw;
        const bool ok = y >= 0 && y < T::AW && x >= 0 && x < T::AW;
        const int row = img * T::UROWS_IMG + (T::SPLIT_KH ? i * T::AW + x : y * T::AW + x);
        uaddr[s][t][q] = (ok ? row : T::UROWS) * T::SUB + cch * 2;
      }
    }
  // staging: NFU float4 units of U, NFV of V per thread and unit
  int sdst[T::NF], ssrc[T::NF];  // LDS byte offset (plane 0), float4 offset inside the unit's source block
#pragma unroll
  for (int k = 0; k < T::NFU; ++k) {
    const int f = tid + k * 256;
    const int row = f / (CU / 4), c4 = f % (CU / 4);
    sdst[k] = row * T::SUB + c4 * 8;
    if (T::SPLIT_KH) {  // row = (img, i, x): source pixel (2i-1+kh, x), clamped (a clamped row is never read)
      const int img = row / T::UROWS_IMG, r = row % T::UROWS_IMG, i = r / T::AW, x = r % T::AW;
      int y = 2 * i - 1 + kh;
      y = y < 0 ? 0 : (y >= T::AW ? T::AW - 1 : y);
      ssrc[k] = ((img * T::AW + y) * T::AW + x) * (CU / 4) + c4;
    } else {
      ssrc[k] = f;
    }
  }
#pragma unroll
  for (int k = 0; k < T::NFV; ++k) {
    const int f = tid + k * 256;
    sdst[T::NFU + k] = T::OFF_V + (f / (CV / 4)) * T::SVB + (f % (CV / 4)) * 8;
    ssrc[T::NFU + k] = f;
  }
  constexpr long long U_UNIT = (long long)T::SU * 4 * T::PIX * CU / 4, V_UNIT = (long long)T::SU * T::PIX * CV / 4;  // float4s

  const long long units = g.n / T::SU;
  // readfirstlane: the 64-bit division is emitted on the vector ALU, which makes the trip count of the unit loop 'divergent'
  // for hipcc (exec-masked loop, every live-out accumulator read back per iteration: 32 v_accvgpr_read + a drain of the matrix pipe per tile)
  const long long u0 = MVK_RFL((int)(units * worker / workers));
  const long long u1 = MVK_RFL((int)(units * (worker + 1) / workers));
  f32x4 raw[T::NF];
  auto load_f4 = [&](long long u, int k) {
    const long long uc = (MVK_IC_ABL & 1) ? u0 : (u < u1 ? u : u1 - 1);
    const f32x4* base = k < T::NFU ? reinterpret_cast<const f32x4*>(g.U) + uc * U_UNIT
                                   : reinterpret_cast<const f32x4*>(g.V) + uc * V_UNIT;
    raw[k] = base[ssrc[k]];
  };
  auto write_f4 = [&](char* buf, int k) {
    const int pl = k < T::NFU ? T::PLANE_U : T::PLANE_V;
    if constexpr (NP == 3) {
      unsigned a0, a1, a2, b0, b1, b2;
      bf3_split(raw[k][0], raw[k][1], a0, a1, a2);
      bf3_split(raw[k][2], raw[k][3], b0, b1, b2);
      *reinterpret_cast<u32x2*>(buf + sdst[k]) = u32x2{a0, b0};
      *reinterpret_cast<u32x2*>(buf + pl + sdst[k]) = u32x2{a1, b1};
      *reinterpret_cast<u32x2*>(buf + 2 * pl + sdst[k]) = u32x2{a2, b2};
    } else if (k < T::NFU) {  // U: (uh, ul)
      unsigned a0, a1, b0, b1;
      f16_split_su(raw[k][0], raw[k][1], su, a0, a1);
      f16_split_su(raw[k][2], raw[k][3], su, b0, b1);
      *reinterpret_cast<u32x2*>(buf + sdst[k]) = u32x2{a0, b0};
      *reinterpret_cast<u32x2*>(buf + pl + sdst[k]) = u32x2{a1, b1};
    } else {  // V: planes (vh, vH = fp16(v sv 2^11), vl = fp16(v sv 2^11 - vH))
      unsigned h01, H01, l01, h23, H23, l23;
      f16_split3_su(raw[k][0], raw[k][1], sv, sv11, h01, H01, l01);
      f16_split3_su(raw[k][2], raw[k][3], sv, sv11, h23, H23, l23);
      *reinterpret_cast<u32x2*>(buf + sdst[k]) = u32x2{h01, h23};
      *reinterpret_cast<u32x2*>(buf + pl + sdst[k]) = u32x2{H01, H23};
      *reinterpret_cast<u32x2*>(buf + 2 * pl + sdst[k]) = u32x2{l01, l23};
    }
  };

  f32x16 acc[T::MT][T::NT];
#pragma unroll
  for (int a = 0; a < T::MT; ++a)
#pragma unroll
    for (int b = 0; b < T::NT; ++b) acc[a][b] = f32x16{0};

  if (u0 < u1) {
#pragma unroll
    for (int k = 0; k < T::NF; ++k) load_f4(u0, k);
#pragma unroll
    for (int k = 0; k < T::NF; ++k) write_f4(lds, k);
#pragma unroll
    for (int k = 0; k < T::NF; ++k) load_f4(u0 + 1, k);
  }
  __syncthreads();

  frag Af[T::MT][NPU], Bf[T::NT][NPV];
  auto read_frags = [&](frag (&A)[T::MT][NPU], frag (&B)[T::NT][NPV], const char* buf, int s) {
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
      for (int b = 0; b < T::NT; ++b)
        B[b][pc] = ic_tr_pair<frag>(buf + vaddr[s][0] + pc * T::PLANE_V + b * 64, buf + vaddr[s][1] + pc * T::PLANE_V + b * 64);
      if (pc < NPU) {
#pragma unroll
        for (int a = 0; a < T::MT; ++a) {
          const int q = T::SPLIT_KH ? 0 : a, co = T::SPLIT_KH ? a * 64 : 0;
          A[a][pc] = ic_tr_pair<frag>(buf + uaddr[s][0][q] + pc * T::PLANE_U + co, buf + uaddr[s][1][q] + pc * T::PLANE_U + co);
        }
      }
    }
  };
  if (u0 < u1) read_frags(Af, Bf, lds, 0);

  for (long long u = u0; u < u1; ++u) {
    const int cur = (int)((u - u0) & 1);
    const char* const abuf = lds + cur * T::BUF;
    char* const nbuf = lds + (cur ^ 1) * T::BUF;
#pragma unroll
    for (int s = 0; s < T::KS; ++s) {
      frag An[T::MT][NPU], Bn[T::NT][NPV];
      if (s + 1 < T::KS) read_frags(An, Bn, abuf, s + 1);
      constexpr int PER = T::NF / T::KS;
#pragma unroll
      for (int k = s * PER; k < (s + 1) * PER; ++k) {  // conversion of the next unit + reload with the one after it
        write_f4(nbuf, k);
        load_f4(u + 2, k);
      }
      if constexpr (NP == 3) {
        constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {2, 1, 0, 1, 0, 0};  // small terms first
#pragma unroll
        for (int m = 0; m < 6; ++m)
#pragma unroll
          for (int a = 0; a < T::MT; ++a)
#pragma unroll
            for (int b = 0; b < T::NT; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Af[a][PA[m]], Bf[b][PB[m]], acc[a][b], 0, 0, 0);
      } else {
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 2, 1};  // ul vh, uh vl, uh vH
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
          for (int a = 0; a < T::MT; ++a)
#pragma unroll
            for (int b = 0; b < T::NT; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Af[a][PA[m]], Bf[b][PB[m]], acc[a][b], 0, 0, 0);
      }
#ifndef MVK_IW_SCHED
#define MVK_IW_SCHED 4
#endif
      if (MVK_IW_SCHED > 0) {  // "1 MFMA, N others" instead of hipcc's MFMA clusters (see imgconv_kernel)
#pragma unroll
        for (int m = 0; m < 2 * NP * T::MT * T::NT; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x496, (MVK_IW_SCHED > 0 ? MVK_IW_SCHED : 1) * (NP == 3 ? 1 : 2), 0);
        }
      }
      if (s + 1 < T::KS) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          if (pc < NPU) {
#pragma unroll
            for (int a = 0; a < T::MT; ++a) Af[a][pc] = An[a][pc];
          }
#pragma unroll
          for (int b = 0; b < T::NT; ++b) Bf[b][pc] = Bn[b][pc];
        }
      }
    }
    __syncthreads();
    read_frags(Af, Bf, nbuf, 0);
  }

  // partial gradient of this workgroup: slab[worker][(tap * CU + cu)][cv]
  float* const slab = g.slab + (long long)worker * 16 * CU * CV;
  const int col = lane & 31, kg = lane >> 5;
  const float oa = NP == 2 ? f16_inv_scale(su) : 1.f, ob = NP == 2 ? f16_inv_scale(sv) * (1.f / 2048.f) : 1.f;
#pragma unroll
  for (int a = 0; a < T::MT; ++a) {
    const int kw = T::SPLIT_KH ? wave : a, cut = T::SPLIT_KH ? a : 0;
    const int tap = kh * 4 + kw;
#pragma unroll
    for (int b = 0; b < T::NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cu = cut * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        slab[(long long)(tap * CU + cu) * CV + b * 32 + col] = NP == 2 ? acc[a][b][r] * oa * ob : acc[a][b][r];
      }
  }
  mvk_prof_end(g.prof);
}

template <int HS, int CU, int CV, int NP>
static int imgwgrad_launch_np(const ImgWgradArgs& a, int* nz, hipStream_t s);
template <int HS, int CU, int CV>
static int imgwgrad_launch(const ImgWgradArgs& a, int* nz, hipStream_t s) {
  return a.u_amax && a.v_amax ? imgwgrad_launch_np<HS, CU, CV, 2>(a, nz, s) : imgwgrad_launch_np<HS, CU, CV, 3>(a, nz, s);
}
template <int HS, int CU, int CV, int NP>
static int imgwgrad_launch_np(const ImgWgradArgs& a, int* nz, hipStream_t s) {
  using T = WCfg<HS, CU, CV, NP>;
  static bool attr_done = false;
  auto kern = imgwgrad_kernel<HS, CU, CV, NP>;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            T::LDS_BYTES) != hipSuccess)
      return MVK_ELAUNCH;
    attr_done = true;
  }
  // MVK_IMGWGRAD_GRID: fewer workgroups than compute units leave room for launches of other streams (a workgroup owns a
  // whole compute unit's registers)
  static const int grid_env = mvk_tune("MVK_IMGWGRAD_GRID") ? atoi(mvk_tune("MVK_IMGWGRAD_GRID")) : 256;
  const int grid = (grid_env >= T::WG_TYPES && grid_env <= 256) ? grid_env / T::WG_TYPES * T::WG_TYPES : 256;
  *nz = grid / T::WG_TYPES;
  ImgWgradArgs ap = a;
  ap.prof = prof_next(4, 2.0 * a.n * HS * HS * 16.0 * CU * CV);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), T::LDS_BYTES, s, ap);
  MVK_CHECK_LAUNCH();
  prof_fold(ap.prof, s);
  return MVK_OK;
}

// slab: [*nz][16 Cu][Cv] partial gradients (needs 256 / types * 16 Cu Cv floats); 1: shape not covered
int imgconv_wgrad(const float* U, const float* V, float* slab, long long slab_floats, int n, int h, int w, int Cu, int Cv,
                  int* nz, const float* u_amax, const float* v_amax, hipStream_t s) {
  if (h != w) return 1;
  ImgWgradArgs a{U, V, slab, n, nullptr, u_amax, v_amax};
  if (h == 8 && Cu == 32 && Cv == 64 && slab_floats >= 256ll * 16 * Cu * Cv) return imgwgrad_launch<8, 32, 64>(a, nz, s);
  if (h == 4 && Cu == 64 && Cv == 128 && n % 2 == 0 && slab_floats >= 64ll * 16 * Cu * Cv)
    return imgwgrad_launch<4, 64, 128>(a, nz, s);
  return 1;
}

// 1: shape not covered (the caller falls back to the implicit-GEMM engine)
int imgconv_up(const float* V, const float* Wup, const void* wfrag, const float* bias, float* U, int n, int h, int w, int Cu,
               int Cv, int act, const float* u_act_src, int u_act, float* colsum_part, int* part_rows, const float* x_amax,
               const float* w_amax, float* y_amax, hipStream_t s) {
  if (h != w) return 1;
  ImgConvArgs a{V, Wup, wfrag, bias, U, u_act_src, colsum_part, n, act, u_act, nullptr, x_amax, w_amax, y_amax};
  if (h == 8 && Cv == 64 && Cu == 32) return imgconv_launch<IC_UP, 8, 64, 32>(a, part_rows, s);
  if (h == 4 && Cv == 128 && Cu == 64 && n % 2 == 0) return imgconv_launch<IC_UP, 4, 128, 64>(a, part_rows, s);
  return 1;
}

int imgconv_down(const float* U, const float* Wdown, const void* wfrag, const float* bias, float* V, int n, int h, int w,
                 int Cu, int Cv, int act, const float* v_act_src, int v_act, float* colsum_part, int* part_rows,
                 const float* x_amax, const float* w_amax, float* y_amax, hipStream_t s) {
  if (h != w) return 1;
  ImgConvArgs a{U, Wdown, wfrag, bias, V, v_act_src, colsum_part, n, act, v_act, nullptr, x_amax, w_amax, y_amax};
  if (h == 8 && Cu == 32 && Cv == 64) return imgconv_launch<IC_DOWN, 8, 32, 64>(a, part_rows, s);
  if (h == 4 && Cu == 64 && Cv == 128 && n % 2 == 0) return imgconv_launch<IC_DOWN, 4, 64, 128>(a, part_rows, s);
  return 1;
}

}  // namespace mvk

#ifdef MVK_ICPROF
extern "C" int mvk_imgconv_debug_buffer(unsigned long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(mvk::g_ic_dbg), &p, sizeof(p)) == hipSuccess ? 0 : -2;
}
#endif
