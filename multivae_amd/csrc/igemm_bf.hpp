// Split-precision variant of the implicit-GEMM engine: fp32 operands, fp32 result, bf16 matrix cores.
//
// gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the rate of the fp32-input MFMA.  Every fp32 operand value x is
// split, while it is staged into LDS, into three bf16 pieces  x = x0 + x1 + x2  (x0 = bf16(x), x1 = bf16(x - x0),
// x2 = bf16(x - x0 - x1), round-to-nearest; |x - (x0+x1+x2)| <= 2^-26 |x|).  A product a*b is accumulated as the
// six piece products of order <= 2
//     a0 b2 + a1 b1 + a2 b0 + a0 b1 + a1 b0 + a0 b0        (dropped: a1 b2, a2 b1, a2 b2  <= 2^-26 |a b|)
// each exact in the fp32 accumulator's input (8-bit x 8-bit significands), so the result carries fp32-level
// error (tests/test_gpu_kernels.py compares both engines with an fp64 product) at 6/16 of the fp32 MFMA time.
//
// LDS image (per operand, per piece): row-major bf16 tiles [row][32 k] with an 80-byte row stride, rows
// regrouped by (row & 3):  byte(row, k) = (row & 3) * G + (row >> 2) * 80 + 2 k,   G = (rows/4) * 80 + 64.
//   * fragment read  (ds_read_b128, lane = row l&31, k-half l>>5): 16 consecutive lanes hit 16 different
//     16-byte bank groups ((l&3)*4 + 5*(l>>2) mod 16);
//   * k-contiguous operands (one float4 = 4 k of one row per lane, 8 lanes per row) write ds_write_b64 to 4
//     rows x 64 contiguous bytes whose (row&3) groups start 64 bytes apart: conflict-free;
//   * row-contiguous operands (one float4 = 4 rows of one k per lane) transpose in registers over the lane's
//     consecutive k values and write one (row&3) group per instruction at an 80-byte lane stride.
#pragma once
#include "bf3.hpp"
#include "igemm_fast.hpp"

#ifndef MVK_BF_OCC_SMALL
#define MVK_BF_OCC_SMALL 3  // 4 (<= 128 VGPRs) spills 92-240 bytes per lane in every 128x32 variant
#endif
namespace mvk {


template <int BM, int BN>
struct BfCfg {
  static constexpr int BKT = 32;
  static constexpr int WAVES_N = (BN >= 64) ? 2 : 1;
  static constexpr int WAVES_M = 4 / WAVES_N;
  static constexpr int WTM = BM / WAVES_M;
  static constexpr int WTN = BN / WAVES_N;
  static constexpr int TM = WTM / 32;
  static constexpr int TN = WTN / 32;
  static constexpr int RS = 80;                        // bytes per LDS row (64 data + 16 pad)
  static constexpr int GA = (BM / 4) * RS + 64;        // bytes between (row & 3) groups
  static constexpr int GB = (BN / 4) * RS + 64;
  static constexpr int A_PIECE = 4 * GA;               // bytes per bf16 piece
  static constexpr int B_PIECE = 4 * GB;
  static constexpr int LDS_BYTES = 3 * (A_PIECE + B_PIECE);
  static constexpr int NA4 = BM * BKT / 4 / 256;       // float4 units per thread
  static constexpr int NB4 = BN * BKT / 4 / 256;
  static_assert(TM >= 1 && TN >= 1 && NA4 >= 1 && NB4 >= 1, "tile too small");
  static_assert(BM * BKT / 4 % 256 == 0 && BN * BKT / 4 % 256 == 0, "tile must split evenly over 256 threads");
};

// two fp32 values -> three dwords, each holding the (lo = x, hi = y) pair of one bf16 piece
#ifdef MVK_EXPER
__device__ int g_bf_flags;
#endif
__device__ __forceinline__ void bf_split3(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
#ifdef MVK_EXPER
  if (g_bf_flags & 1) {  // experiment: no conversion arithmetic
    p0 = __builtin_amdgcn_perm(__float_as_uint(y), __float_as_uint(x), 0x07060302u);
    p1 = p0;
    p2 = p0;
    return;
  }
#endif
  // scalar subtractions on purpose (built with -fno-slp-vectorize): v_pk_add_f32 needs aligned register pairs,
  // which costs copies of freshly loaded registers (and the waits that go with them) beside the MFMAs
  const bf16x2 a0 = __builtin_convertvector(f32x2{x, y}, bf16x2);
  p0 = __builtin_bit_cast(unsigned, a0);
  const float rx1 = x - __uint_as_float(p0 << 16), ry1 = y - __uint_as_float(p0 & 0xffff0000u);
  const bf16x2 a1 = __builtin_convertvector(f32x2{rx1, ry1}, bf16x2);
  p1 = __builtin_bit_cast(unsigned, a1);
  const float rx2 = rx1 - __uint_as_float(p1 << 16), ry2 = ry1 - __uint_as_float(p1 & 0xffff0000u);
  const bf16x2 a2 = __builtin_convertvector(f32x2{rx2, ry2}, bf16x2);
  p2 = __builtin_bit_cast(unsigned, a2);
}

// ---- staging: fp32 registers -> bf16 piece registers -> LDS ----------------------------------------------------
// One thread owns N4 float4 units of an operand tile.
//   k-contiguous operand (KC):   unit u = 4 k values of one row           -> pair j = (u = j/2, half j%2)
//   row-contiguous operand:      unit u = k index kg*N4+u of 4 rows       -> pair j = (row i = j/(N4/2), k pair j%(N4/2))
//                                (N4 == 1: four single values, one per row)
// A "pair" is two k-adjacent values of one row = one dword of each bf16 piece.
template <int N4, bool KC>
struct BfStage {
  static constexpr int NP = (KC || N4 >= 2) ? 2 * N4 : 4;  // pairs (dwords per piece) per thread
};

template <int N4, bool KC, bool ACT>
__device__ __forceinline__ void bf_convert_pair(int j, const u32x4 (&raw)[N4], const u32x4 (&yraw)[ACT ? N4 : 1], int act,
                                                unsigned (&pc)[3][BfStage<N4, KC>::NP]) {
  float x, y = 0.f, gx = 1.f, gy = 1.f;
  if (KC) {
    const int u = j / 2, h = j % 2;
    x = __uint_as_float(raw[u][2 * h]);
    y = __uint_as_float(raw[u][2 * h + 1]);
    if (ACT) {
      gx = mvk_act_grad_from_out(__uint_as_float(yraw[ACT ? u : 0][2 * h]), act);
      gy = mvk_act_grad_from_out(__uint_as_float(yraw[ACT ? u : 0][2 * h + 1]), act);
    }
  } else if (N4 >= 2) {
    constexpr int H = N4 / 2 > 0 ? N4 / 2 : 1;
    const int i = j / H, jj = j % H;
    x = __uint_as_float(raw[(2 * jj) % N4][i]);
    y = __uint_as_float(raw[(2 * jj + 1) % N4][i]);
    if (ACT) {
      gx = mvk_act_grad_from_out(__uint_as_float(yraw[ACT ? (2 * jj) % N4 : 0][i]), act);
      gy = mvk_act_grad_from_out(__uint_as_float(yraw[ACT ? (2 * jj + 1) % N4 : 0][i]), act);
    }
  } else {
    x = __uint_as_float(raw[0][j]);
    if (ACT) gx = mvk_act_grad_from_out(__uint_as_float(yraw[0][j]), act);
  }
  if (ACT) {
    x *= gx;
    y *= gy;
  }
  bf_split3(x, y, pc[0][j], pc[1][j], pc[2][j]);
}

template <int N4, bool KC, int G, int PIECE>
__device__ __forceinline__ void bf_write_pieces(char* base, const unsigned (&pc)[3][BfStage<N4, KC>::NP], int tid, int row4,
                                                int kg) {
  if (KC) {
#pragma unroll
    for (int u = 0; u < N4; ++u) {
      const int idx = tid + u * 256;
      const int row = idx / 8, kq = (idx % 8) * 4;
      char* dst = base + (row & 3) * G + (row >> 2) * 80 + kq * 2;
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2*>(dst + p * PIECE) = u32x2{pc[p][2 * u], pc[p][2 * u + 1]};
    }
  } else {
    constexpr int H = N4 / 2 > 0 ? N4 / 2 : 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      char* dst = base + i * G + row4 * 80 + kg * N4 * 2;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        if (N4 == 1) {
          *reinterpret_cast<unsigned short*>(dst + p * PIECE) = (unsigned short)pc[p][i];
        } else if (N4 == 2) {
          *reinterpret_cast<unsigned*>(dst + p * PIECE) = pc[p][i * H];
        } else if (N4 == 4) {
          *reinterpret_cast<u32x2*>(dst + p * PIECE) = u32x2{pc[p][i * H], pc[p][i * H + 1 % H]};
        } else {
          *reinterpret_cast<u32x4*>(dst + p * PIECE) =
              u32x4{pc[p][i * H], pc[p][i * H + 1 % H], pc[p][i * H + 2 % H], pc[p][i * H + 3 % H]};
        }
      }
    }
  }
}

// The kernel's body with the workgroup's position as arguments: igemm_bf_kernel passes its blockIdx, igemm_bf_pair_kernel (two
// independent problems in ONE grid) the position inside the problem a workgroup belongs to.
template <int BM, int BN, int AMODE, int BMODE, bool AACT, int DEPTH>
__device__ __forceinline__ void igemm_bf_body(const GemmDesc& d, const unsigned bid_x, const unsigned bid_y, const unsigned bid_z) {
  using T = BfCfg<BM, BN>;
  constexpr int BKT = T::BKT;
  __shared__ __attribute__((aligned(16))) char lds_raw[T::LDS_BYTES];
  char* As = lds_raw;
  char* Bs = lds_raw + 3 * T::A_PIECE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / T::WAVES_N, wn = wave % T::WAVES_N;
  const int m0 = bid_x * BM, n0 = bid_y * BN;
  const int l31 = lane & 31, lhi = lane >> 5;
  const AOperand& A = d.a;
  const BOperand& B = d.b;
  constexpr bool A_KC = (AMODE == AM_PLAIN_K || AMODE == AM_ROW);  // memory contiguous along k
  // Pre-split A operand (AM_ROW3): the tensor is stored as 3 bf16 planes, written once by its producer; a unit is
  // 8 consecutive k (16 bytes) of one row per plane and goes to LDS as loaded — no VALU work in the loop.
  constexpr bool A_PRE = (AMODE == AM_ROW3);
  constexpr int NA8 = (BM * BKT / 8 / 256) > 0 ? (BM * BKT / 8 / 256) : 1;
  constexpr int NRA = A_PRE ? 3 * NA8 : T::NA4;  // raw A registers (16 bytes each) per stage
  constexpr bool B_KC = (BMODE == BM_K);

  int ph = 0, pw = 0;
  int kbeg = 0, kend = d.K;
  const float* bp = B.p;
  if (d.zmode == Z_PARITY) {
    ph = bid_z >> 1;
    pw = bid_z & 1;
    bp += (long long)bid_z * B.z_stride;
  } else if (d.zmode == Z_SPLITK) {
    kbeg = bid_z * d.ksplit_tiles * BK;  // launch keeps ksplit_tiles (16-wide) a multiple of 2
    const int e = kbeg + d.ksplit_tiles * BK;
    kend = e < kend ? e : kend;
    if (kbeg >= kend) return;
  }
  const int ntiles = (kend - kbeg + BKT - 1) / BKT;

  f32x16 acc[T::TM][T::TN];
#pragma unroll
  for (int a = 0; a < T::TM; ++a)
#pragma unroll
    for (int b = 0; b < T::TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  f32x16 acc2;  // second accumulator of single-tile waves (see step)
#pragma unroll
  for (int r = 0; r < 16; ++r) acc2[r] = 0.f;

  // buffer loads with byte offset 0x7fffffff are past num_records and return 0 (see igemm_fast.hpp)
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A.p, 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY =
      __builtin_amdgcn_make_buffer_rsrc((void*)(AACT ? A.act_src : A.p), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)bp, 0, 0x7ffffff0, 0x00020000);
  const char* a_bytes = reinterpret_cast<const char*>(A.p);
  const __amdgpu_buffer_rsrc_t rsA1 =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a_bytes + (A_PRE ? A.plane_bytes : 0)), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA2 =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a_bytes + (A_PRE ? 2 * A.plane_bytes : 0)), 0, 0x7ffffff0, 0x00020000);

  // ---- per-thread unit geometry ---------------------------------------------------------------------------
  // k-contiguous operand: unit idx = tid + 256 u -> row idx / 8, k quad (idx % 8) * 4
  // row-contiguous operand: rows 4 * (tid % (R/4)) .. +3, k = (tid / (R/4)) * N4 + u
  const int a_row4 = tid % (BM / 4), a_kg = tid / (BM / 4);
  const int b_row4 = tid % (BN / 4), b_kg = tid / (BN / 4);
  int abase[T::NA4], pa[T::NA4], pb[T::NA4], pc[T::NA4];
  int col_th = -(1 << 20), col_tw = 0, col_ch = 0;  // AM_COL: tap row / column offset and channel of this thread's rows
#pragma unroll
  for (int u = 0; u < T::NA4; ++u) {
    const int idx = tid + u * 256;
    abase[u] = -1;
    pa[u] = -1;
    pb[u] = pc[u] = 0;
    if (AMODE == AM_PLAIN_K) {
      const int r = m0 + idx / 8;
      if (r < d.M) abase[u] = r * (int)A.sr + (idx % 8) * 4;
    } else if (AMODE == AM_PLAIN_R) {
      const int r = m0 + a_row4 * 4;
      if (r < d.M) abase[u] = (a_kg * T::NA4 + u) * (int)A.sk + r;
    } else if (AMODE == AM_ROW3) {
      pb[u] = -(1 << 20);
      if (u < NA8) {
        const int r = m0 + idx / 4;  // unit = (row, k octet)
        if (r < d.M) {
          Pos ps = decode_pos(r, A.OH, A.OW);
          const int mul = (A.kind == A_UP) ? 1 : a_mul(A);
          pb[u] = mul * ps.i;
          pc[u] = mul * ps.j;
          pa[u] = (((ps.n * A.H + pb[u]) * A.W + pc[u]) * A.C + (idx % 4) * 8) * 2;  // bytes (bf16)
        }
      }
    } else if (AMODE == AM_ROW) {
      // hoisted gather geometry: pa = byte offset of tap (0,0) at this unit's k quad, pb / pc = mul*i, mul*j
      const int r = m0 + idx / 8;
      pb[u] = -(1 << 20);  // invalid row: every bounds test fails
      if (r < d.M) {
        Pos ps = decode_pos(r, A.OH, A.OW);
        const int mul = (A.kind == A_UP) ? 1 : a_mul(A);
        pb[u] = mul * ps.i;
        pc[u] = mul * ps.j;
        pa[u] = (((ps.n * A.H + pb[u]) * A.W + pc[u]) * A.C + (idx % 8) * 4) * 4;
      }
    } else {  // AM_COL: rows = (tap, channel)
      const int r = m0 + a_row4 * 4;
      if (r < d.M) {
        const int tap = r / A.C;
        pa[u] = tap;
        pb[u] = r - tap * A.C;
        int kh, kw;
        a_tap(A, tap, kh, kw);
        col_th = kh - 1;
        col_tw = kw - 1;
        col_ch = pb[u];
      }
    }
  }
  int bbase[T::NB4];
#pragma unroll
  for (int u = 0; u < T::NB4; ++u) {
    const int idx = tid + u * 256;
    bbase[u] = -1;
    if (BMODE == BM_K) {
      const int n = n0 + idx / 8;
      if (n < d.N) bbase[u] = n * (int)B.sn + (idx % 8) * 4;
    } else {
      const int n = n0 + b_row4 * 4;
      if (n < d.N) bbase[u] = (b_kg * T::NB4 + u) * (int)B.sk + n;
    }
  }
  const int ow_sh = ((A.OW & (A.OW - 1)) == 0) ? __builtin_ctz(A.OW > 0 ? A.OW : 1) : -1;
  const int oh_sh = ((A.OH & (A.OH - 1)) == 0) ? __builtin_ctz(A.OH > 0 ? A.OH : 1) : -1;
  const bool col_fast = ow_sh >= 0 && oh_sh >= 0 && (A.OW % T::NA4) == 0;

  constexpr int NYA = AACT ? T::NA4 : 1;
  // DEPTH raw (fp32) register stages: the loads run DEPTH k-tiles ahead of the MFMAs.  They are issued
  // unconditionally (past the end of K they are out-of-range = zero-fill, no traffic): a conditional load would
  // make the compiler's vmcnt bookkeeping fall back to waiting for the newest stage.
  u32x4 ra[DEPTH][NRA], ya[DEPTH][NYA], rb[DEPTH][T::NB4];

  auto load_tiles = [&](int k0, u32x4 (&ra)[NRA], u32x4 (&ya)[NYA], u32x4 (&rb)[T::NB4]) {
    auto sel = [](bool ok, int off) { return __builtin_unpredictable(ok) ? off : 0x7fffffff; };  // keep it a v_cndmask
#ifdef MVK_EXPER
    if (d.dbg_flags & 2) {  // experiment: no global traffic
      for (int u = 0; u < NRA; ++u) ra[u] = u32x4{1, 2, 3, 4};
      for (int u = 0; u < T::NB4; ++u) rb[u] = u32x4{1, 2, 3, 4};
      return;
    }
    const bool skip_a = d.dbg_flags & 64, skip_b = d.dbg_flags & 128;
    if (skip_a) for (int u = 0; u < NRA; ++u) ra[u] = u32x4{1, 2, 3, 4};
    if (skip_b) for (int u = 0; u < T::NB4; ++u) rb[u] = u32x4{1, 2, 3, 4};
#elif defined(MVK_X_NOA) || defined(MVK_X_NOB)  // compile-time experiments (tools/build_variants.sh): schedule undisturbed
#ifdef MVK_X_NOA
    constexpr bool skip_a = true;
    for (int u = 0; u < NRA; ++u) ra[u] = u32x4{1, 2, 3, 4};
#else
    constexpr bool skip_a = false;
#endif
#ifdef MVK_X_NOB
    constexpr bool skip_b = true;
    for (int u = 0; u < T::NB4; ++u) rb[u] = u32x4{1, 2, 3, 4};
#else
    constexpr bool skip_b = false;
#endif
#else
    constexpr bool skip_a = false, skip_b = false;
#endif
    if (skip_a) {
    } else if (AMODE == AM_PLAIN_K) {
#pragma unroll
      for (int u = 0; u < T::NA4; ++u) {
        const int idx = tid + u * 256;
        const bool ok = abase[u] >= 0 && (k0 + (idx % 8) * 4) < kend;
        const int off = sel(ok, (abase[u] + k0) * 4);
        ra[u] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
        if (AACT) ya[u] = __builtin_amdgcn_raw_buffer_load_b128(rsY, off, 0, 0);
      }
    } else if (AMODE == AM_PLAIN_R) {
#pragma unroll
      for (int u = 0; u < T::NA4; ++u) {
        const bool ok = abase[u] >= 0 && (k0 + a_kg * T::NA4 + u) < kend;
        const int off = sel(ok, (abase[u] + k0 * (int)A.sk) * 4);
        ra[u] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
        if (AACT) ya[u] = __builtin_amdgcn_raw_buffer_load_b128(rsY, off, 0, 0);
      }
    } else if (AMODE == AM_ROW3) {
      const int tap = k0 / A.C;  // block-uniform: C % 32 == 0
      const int c0 = k0 - tap * A.C;
      int dh, dw;
      if (A.kind == A_UP) {
        dh = ph - (tap >> 1);
        dw = pw - (tap & 1);
      } else {
        int kh_, kw_;
        a_tap(A, tap, kh_, kw_);
        dh = kh_ - 1;
        dw = kw_ - 1;
      }
      if (k0 >= kend) dh = -(1 << 20);
      const int delta = ((dh * A.W + dw) * A.C + c0) * 2;
#pragma unroll
      for (int u = 0; u < NA8; ++u) {
        const bool ok = (unsigned)(pb[u] + dh) < (unsigned)A.H && (unsigned)(pc[u] + dw) < (unsigned)A.W;
        const int off = sel(ok, pa[u] + delta);
        ra[(3 * u) % NRA] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
        ra[(3 * u + 1) % NRA] = __builtin_amdgcn_raw_buffer_load_b128(rsA1, off, 0, 0);
        ra[(3 * u + 2) % NRA] = __builtin_amdgcn_raw_buffer_load_b128(rsA2, off, 0, 0);
      }
    } else if (AMODE == AM_ROW) {
      const int tap = k0 / A.C;  // block-uniform: C % 32 == 0
      const int c0 = k0 - tap * A.C;
      int dh, dw;
      if (A.kind == A_UP) {
        dh = ph - (tap >> 1);
        dw = pw - (tap & 1);
      } else {
        int kh_, kw_;
        a_tap(A, tap, kh_, kw_);
        dh = kh_ - 1;
        dw = kw_ - 1;
      }
      if (k0 >= kend) dh = -(1 << 20);  // prefetch past the end: all out of range
      const int delta = ((dh * A.W + dw) * A.C + c0) * 4;
#pragma unroll
      for (int u = 0; u < T::NA4; ++u) {
        const bool ok = (unsigned)(pb[u] + dh) < (unsigned)A.H && (unsigned)(pc[u] + dw) < (unsigned)A.W;
        ra[u] = __builtin_amdgcn_raw_buffer_load_b128(rsA, sel(ok, pa[u] + delta), 0, 0);
      }
    } else {  // AM_COL: k = output position; this thread's units are NA4 consecutive positions
      const int pos0 = k0 + a_kg * T::NA4;
      if (col_fast) {  // all of them in one output row: decode once
        const int pj0 = pos0 & (A.OW - 1);
        const int t = pos0 >> ow_sh;
        const int pi = t & (A.OH - 1), pn = t >> oh_sh;
        const int hh = a_mul(A) * pi + col_th;
        const bool hok = (unsigned)hh < (unsigned)A.H;
        const int rowoff = (pn * A.H + hh) * A.W * A.C + col_ch;
#pragma unroll
        for (int u = 0; u < T::NA4; ++u) {
          const int ww = a_mul(A) * (pj0 + u) + col_tw;
          const bool ok = hok && (unsigned)ww < (unsigned)A.W && (pos0 + u) < kend;
          ra[u] = __builtin_amdgcn_raw_buffer_load_b128(rsA, sel(ok, (rowoff + ww * A.C) * 4), 0, 0);
        }
      } else {
#pragma unroll
        for (int u = 0; u < T::NA4; ++u) {
          const int pos = pos0 + u;
          const int pj = pos % A.OW;
          const int t = pos / A.OW;
          const int pi = t % A.OH, pn = t / A.OH;
          const int hh = a_mul(A) * pi + col_th, ww = a_mul(A) * pj + col_tw;
          const bool ok = pos < kend && (unsigned)hh < (unsigned)A.H && (unsigned)ww < (unsigned)A.W;
          ra[u] = __builtin_amdgcn_raw_buffer_load_b128(rsA, sel(ok, (((pn * A.H + hh) * A.W + ww) * A.C + col_ch) * 4), 0, 0);
        }
      }
    }
    if (skip_b) {
    } else if (BMODE == BM_K) {
#pragma unroll
      for (int u = 0; u < T::NB4; ++u) {
        const int idx = tid + u * 256;
        const bool ok = bbase[u] >= 0 && (k0 + (idx % 8) * 4) < kend;
        rb[u] = __builtin_amdgcn_raw_buffer_load_b128(rsB, sel(ok, (bbase[u] + k0) * 4), 0, 0);
      }
    } else {
#pragma unroll
      for (int u = 0; u < T::NB4; ++u) {
        const bool ok = bbase[u] >= 0 && (k0 + b_kg * T::NB4 + u) < kend;
        rb[u] = __builtin_amdgcn_raw_buffer_load_b128(rsB, sel(ok, (bbase[u] + k0 * (int)B.sk) * 4), 0, 0);
      }
    }
  };

  // fragment addresses: row = wave tile + 32 a + (l & 31)  ->  (row & 3) * G + (row >> 2) * 80
  const char* a_frag = As + (l31 & 3) * T::GA + ((wm * T::WTM + l31) >> 2) * 80 + lhi * 16;
  const char* b_frag = Bs + (l31 & 3) * T::GB + ((wn * T::WTN + l31) >> 2) * 80 + lhi * 16;

  using SA_ = BfStage<T::NA4, A_KC>;
  using SB_ = BfStage<T::NB4, B_KC>;
  constexpr int SA_NP = A_PRE ? 0 : SA_::NP;
  constexpr int NPAIR = SA_NP + SB_::NP;  // conversion work items per thread and k-tile
  constexpr int NSLOT = (BKT / 16) * 6;     // MFMA groups per k-tile
  unsigned pca[3][SA_::NP], pcb[3][SB_::NP];  // pca unused for a pre-split A
  const u32x4 ynul[1] = {u32x4{0, 0, 0, 0}};

  auto convert_range = [&](int lo, int hi, const u32x4 (&ra)[NRA], const u32x4 (&ya)[NYA], const u32x4 (&rb)[T::NB4]) {
#pragma unroll
    for (int j = lo; j < hi; ++j) {
      if (j < SA_NP) {
        if (!A_PRE) bf_convert_pair<T::NA4, A_KC, AACT>(j, reinterpret_cast<const u32x4(&)[T::NA4]>(ra), ya, A.act, pca);
      } else {
        bf_convert_pair<T::NB4, B_KC, false>(j - SA_NP, rb, ynul, 0, pcb);
      }
    }
  };
  auto write_pieces = [&](const u32x4 (&ra)[NRA]) {
    if (A_PRE) {
#pragma unroll
      for (int u = 0; u < NA8; ++u) {
        const int idx = tid + u * 256;
        const int row = idx / 4;
        char* dst = As + (row & 3) * T::GA + (row >> 2) * 80 + (idx % 4) * 16;
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(dst + p * T::A_PIECE) = ra[(3 * u + p) % NRA];
      }
    } else {
      bf_write_pieces<T::NA4, A_KC, T::GA, T::A_PIECE>(As, pca, tid, a_row4, a_kg);
    }
    bf_write_pieces<T::NB4, B_KC, T::GB, T::B_PIECE>(Bs, pcb, tid, b_row4, b_kg);
  };

  // One k-tile: MFMAs over the tile in LDS, with the split of the NEXT tile's registers spread between the MFMA
  // groups (VALU work runs beside the matrix pipe), then barrier / piece write / barrier.
#ifdef MVK_PHASES
  unsigned long long tphase[5] = {0, 0, 0, 0, 0};
#define MVK_T(i) { const unsigned long long now_ = __builtin_readcyclecounter(); tphase[i] += now_ - tlast; tlast = now_; }
#else
#define MVK_T(i)
#endif
  auto step = [&](const u32x4 (&ra)[NRA], const u32x4 (&ya)[NYA], const u32x4 (&rb)[T::NB4]) {
#ifdef MVK_PHASES
    unsigned long long tlast = __builtin_readcyclecounter();
#endif
#ifdef MVK_EXPER
    const bool no_mfma = d.dbg_flags & 32, no_write = d.dbg_flags & 16;
#else
    constexpr bool no_mfma = false, no_write = false;
#endif
    constexpr int PA[6] = {0, 1, 2, 0, 1, 0};  // smallest terms first
    constexpr int PB[6] = {2, 1, 0, 1, 0, 0};
    if (T::TM * T::TN == 1) {
      // one MFMA tile per wave: alternate the two k-steps on two accumulators so consecutive MFMAs are independent
      bf16x8 af[2][3], bfr[2][3];
      if (!no_mfma) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            af[ks][p] = *reinterpret_cast<const bf16x8*>(a_frag + p * T::A_PIECE + ks * 32);
            bfr[ks][p] = *reinterpret_cast<const bf16x8*>(b_frag + p * T::B_PIECE + ks * 32);
          }
      }
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          if (!no_mfma) {
            if (ks == 0) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][PA[q]], bfr[0][PB[q]], acc[0][0], 0, 0, 0);
            else acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][PA[q]], bfr[1][PB[q]], acc2, 0, 0, 0);
          }
          const int slot = q * 2 + ks;
          if (!no_write) convert_range(slot * NPAIR / NSLOT, (slot + 1) * NPAIR / NSLOT, ra, ya, rb);
        }
    } else {
#pragma unroll
      for (int ks = 0; ks < BKT / 16; ++ks) {
        bf16x8 af[T::TM][3], bfr[T::TN][3];
        if (!no_mfma) {
#pragma unroll
          for (int a = 0; a < T::TM; ++a)
#pragma unroll
            for (int p = 0; p < 3; ++p)
              af[a][p] = *reinterpret_cast<const bf16x8*>(a_frag + p * T::A_PIECE + a * 8 * 80 + ks * 32);
#pragma unroll
          for (int b = 0; b < T::TN; ++b)
#pragma unroll
            for (int p = 0; p < 3; ++p)
              bfr[b][p] = *reinterpret_cast<const bf16x8*>(b_frag + p * T::B_PIECE + b * 8 * 80 + ks * 32);
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          if (!no_mfma) {
#pragma unroll
            for (int a = 0; a < T::TM; ++a)
#pragma unroll
              for (int b = 0; b < T::TN; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][PA[q]], bfr[b][PB[q]], acc[a][b], 0, 0, 0);
          }
          const int slot = ks * 6 + q;
          if (!no_write) convert_range(slot * NPAIR / NSLOT, (slot + 1) * NPAIR / NSLOT, ra, ya, rb);
        }
      }
    }
    MVK_T(0)
    __syncthreads();  // every wave is done reading the tile
    MVK_T(1)
    if (!no_write) write_pieces(ra);
    MVK_T(2)
#ifdef MVK_EXPER
    if (no_write) {  // keep the loads alive
      unsigned s_ = 0;
      for (int u = 0; u < T::NA4; ++u) s_ += ra[u].x ^ ra[u].y ^ ra[u].z ^ ra[u].w;
      for (int u = 0; u < T::NB4; ++u) s_ += rb[u].x ^ rb[u].y ^ rb[u].z ^ rb[u].w;
      if (s_ == 0x12345677u) As[tid] = 1;
    }
#endif
    __syncthreads();
    MVK_T(3)
  };

  // prologue: tile 0 -> LDS, tiles 1 .. DEPTH-1 -> registers (in flight)
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) load_tiles(kbeg + i * BKT, ra[i], ya[i], rb[i]);
  convert_range(0, NPAIR, ra[0], ya[0], rb[0]);
  write_pieces(ra[0]);
  __syncthreads();
  int t = 0;
  for (; t + DEPTH <= ntiles; t += DEPTH) {  // full groups only: no exit inside the unrolled body (keeps the
#pragma unroll                               // register stages static: no copies, no waits on the back edge)
    for (int i = 0; i < DEPTH; ++i) {
      // LDS: tile t+i.  Stage i is free: fetch tile t+i+DEPTH into it; split tile t+i+1 (stage i+1) beside the MFMAs.
      load_tiles(kbeg + (t + i + DEPTH) * BKT, ra[i], ya[i], rb[i]);
      __builtin_amdgcn_sched_barrier(0);  // keep the loads at the top of the step: the scheduler would sink them
      step(ra[(i + 1) % DEPTH], ya[(i + 1) % DEPTH], rb[(i + 1) % DEPTH]);
    }
  }
#pragma unroll
  for (int i = 0; i < DEPTH - 1; ++i)  // remaining ntiles % DEPTH tiles (their successors are zero tiles)
    if (t + i < ntiles) step(ra[(i + 1) % DEPTH], ya[(i + 1) % DEPTH], rb[(i + 1) % DEPTH]);
  if (T::TM * T::TN == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] += acc2[r];
  }
#ifdef MVK_PHASES
  if (d.dbg && lane == 0) {
    for (int i = 0; i < 4; ++i) atomicAdd(d.dbg + i, tphase[i]);
    atomicAdd(d.dbg + 5, (unsigned long long)ntiles);
    atomicAdd(d.dbg + 7, 1ull);
  }
#endif
  float* lds = reinterpret_cast<float*>(lds_raw);
  if (run_epilogue_vec<T, BM, BN, T::LDS_BYTES / 4>(d, acc, lds, tid, m0, n0, wm, wn, l31, lhi, ph, pw)) return;
  run_epilogue<T>(d, acc, lds, tid, m0, n0, wm, wn, l31, lhi, ph, pw, (int)bid_z);
}

template <int BM, int BN, int AMODE, int BMODE, bool AACT, int DEPTH>
__global__ __launch_bounds__(256, (BM * BN <= 128 * 32) ? MVK_BF_OCC_SMALL : ((BM * BN <= 128 * 64) ? 3 : 2)) void igemm_bf_kernel(const GemmDesc d) {
  igemm_bf_body<BM, BN, AMODE, BMODE, AACT, DEPTH>(d, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Two independent split-K problems of the same kernel configuration in ONE launch (the two 4x4 / stride-2 weight gradients of the
// convolutional encoder at the training batch: 256 workgroups each, two dependent launches of ~35 us on the step's last chain;
// three workgroups fit a CU, so one grid of 512 runs them side by side).  grid.x = n0 + n1 workgroups; the first n0 belong to d0.
template <int BM, int BN, int AMODE, int BMODE, bool AACT, int DEPTH>
__global__ __launch_bounds__(256, (BM * BN <= 128 * 32) ? MVK_BF_OCC_SMALL : ((BM * BN <= 128 * 64) ? 3 : 2)) void igemm_bf_pair_kernel(
    const GemmDesc d0, const GemmDesc d1, const unsigned n0, const unsigned gx0, const unsigned gy0, const unsigned gx1,
    const unsigned gy1) {
  unsigned b = blockIdx.x;
  if (b < n0) {
    igemm_bf_body<BM, BN, AMODE, BMODE, AACT, DEPTH>(d0, b % gx0, (b / gx0) % gy0, b / (gx0 * gy0));
  } else {
    b -= n0;
    igemm_bf_body<BM, BN, AMODE, BMODE, AACT, DEPTH>(d1, b % gx1, (b / gx1) % gy1, b / (gx1 * gy1));
  }
}

}  // namespace mvk
