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
#include "igemm_fast.hpp"

namespace mvk {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

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
__device__ __forceinline__ void bf_split3(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
  const f32x2 v = {x, y};
  const bf16x2 a0 = __builtin_convertvector(v, bf16x2);
  const f32x2 r1 = v - __builtin_convertvector(a0, f32x2);
  const bf16x2 a1 = __builtin_convertvector(r1, bf16x2);
  const f32x2 r2 = r1 - __builtin_convertvector(a1, f32x2);
  const bf16x2 a2 = __builtin_convertvector(r2, bf16x2);
  p0 = __builtin_bit_cast(unsigned, a0);
  p1 = __builtin_bit_cast(unsigned, a1);
  p2 = __builtin_bit_cast(unsigned, a2);
}

// Stage the N4 float4 units of one thread (row-contiguous operand: unit u = k index kg*N4+u, 4 consecutive rows)
// into the LDS image: for each of the 4 rows, N4 consecutive k values per piece.
template <int N4, int G, int PIECE>
__device__ __forceinline__ void bf_store_rows(char* base, const f32x4 (&v)[N4], int row4, int kg) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    char* dst = base + i * G + row4 * 80 + kg * N4 * 2;
    if (N4 == 1) {
      unsigned p0, p1, p2;
      bf_split3(v[0][i], 0.f, p0, p1, p2);
      *reinterpret_cast<unsigned short*>(dst) = (unsigned short)p0;
      *reinterpret_cast<unsigned short*>(dst + PIECE) = (unsigned short)p1;
      *reinterpret_cast<unsigned short*>(dst + 2 * PIECE) = (unsigned short)p2;
    } else {
      constexpr int NP = N4 / 2 > 0 ? N4 / 2 : 1;
      unsigned q0[NP], q1[NP], q2[NP];
#pragma unroll
      for (int j = 0; j < N4 / 2; ++j) bf_split3(v[2 * j][i], v[(2 * j + 1) % N4][i], q0[j], q1[j], q2[j]);
      if (N4 == 2) {
        *reinterpret_cast<unsigned*>(dst) = q0[0];
        *reinterpret_cast<unsigned*>(dst + PIECE) = q1[0];
        *reinterpret_cast<unsigned*>(dst + 2 * PIECE) = q2[0];
      } else if (N4 == 4) {
        *reinterpret_cast<u32x2*>(dst) = u32x2{q0[0], q0[1 % NP]};
        *reinterpret_cast<u32x2*>(dst + PIECE) = u32x2{q1[0], q1[1 % NP]};
        *reinterpret_cast<u32x2*>(dst + 2 * PIECE) = u32x2{q2[0], q2[1 % NP]};
      } else {  // 8
        *reinterpret_cast<u32x4*>(dst) = u32x4{q0[0], q0[1 % NP], q0[2 % NP], q0[3 % NP]};
        *reinterpret_cast<u32x4*>(dst + PIECE) = u32x4{q1[0], q1[1 % NP], q1[2 % NP], q1[3 % NP]};
        *reinterpret_cast<u32x4*>(dst + 2 * PIECE) = u32x4{q2[0], q2[1 % NP], q2[2 % NP], q2[3 % NP]};
      }
    }
  }
}

// k-contiguous operand: one float4 = 4 k values of one row
template <int G, int PIECE>
__device__ __forceinline__ void bf_store_k4(char* base, const f32x4& v, int row, int kq) {
  unsigned a0, a1, a2, b0, b1, b2;
  bf_split3(v[0], v[1], a0, a1, a2);
  bf_split3(v[2], v[3], b0, b1, b2);
  char* dst = base + (row & 3) * G + (row >> 2) * 80 + kq * 2;
  *reinterpret_cast<u32x2*>(dst) = u32x2{a0, b0};
  *reinterpret_cast<u32x2*>(dst + PIECE) = u32x2{a1, b1};
  *reinterpret_cast<u32x2*>(dst + 2 * PIECE) = u32x2{a2, b2};
}

template <int BM, int BN, int AMODE, int BMODE, bool AACT>
__global__ __launch_bounds__(256, 2) void igemm_bf_kernel(const GemmDesc d) {
  using T = BfCfg<BM, BN>;
  constexpr int BKT = T::BKT;
  __shared__ __attribute__((aligned(16))) char lds_raw[T::LDS_BYTES];
  char* As = lds_raw;
  char* Bs = lds_raw + 3 * T::A_PIECE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / T::WAVES_N, wn = wave % T::WAVES_N;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int l31 = lane & 31, lhi = lane >> 5;
  const AOperand& A = d.a;
  const BOperand& B = d.b;
  constexpr bool A_KC = (AMODE == AM_PLAIN_K || AMODE == AM_ROW);  // memory contiguous along k
  constexpr bool B_KC = (BMODE == BM_K);

  int ph = 0, pw = 0;
  int kbeg = 0, kend = d.K;
  const float* bp = B.p;
  if (d.zmode == Z_PARITY) {
    ph = blockIdx.z >> 1;
    pw = blockIdx.z & 1;
    bp += (long long)blockIdx.z * B.z_stride;
  } else if (d.zmode == Z_SPLITK) {
    kbeg = blockIdx.z * d.ksplit_tiles * BK;  // launch keeps ksplit_tiles (16-wide) a multiple of 2
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

  constexpr int OOB = 0x7fffffff;  // buffer loads past num_records return 0 (see igemm_fast.hpp)
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A.p, 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY =
      __builtin_amdgcn_make_buffer_rsrc((void*)(AACT ? A.act_src : A.p), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)bp, 0, 0x7ffffff0, 0x00020000);

  // ---- per-thread unit geometry ---------------------------------------------------------------------------
  // k-contiguous operand: unit idx = tid + 256 u -> row idx / 8, k quad (idx % 8) * 4
  // row-contiguous operand: rows 4 * (tid % (R/4)) .. +3, k = (tid / (R/4)) * N4 + u
  const int a_row4 = tid % (BM / 4), a_kg = tid / (BM / 4);
  const int b_row4 = tid % (BN / 4), b_kg = tid / (BN / 4);
  int abase[T::NA4], pa[T::NA4], pb[T::NA4], pc[T::NA4];
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
    } else if (AMODE == AM_ROW) {
      const int r = m0 + idx / 8;
      if (r < d.M) {
        Pos ps = decode_pos(r, A.OH, A.OW);
        pa[u] = ps.n;
        pb[u] = ps.i;
        pc[u] = ps.j;
      }
    } else {  // AM_COL: rows = (tap, channel)
      const int r = m0 + a_row4 * 4;
      if (r < d.M) {
        const int tap = r / A.C;
        pa[u] = tap;
        pb[u] = r - tap * A.C;
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

  u32x4 ra[T::NA4], ya[AACT ? T::NA4 : 1], rb[T::NB4];

  auto load_tiles = [&](int k0) {
    if (AMODE == AM_PLAIN_K) {
#pragma unroll
      for (int u = 0; u < T::NA4; ++u) {
        const int idx = tid + u * 256;
        const bool ok = abase[u] >= 0 && (k0 + (idx % 8) * 4) < kend;
        const int off = ok ? (abase[u] + k0) * 4 : OOB;
        ra[u] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
        if (AACT) ya[u] = __builtin_amdgcn_raw_buffer_load_b128(rsY, off, 0, 0);
      }
    } else if (AMODE == AM_PLAIN_R) {
#pragma unroll
      for (int u = 0; u < T::NA4; ++u) {
        const bool ok = abase[u] >= 0 && (k0 + a_kg * T::NA4 + u) < kend;
        const int off = ok ? (abase[u] + k0 * (int)A.sk) * 4 : OOB;
        ra[u] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
        if (AACT) ya[u] = __builtin_amdgcn_raw_buffer_load_b128(rsY, off, 0, 0);
      }
    } else if (AMODE == AM_ROW) {
      const int tap = k0 / A.C;  // block-uniform: C % 32 == 0
      const int c0 = k0 - tap * A.C;
      int dh, dw;
      if (A.kind == A_UP) {
        dh = ph - (tap >> 1);
        dw = pw - (tap & 1);
      } else {
        dh = (tap >> 2) - 1;
        dw = (tap & 3) - 1;
      }
      const int mul = (A.kind == A_UP) ? 1 : 2;
#pragma unroll
      for (int u = 0; u < T::NA4; ++u) {
        const int idx = tid + u * 256;
        const int kq = (idx % 8) * 4;
        const int hh = mul * pb[u] + dh, ww = mul * pc[u] + dw;
        const bool ok = pa[u] >= 0 && (k0 + kq) < kend && hh >= 0 && hh < A.H && ww >= 0 && ww < A.W;
        const int off = ok ? (((pa[u] * A.H + hh) * A.W + ww) * A.C + c0 + kq) * 4 : OOB;
        ra[u] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
      }
    } else {  // AM_COL: k = output position
#pragma unroll
      for (int u = 0; u < T::NA4; ++u) {
        const int pos = k0 + a_kg * T::NA4 + u;
        int pj, pi, pn;
        if (ow_sh >= 0 && oh_sh >= 0) {
          pj = pos & (A.OW - 1);
          const int t = pos >> ow_sh;
          pi = t & (A.OH - 1);
          pn = t >> oh_sh;
        } else {
          pj = pos % A.OW;
          const int t = pos / A.OW;
          pi = t % A.OH;
          pn = t / A.OH;
        }
        const int hh = 2 * pi - 1 + (pa[u] >> 2), ww = 2 * pj - 1 + (pa[u] & 3);
        const bool ok = pa[u] >= 0 && pos < kend && hh >= 0 && hh < A.H && ww >= 0 && ww < A.W;
        const int off = ok ? (((pn * A.H + hh) * A.W + ww) * A.C + pb[u]) * 4 : OOB;
        ra[u] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
      }
    }
    if (BMODE == BM_K) {
#pragma unroll
      for (int u = 0; u < T::NB4; ++u) {
        const int idx = tid + u * 256;
        const bool ok = bbase[u] >= 0 && (k0 + (idx % 8) * 4) < kend;
        rb[u] = __builtin_amdgcn_raw_buffer_load_b128(rsB, ok ? (bbase[u] + k0) * 4 : OOB, 0, 0);
      }
    } else {
#pragma unroll
      for (int u = 0; u < T::NB4; ++u) {
        const bool ok = bbase[u] >= 0 && (k0 + b_kg * T::NB4 + u) < kend;
        rb[u] = __builtin_amdgcn_raw_buffer_load_b128(rsB, ok ? (bbase[u] + k0 * (int)B.sk) * 4 : OOB, 0, 0);
      }
    }
  };

  // registers -> three bf16 pieces -> LDS
  auto store_tiles = [&]() {
    f32x4 va[T::NA4];
#pragma unroll
    for (int u = 0; u < T::NA4; ++u) {
      va[u] = __builtin_bit_cast(f32x4, ra[u]);
      if (AACT) {
        const f32x4 y = __builtin_bit_cast(f32x4, ya[u]);
#pragma unroll
        for (int i = 0; i < 4; ++i) va[u][i] *= mvk_act_grad_from_out(y[i], A.act);
      }
    }
    if (A_KC) {
#pragma unroll
      for (int u = 0; u < T::NA4; ++u) {
        const int idx = tid + u * 256;
        bf_store_k4<T::GA, T::A_PIECE>(As, va[u], idx / 8, (idx % 8) * 4);
      }
    } else {
      bf_store_rows<T::NA4, T::GA, T::A_PIECE>(As, va, a_row4, a_kg);
    }
    f32x4 vb[T::NB4];
#pragma unroll
    for (int u = 0; u < T::NB4; ++u) vb[u] = __builtin_bit_cast(f32x4, rb[u]);
    if (B_KC) {
#pragma unroll
      for (int u = 0; u < T::NB4; ++u) {
        const int idx = tid + u * 256;
        bf_store_k4<T::GB, T::B_PIECE>(Bs, vb[u], idx / 8, (idx % 8) * 4);
      }
    } else {
      bf_store_rows<T::NB4, T::GB, T::B_PIECE>(Bs, vb, b_row4, b_kg);
    }
  };

  // fragment addresses: row = wave tile + 32 a + (l & 31)  ->  (row & 3) * G + (row >> 2) * 80
  const char* a_frag = As + (l31 & 3) * T::GA + ((wm * T::WTM + l31) >> 2) * 80 + lhi * 16;
  const char* b_frag = Bs + (l31 & 3) * T::GB + ((wn * T::WTN + l31) >> 2) * 80 + lhi * 16;

  auto compute = [&]() {
#pragma unroll
    for (int ks = 0; ks < BKT / 16; ++ks) {
      bf16x8 af[T::TM][3], bfr[T::TN][3];
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
      // smallest terms first; consecutive MFMAs go to different accumulators
      constexpr int PA[6] = {0, 1, 2, 0, 1, 0};
      constexpr int PB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int a = 0; a < T::TM; ++a)
#pragma unroll
          for (int b = 0; b < T::TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][PA[q]], bfr[b][PB[q]], acc[a][b], 0, 0, 0);
    }
  };

  load_tiles(kbeg);
  store_tiles();
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const bool more = (t + 1) < ntiles;
    if (more) load_tiles(kbeg + (t + 1) * BKT);  // in flight across the MFMAs
    compute();
    __syncthreads();                            // every wave is done reading the tile
    if (more) store_tiles();
    __syncthreads();
  }
  float* lds = reinterpret_cast<float*>(lds_raw);
  if (run_epilogue_vec<T, BM, BN, T::LDS_BYTES / 4>(d, acc, lds, tid, m0, n0, wm, wn, l31, lhi, ph, pw)) return;
  run_epilogue<T>(d, acc, lds, tid, m0, n0, wm, wn, l31, lhi, ph, pw);
}

}  // namespace mvk
