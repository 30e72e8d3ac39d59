// Mode-specialised, software-pipelined variant of the fp32 MFMA implicit-GEMM engine.
//
// igemm.hpp's generic kernel decides everything from runtime flags; the compiler then waits for every
// conditional global load right where it is issued and the loads never overlap the MFMAs.  Here the operand
// modes are template parameters, so one k-tile iteration is straight-line code:
//     issue ALL global loads of tile t+1 (16-byte loads, addresses from hoisted per-thread bases)
//     ds_read + MFMA over tile t                      <- loads in flight
//     wait, (optional activation-derivative multiply), write tile t+1 to the other LDS buffer, barrier
// BKT = 32 halves the number of barriers per MFMA where the LDS budget allows.
#pragma once
#include "igemm.hpp"

namespace mvk {

enum AMode { AM_PLAIN_K = 1, AM_PLAIN_R = 2, AM_ROW = 3, AM_COL = 4, AM_ROW3 = 5 };  // ROW3: AM_ROW on a pre-split (3 x bf16) tensor
enum BMode { BM_K = 1, BM_N = 2 };

template <int BM, int BN, int BKT, bool A_TRANSPOSED_WRITE = false, bool B_TRANSPOSED_WRITE = false>
struct FastCfg {
  static constexpr int WAVES_N = (BN >= 64) ? 2 : 1;
  static constexpr int WAVES_M = 4 / WAVES_N;
  static constexpr int WTM = BM / WAVES_M;
  static constexpr int WTN = BN / WAVES_N;
  static constexpr int TM = WTM / 32;
  static constexpr int TN = WTN / 32;
  // k-major LDS tiles.  Operands whose memory is contiguous along k are written transposed (4 x ds_write_b32 per
  // 16-byte unit): a row stride == 1 (mod 32) makes those writes bank-conflict-free (32 lanes = 8 k-groups x 4
  // rows -> banks 4*kq + row).  Operands contiguous along the row index are written with ds_write_b128 and need
  // a 16-byte aligned stride.  Fragment reads (32 consecutive rows of one k) are conflict-free for any stride.
  static constexpr int SA = A_TRANSPOSED_WRITE ? BM + 1 : BM + 4;
  static constexpr int SB = B_TRANSPOSED_WRITE ? BN + 1 : BN + 4;
  static constexpr int UA4 = BM * BKT / 4;
  static constexpr int UB4 = BN * BKT / 4;
  static constexpr int NA4 = (UA4 + 255) / 256;
  static constexpr int NB4 = (UB4 + 255) / 256;
  static_assert(TM >= 1 && TN >= 1, "tile too small");
};

template <int BM, int BN, int BKT, int AMODE, int BMODE, bool AACT>
__global__ __launch_bounds__(256, MVK_MIN_WAVES) void igemm_fast_kernel(const GemmDesc d) {
  using T = FastCfg<BM, BN, BKT, (AMODE == AM_PLAIN_K || AMODE == AM_ROW), (BMODE == BM_K)>;
  __shared__ __attribute__((aligned(16))) float lds[2 * BKT * (T::SA + T::SB) + 8];
  float* As = lds;
  float* Bs = lds + ((2 * BKT * T::SA + 3) & ~3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / T::WAVES_N, wn = wave % T::WAVES_N;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int l31 = lane & 31, lhi = lane >> 5;
  const AOperand& A = d.a;
  const BOperand& B = d.b;

  int ph = 0, pw = 0;
  int kbeg = 0, kend = d.K;
  const float* bp = B.p;
  if (d.zmode == Z_PARITY) {
    ph = blockIdx.z >> 1;
    pw = blockIdx.z & 1;
    bp += (long long)blockIdx.z * B.z_stride;
  } else if (d.zmode == Z_SPLITK) {
    kbeg = blockIdx.z * d.ksplit_tiles * BK;  // ksplit_tiles counts 16-wide tiles; launch keeps it a multiple of 2
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

  // ------------------------------------------------------------------------------------------------
  // hoisted per-unit state
  // ------------------------------------------------------------------------------------------------
  // Operands are read through buffer descriptors with hardware bounds checking: an out-of-range byte offset
  // (OOB) returns zeros, so padding / tile tails need neither branches nor pointer selects (a select between a
  // real address and a constant zero makes hipcc emit flat_load, whose lgkmcnt use serialises the LDS waits
  // of the MFMA phase with the global loads).  Offsets are 32-bit: every tensor on the path is < 2 GiB.
  constexpr int OOB = 0x7fffffff;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A.p, 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY =
      __builtin_amdgcn_make_buffer_rsrc((void*)(AACT ? A.act_src : A.p), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)bp, 0, 0x7ffffff0, 0x00020000);

  int abase[T::NA4];  // PLAIN: element offset without the k0 term; -1 = row out of range
  int pa[T::NA4], pb[T::NA4], pc[T::NA4];  // ROW: (n,i,j)  COL: (tap, channel, -)
#pragma unroll
  for (int u = 0; u < T::NA4; ++u) {
    const int idx = tid + u * 256;
    abase[u] = -1;
    pa[u] = -1;
    pb[u] = pc[u] = 0;
    if (idx >= T::UA4) continue;
    if (AMODE == AM_PLAIN_K) {
      const int r = m0 + idx / (BKT / 4);
      if (r < d.M) abase[u] = r * (int)A.sr + (idx % (BKT / 4)) * 4;
    } else if (AMODE == AM_PLAIN_R) {
      const int r = m0 + (idx % (BM / 4)) * 4;
      if (r < d.M) abase[u] = (idx / (BM / 4)) * (int)A.sk + r;
    } else if (AMODE == AM_ROW) {
      const int r = m0 + idx / (BKT / 4);
      if (r < d.M) {
        Pos ps = decode_pos(r, A.OH, A.OW);
        pa[u] = ps.n;
        pb[u] = ps.i;
        pc[u] = ps.j;
      }
    } else {  // AM_COL
      const int r = m0 + (idx % (BM / 4)) * 4;
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
    if (idx >= T::UB4) continue;
    if (BMODE == BM_K) {
      const int n = n0 + idx / (BKT / 4);
      if (n < d.N) bbase[u] = n * (int)B.sn + (idx % (BKT / 4)) * 4;
    } else {
      const int n = n0 + (idx % (BN / 4)) * 4;
      if (n < d.N) bbase[u] = (idx / (BN / 4)) * (int)B.sk + n;
    }
  }
  const int ow_sh = ((A.OW & (A.OW - 1)) == 0) ? __builtin_ctz(A.OW > 0 ? A.OW : 1) : -1;
  const int oh_sh = ((A.OH & (A.OH - 1)) == 0) ? __builtin_ctz(A.OH > 0 ? A.OH : 1) : -1;

  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 ra[T::NA4], ya[AACT ? T::NA4 : 1], rb[T::NB4];

  // ------------------------------------------------------------------------------------------------
  // global -> registers (no dependent use: the loads stay in flight across the MFMAs)
  // ------------------------------------------------------------------------------------------------
  auto load_tiles = [&](int k0) {
#if defined(MVK_PHASES) || defined(MVK_EXPER)
    if (d.dbg_flags & 1) {  // experiment: no A traffic
      for (int u = 0; u < T::NA4; ++u) ra[u] = u32x4{0, 0, 0, 0};
      goto load_b_only;
    }
#endif
    if (AMODE == AM_PLAIN_K) {
#pragma unroll
      for (int u = 0; u < T::NA4; ++u) {
        const int idx = tid + u * 256;
        const bool ok = abase[u] >= 0 && (k0 + (idx % (BKT / 4)) * 4) < kend;
        const int off = ok ? (abase[u] + k0) * 4 : OOB;
        ra[u] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
        if (AACT) ya[u] = __builtin_amdgcn_raw_buffer_load_b128(rsY, off, 0, 0);
      }
    } else if (AMODE == AM_PLAIN_R) {
#pragma unroll
      for (int u = 0; u < T::NA4; ++u) {
        const int idx = tid + u * 256;
        const bool ok = abase[u] >= 0 && (k0 + idx / (BM / 4)) < kend;
        const int off = ok ? (abase[u] + k0 * (int)A.sk) * 4 : OOB;
        ra[u] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
        if (AACT) ya[u] = __builtin_amdgcn_raw_buffer_load_b128(rsY, off, 0, 0);
      }
    } else if (AMODE == AM_ROW) {
      const int tap = k0 / A.C;  // block-uniform: C % BKT == 0
      const int c0 = k0 - tap * A.C;
      int dh, dw;
      if (A.kind == A_UP) {
        dh = ph - (tap >> 1);
        dw = pw - (tap & 1);
      } else {
        int kh, kw;
        a_tap(A, tap, kh, kw);
        dh = kh - 1;
        dw = kw - 1;
      }
      const int mul = (A.kind == A_UP) ? 1 : a_mul(A);
#pragma unroll
      for (int u = 0; u < T::NA4; ++u) {
        const int idx = tid + u * 256;
        const int kq = (idx % (BKT / 4)) * 4;
        const int hh = mul * pb[u] + dh, ww = mul * pc[u] + dw;
        const bool ok = pa[u] >= 0 && (k0 + kq) < kend && hh >= 0 && hh < A.H && ww >= 0 && ww < A.W;
#if defined(MVK_PHASES) || defined(MVK_EXPER)
        const int img = (d.dbg_flags & 4) ? (pa[u] & 7) : pa[u];  // experiment: all blocks read 8 images (cache hits)
#else
        const int img = pa[u];
#endif
        const int off = ok ? (((img * A.H + hh) * A.W + ww) * A.C + c0 + kq) * 4 : OOB;
        ra[u] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
      }
    } else {  // AM_COL: rows = (tap, channel), k = position
#pragma unroll
      for (int u = 0; u < T::NA4; ++u) {
        const int idx = tid + u * 256;
        const int pos = k0 + idx / (BM / 4);
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
        int kh, kw;
        a_tap(A, pa[u], kh, kw);
        const int hh = a_mul(A) * pi - 1 + kh, ww = a_mul(A) * pj - 1 + kw;
        const bool ok = pa[u] >= 0 && pos < kend && hh >= 0 && hh < A.H && ww >= 0 && ww < A.W;
        const int off = ok ? (((pn * A.H + hh) * A.W + ww) * A.C + pb[u]) * 4 : OOB;
        ra[u] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
      }
    }
#if defined(MVK_PHASES) || defined(MVK_EXPER)
  load_b_only:
    if (d.dbg_flags & 2) {  // experiment: no B traffic
      for (int u = 0; u < T::NB4; ++u) rb[u] = u32x4{0, 0, 0, 0};
      return;
    }
#endif
    if (BMODE == BM_K) {
#pragma unroll
      for (int u = 0; u < T::NB4; ++u) {
        const int idx = tid + u * 256;
        const bool ok = bbase[u] >= 0 && (k0 + (idx % (BKT / 4)) * 4) < kend;
        rb[u] = __builtin_amdgcn_raw_buffer_load_b128(rsB, ok ? (bbase[u] + k0) * 4 : OOB, 0, 0);
      }
    } else {
#pragma unroll
      for (int u = 0; u < T::NB4; ++u) {
        const int idx = tid + u * 256;
        const bool ok = bbase[u] >= 0 && (k0 + idx / (BN / 4)) < kend;
        rb[u] = __builtin_amdgcn_raw_buffer_load_b128(rsB, ok ? (bbase[u] + k0 * (int)B.sk) * 4 : OOB, 0, 0);
      }
    }
  };

  // ------------------------------------------------------------------------------------------------
  // registers -> LDS (k-major tiles)
  // ------------------------------------------------------------------------------------------------
  auto store_tiles = [&](int buf) {
#if defined(MVK_PHASES) || defined(MVK_EXPER)
    if (d.dbg_flags & 16) return;  // experiment: no LDS writes
#endif
    float* da = As + buf * BKT * T::SA;
    float* db = Bs + buf * BKT * T::SB;
#pragma unroll
    for (int u = 0; u < T::NA4; ++u) {
      const int idx = tid + u * 256;
      if (idx >= T::UA4) continue;
      float4 v = make_float4(__uint_as_float(ra[u].x), __uint_as_float(ra[u].y), __uint_as_float(ra[u].z),
                             __uint_as_float(ra[u].w));
      if (AACT) {
        v.x *= mvk_act_grad_from_out(__uint_as_float(ya[u].x), A.act);
        v.y *= mvk_act_grad_from_out(__uint_as_float(ya[u].y), A.act);
        v.z *= mvk_act_grad_from_out(__uint_as_float(ya[u].z), A.act);
        v.w *= mvk_act_grad_from_out(__uint_as_float(ya[u].w), A.act);
      }
      if (AMODE == AM_PLAIN_K || AMODE == AM_ROW) {
        const int row = idx / (BKT / 4), kq = (idx % (BKT / 4)) * 4;
        da[(kq + 0) * T::SA + row] = v.x;
        da[(kq + 1) * T::SA + row] = v.y;
        da[(kq + 2) * T::SA + row] = v.z;
        da[(kq + 3) * T::SA + row] = v.w;
      } else {
        const int kk = idx / (BM / 4), row = (idx % (BM / 4)) * 4;
        *reinterpret_cast<float4*>(da + kk * T::SA + row) = v;
      }
    }
#pragma unroll
    for (int u = 0; u < T::NB4; ++u) {
      const int idx = tid + u * 256;
      if (idx >= T::UB4) continue;
      const float4 v = make_float4(__uint_as_float(rb[u].x), __uint_as_float(rb[u].y), __uint_as_float(rb[u].z),
                                   __uint_as_float(rb[u].w));
      if (BMODE == BM_K) {
        const int col = idx / (BKT / 4), kq = (idx % (BKT / 4)) * 4;
        db[(kq + 0) * T::SB + col] = v.x;
        db[(kq + 1) * T::SB + col] = v.y;
        db[(kq + 2) * T::SB + col] = v.z;
        db[(kq + 3) * T::SB + col] = v.w;
      } else {
        const int kk = idx / (BN / 4), col = (idx % (BN / 4)) * 4;
        *reinterpret_cast<float4*>(db + kk * T::SB + col) = v;
      }
    }
  };

  auto compute = [&](int buf) {
    const float* a_s = As + buf * BKT * T::SA + wm * T::WTM + l31;
    const float* b_s = Bs + buf * BKT * T::SB + wn * T::WTN + l31;
#pragma unroll
    for (int ks = 0; ks < BKT / 2; ++ks) {
      float av[T::TM], bv[T::TN];
#pragma unroll
      for (int a = 0; a < T::TM; ++a) av[a] = a_s[(2 * ks + lhi) * T::SA + a * 32];
#pragma unroll
      for (int b = 0; b < T::TN; ++b) bv[b] = b_s[(2 * ks + lhi) * T::SB + b * 32];
#pragma unroll
      for (int a = 0; a < T::TM; ++a)
#pragma unroll
        for (int b = 0; b < T::TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
  };

#ifdef MVK_PHASES
  unsigned long long ph_t[5] = {0, 0, 0, 0, 0};
  const unsigned long long tk0 = __builtin_readcyclecounter();
#define MVK_TICK(i, prev) { unsigned long long now__ = __builtin_readcyclecounter(); ph_t[i] += now__ - prev; prev = now__; }
#else
#define MVK_TICK(i, prev)
#endif
  load_tiles(kbeg);
  store_tiles(0);
  __syncthreads();
  int buf = 0;
#ifdef MVK_PHASES
  unsigned long long tk = __builtin_readcyclecounter();
  ph_t[4] = tk - tk0;
#endif
  for (int t = 0; t < ntiles; ++t) {
    const bool more = (t + 1) < ntiles;
    if (more) load_tiles(kbeg + (t + 1) * BKT);
    MVK_TICK(0, tk)
    compute(buf);
    MVK_TICK(1, tk)
#ifdef MVK_PHASES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MVK_TICK(4, tk)   // pure global-load wait (ph_t[4] also holds the prologue time; subtract it when reading)
#endif
    if (more) store_tiles(buf ^ 1);
    MVK_TICK(2, tk)
    __syncthreads();
    MVK_TICK(3, tk)
    buf ^= 1;
  }
#ifdef MVK_PHASES
  if (d.dbg && lane == 0) {
    for (int i = 0; i < 5; ++i) atomicAdd(d.dbg + i, ph_t[i]);
    atomicAdd(d.dbg + 5, (unsigned long long)ntiles);
    unsigned long long te = __builtin_readcyclecounter();
    atomicAdd(d.dbg + 6, te - tk0);
    atomicAdd(d.dbg + 7, 1ull);
  }
#endif
#if defined(MVK_PHASES) || defined(MVK_EXPER)
  if (d.dbg_flags & 8) {  // experiment: no epilogue (keep the accumulators alive)
    float s = 0.f;
    for (int a = 0; a < T::TM; ++a) for (int b = 0; b < T::TN; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    if (s == 12345.678f) d.e.out[0] = s;
    return;
  }
#endif
  if (run_epilogue_vec<T, BM, BN, 2 * BKT * (T::SA + T::SB)>(d, acc, lds, tid, m0, n0, wm, wn, l31, lhi, ph, pw)) return;
  run_epilogue<T>(d, acc, lds, tid, m0, n0, wm, wn, l31, lhi, ph, pw);
}

}  // namespace mvk
