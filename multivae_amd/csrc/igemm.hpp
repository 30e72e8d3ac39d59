// fp32 MFMA implicit-GEMM engine for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains).
//
// One kernel template computes C = op(A) * op(B) for every dense layer on the path:
//   - A may be a plain strided matrix or an *implicit* im2col view of an NHWC/NCHW activation tensor
//     (4x4 / stride-2 / pad-1 convolution "down" gather, its transposed "up" gather, or either one with the
//     roles of the GEMM row and reduction index swapped for the weight-gradient GEMM);
//   - B is a plain strided matrix (weights are pre-packed so that the GEMM N index is contiguous);
//   - the epilogue fuses bias, activation, multiplication by a saved activation's derivative, the
//     parity scatter of the transposed convolution, the scatter back into the reference weight layout and
//     atomic split-K accumulation.
// LDS tiles are k-major (As[kk][i], Bs[kk][j]) so both MFMA operands are bank-conflict-free ds_read_b32
// (lanes 0-31 read 32 consecutive floats of one k row, lanes 32-63 the next k row).
#pragma once
#include "common.hpp"

namespace mvk {

enum AKind { A_PLAIN = 0, A_DOWN = 1, A_DOWN_NCHW = 2, A_UP = 3 };
// A_DOWN also covers every "same grid or strided, zero padded by 1" window through (tw, mul): 4x4/stride 2 (tw = 4,
// mul = 2) and 3x3/stride 1 (tw = 3, mul = 1): input row = mul * i - 1 + kh, kh = tap / tw, kw = tap % tw.
enum EpiKind { E_ROWMAJOR = 0, E_UP = 1, E_CONVREF = 2, E_UNFLATREF = 3, E_UP_NCHW = 4 };
enum ZMode { Z_NONE = 0, Z_SPLITK = 1, Z_PARITY = 2 };

struct AOperand {
  const float* p;
  int kind;
  int trans;        // gather kinds: 0 -> GEMM row = position, k = (tap,channel); 1 -> swapped (wgrad)
  long long sr, sk;  // A_PLAIN: element (r,k) at p[r*sr + k*sk]
  int contig_k;     // 1: memory contiguous along k; 0: contiguous along r
  int vec4;         // 16-byte vector loads allowed along the contiguous direction
  int C, H, W;      // gathered tensor: channels and spatial size
  int OH, OW;       // spatial size enumerating positions
  const float* act_src;  // optional: loaded value *= act'(act_src[same offset])
  int act;
  int tw, mul;           // A_DOWN / A_DOWN_NCHW window: taps per window row and stride (0 = the 4x4/stride-2 default)
  int bf3;               // operand is stored pre-split: 3 bf16 planes [piece][element], p -> plane 0
  long long plane_bytes; // bytes between planes
};

struct BOperand {
  const float* p;
  long long sk, sn;  // element (k,n) at p[k*sk + n*sn]
  int contig_k;
  int vec4;
  long long z_stride;  // Z_PARITY: per-parity offset
};

struct Epilogue {
  float* out;
  int kind;
  long long ld;
  const float* bias;
  int bias_mod;
  int act;
  const float* act_src;  // multiply by act'(act_src[out offset]) before storing
  int src_act;
  int atomic;
  int Cu, OH, OW;  // E_UP / E_CONVREF / E_UNFLATREF geometry
  int taps;        // E_CONVREF: window size of the reference weight layout (0 = 16)
  float* ws;       // Z_SPLITK: if set, raw partial tiles go to ws[z][m][n] and splitk_reduce_kernel finishes
  float* colsum_part;  // optional: per-workgroup column sums of the stored values -> colsum_part[(z*gridDim.x + bx)*N + n]
                       // (vectorised epilogue only; the host finishes them with an ordered reduce: bias gradients)
  const float* res;    // optional residual: out = res[out offset] + res_alpha * (value after activation / mask)
  float res_alpha;     // (the column sums are those of the value BEFORE the residual)
};

struct GemmDesc {
  AOperand a;
  BOperand b;
  Epilogue e;
  int M, N, K;
  int zmode;
  int ksplit_tiles;  // Z_SPLITK: k-tiles per z slice
  unsigned long long* dbg;  // phase-cycle counters (debug builds with -DMVK_PHASES only)
  int dbg_flags;            // experiment switches (debug builds only)
};

#ifndef MVK_MIN_WAVES
#define MVK_MIN_WAVES 3
#endif
constexpr int BK = 16;

template <int BM, int BN>
struct TileCfg {
  static constexpr int WAVES_N = (BN >= 64) ? 2 : 1;
  static constexpr int WAVES_M = 4 / WAVES_N;
  static constexpr int WTM = BM / WAVES_M;  // wave tile
  static constexpr int WTN = BN / WAVES_N;
  static constexpr int TM = WTM / 32;
  static constexpr int TN = WTN / 32;
  static constexpr int SA = BM + 4;  // LDS row strides (floats), multiple of 4 for b128 writes
  static constexpr int SB = BN + 4;
  // staging: scalar path = R*BK/256 floats per thread; vector path = ceil(R*BK/4/256) float4 per thread
  static constexpr int NA1 = BM * BK / 256;
  static constexpr int NB1 = BN * BK / 256;
  static constexpr int NA4 = (BM * BK / 4 + 255) / 256;
  static constexpr int NB4 = (BN * BK / 4 + 255) / 256;
  static constexpr int NA = (4 * NA4 > NA1) ? 4 * NA4 : NA1;
  static constexpr int NB = (4 * NB4 > NB1) ? 4 * NB4 : NB1;
  static constexpr int UA4 = BM * BK / 4;  // number of float4 units in a tile
  static constexpr int UB4 = BN * BK / 4;
  static_assert(TM >= 1 && TN >= 1, "tile too small");
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- position decode helpers -------------------------------------------------------------------------
struct Pos {
  int n, i, j;
};
__device__ __forceinline__ Pos decode_pos(int pos, int OH, int OW) {
  Pos p;
  p.j = pos % OW;
  int t = pos / OW;
  p.i = t % OH;
  p.n = t / OH;
  return p;
}

// window geometry of the A_DOWN kinds
__device__ __forceinline__ int a_mul(const AOperand& a) { return a.mul > 0 ? a.mul : 2; }
__device__ __forceinline__ void a_tap(const AOperand& a, int tap, int& kh, int& kw) {
  if (a.tw == 3) {
    kh = tap / 3;
    kw = tap - 3 * kh;
  } else {
    kh = tap >> 2;
    kw = tap & 3;
  }
}

// offset (in floats) of gathered element for position `ps`, reduction index kk=(tap,c); returns validity
__device__ __forceinline__ bool gather_off(const AOperand& a, const Pos& ps, int kk, int ph, int pw,
                                           long long& off) {
  int tap = kk / a.C;
  int c = kk - tap * a.C;
  if (a.kind == A_UP) {
    int ta = tap >> 1, tb = tap & 1;
    int oh = ps.i + ph - ta;
    int ow = ps.j + pw - tb;
    off = (((long long)ps.n * a.H + oh) * a.W + ow) * a.C + c;
    return (oh >= 0) & (oh < a.H) & (ow >= 0) & (ow < a.W);
  }
  int kh, kw;
  a_tap(a, tap, kh, kw);
  int h = a_mul(a) * ps.i - 1 + kh;
  int w = a_mul(a) * ps.j - 1 + kw;
  bool ok = (h >= 0) & (h < a.H) & (w >= 0) & (w < a.W);
  if (a.kind == A_DOWN)
    off = (((long long)ps.n * a.H + h) * a.W + w) * a.C + c;
  else
    off = (((long long)ps.n * a.C + c) * a.H + h) * a.W + w;
  return ok;
}

// Conditions of the float4 LDS-staged epilogue (host and device agree on them: the host decides on fusing the
// bias-gradient column sums into the launch).
__host__ __device__ inline bool epilogue_vec_ok(const Epilogue& E, int N) {
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
  return !E.ws && !E.atomic && (E.kind == E_ROWMAJOR || E.kind == E_UP) && (N % 4 == 0) && al16(E.out) &&
         (!E.act_src || al16(E.act_src)) && (!E.res || al16(E.res)) && (E.kind == E_UP ? (E.Cu % 4 == 0) : (E.ld % 4 == 0)) &&
         (!E.bias || (E.bias_mod % 4 == 0 && al16(E.bias)));
}

// ---- shared epilogue -------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ void run_epilogue(const GemmDesc& d, f32x16 (&acc)[T::TM][T::TN], float* lds, int tid, int m0,
                                             int n0, int wm, int wn, int l31, int lhi, int ph, int pw, int bz = -1) {
  // ---------------- epilogue ----------------
  // The accumulators of one 32x32 MFMA tile are parked in LDS (one private column per thread, so no
  // barrier is needed) and consumed by a rolled loop: the address / activation code exists once instead of
  // 16*TM*TN times.  The tile buffers are free: the main loop ended with a barrier.
  const Epilogue& E = d.e;
  if (E.ws) {
    // split-K partial: plain coalesced stores of the raw accumulators into this slice's slab
    // (bz: the split-K slice when the caller's workgroup position is not its blockIdx — igemm_bf_pair_kernel)
    float* slab = E.ws + (long long)(bz >= 0 ? (unsigned)bz : blockIdx.z) * d.M * d.N;
#pragma unroll
    for (int a = 0; a < T::TM; ++a) {
#pragma unroll
      for (int b = 0; b < T::TN; ++b) {
        const int n = n0 + wn * T::WTN + b * 32 + l31;
        const int mbase = m0 + wm * T::WTM + a * 32 + 4 * lhi;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mbase + (r & 3) + 8 * (r >> 2);
          if (m < d.M && n < d.N) slab[(long long)m * d.N + n] = acc[a][b][r];
        }
      }
    }
    return;
  }
  float* park = lds + tid;  // element r at park[r * 256]
#pragma unroll
  for (int a = 0; a < T::TM; ++a) {
#pragma unroll
    for (int b = 0; b < T::TN; ++b) {
#pragma unroll
      for (int r = 0; r < 16; ++r) park[r * 256] = acc[a][b][r];
      const int n = n0 + wn * T::WTN + b * 32 + l31;
      const int mbase = m0 + wm * T::WTM + a * 32 + 4 * lhi;
      if (n < d.N) {
        const float bias_v = E.bias ? E.bias[n % E.bias_mod] : 0.f;
#pragma unroll 1
        for (int r = 0; r < 16; ++r) {
          const int m = mbase + (r & 3) + 8 * (r >> 2);
          if (m >= d.M) continue;
          float v = park[r * 256];
          long long off;
          if (E.kind == E_ROWMAJOR) {
            off = (long long)m * E.ld + n;
          } else if (E.kind == E_UP) {
            Pos ps = decode_pos(m, E.OH, E.OW);
            off = (((long long)ps.n * (2 * E.OH) + 2 * ps.i + ph) * (2 * E.OW) + 2 * ps.j + pw) * E.Cu + n;
          } else if (E.kind == E_UP_NCHW) {
            Pos ps = decode_pos(m, E.OH, E.OW);
            off = (((long long)ps.n * E.Cu + n) * (2 * E.OH) + 2 * ps.i + ph) * (2 * E.OW) + 2 * ps.j + pw;
          } else if (E.kind == E_CONVREF) {
            int tap = m / E.Cu;
            int cu = m - tap * E.Cu;
            off = ((long long)n * E.Cu + cu) * (E.taps > 0 ? E.taps : 16) + tap;
          } else {  // E_UNFLATREF: m = ci, n = (tap, co)
            int tap = n / E.Cu;
            int co = n - tap * E.Cu;
            off = ((long long)m * E.Cu + co) * 16 + tap;
          }
          v = mvk_act(v + bias_v, E.act);
          if (E.act_src) v *= mvk_act_grad_from_out(E.act_src[off], E.src_act);
          if (E.res) v = fmaf(E.res_alpha, v, E.res[off]);
          if (E.atomic)
            atomicAdd(E.out + off, v);
          else
            E.out[off] = v;
        }
      }
    }
  }
}

// ---- vectorised epilogue ------------------------------------------------------------------------------------------
// For row-major / NHWC-parity outputs with N % 4 == 0: the accumulators of `RP` tile rows at a time are parked in
// LDS as a [RP][BN+4] slab, then every thread emits whole float4 row segments: one address computation, one
// bias / activation-mask float4 load and one 16-byte store per 4 outputs, all mask loads independent (the scalar
// epilogue above does a dependent 4-byte load + address decode per element).
template <class T, int BM, int BN, int LDS_FLOATS>
__device__ __forceinline__ bool run_epilogue_vec(const GemmDesc& d, f32x16 (&acc)[T::TM][T::TN], float* lds, int tid,
                                                 int m0, int n0, int wm, int wn, int l31, int lhi, int ph, int pw) {
  const Epilogue& E = d.e;
  constexpr int CS = BN + 4;
  // rows per pass: a multiple of 32 (one MFMA tile row block per wave row) that fits the LDS budget
  constexpr int RP_MAX = LDS_FLOATS / CS;
  constexpr int RP = (RP_MAX >= BM) ? BM : ((RP_MAX >= T::WTM && T::WAVES_M > 1) ? T::WTM : (RP_MAX >= 32 ? 32 : 0));
  if (RP == 0) return false;
  const bool vec_ok = epilogue_vec_ok(E, d.N);
  if (!vec_ok) return false;
  constexpr int NPASS = BM / RP;
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);  // this thread's column quad (fixed: 256 % (BN/4) == 0)
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    const int r_lo = p * RP;
    // park: each lane writes its accumulator elements that fall into rows [r_lo, r_lo + RP)
#pragma unroll
    for (int a = 0; a < T::TM; ++a) {
      const int rbase = wm * T::WTM + a * 32;  // tile-local row of this 32-row block
      if (rbase >= r_lo && rbase < r_lo + RP) {
#pragma unroll
        for (int b = 0; b < T::TN; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            lds[(rbase - r_lo + (r & 3) + 8 * (r >> 2) + 4 * lhi) * CS + wn * T::WTN + b * 32 + l31] = acc[a][b][r];
      }
    }
    __syncthreads();
    constexpr int UNITS = RP * BN / 4;
#pragma unroll
    for (int u = 0; u < (UNITS + 255) / 256; ++u) {
      const int idx = tid + u * 256;
      if (idx < UNITS) {
        const int row = idx / (BN / 4);
        const int c4 = (idx - row * (BN / 4)) * 4;
        const int m = m0 + r_lo + row, n = n0 + c4;
        if (m < d.M && n < d.N) {
          long long off;
          if (E.kind == E_ROWMAJOR) {
            off = (long long)m * E.ld + n;
          } else {
            Pos ps = decode_pos(m, E.OH, E.OW);
            off = (((long long)ps.n * (2 * E.OH) + 2 * ps.i + ph) * (2 * E.OW) + 2 * ps.j + pw) * E.Cu + n;
          }
          float4 v = *reinterpret_cast<const float4*>(lds + row * CS + c4);
          if (E.bias) {
            const float4 bb = *reinterpret_cast<const float4*>(E.bias + (n % E.bias_mod));
            v.x += bb.x;
            v.y += bb.y;
            v.z += bb.z;
            v.w += bb.w;
          }
          v.x = mvk_act(v.x, E.act);
          v.y = mvk_act(v.y, E.act);
          v.z = mvk_act(v.z, E.act);
          v.w = mvk_act(v.w, E.act);
          if (E.act_src) {
            const float4 y = *reinterpret_cast<const float4*>(E.act_src + off);
            v.x *= mvk_act_grad_from_out(y.x, E.src_act);
            v.y *= mvk_act_grad_from_out(y.y, E.src_act);
            v.z *= mvk_act_grad_from_out(y.z, E.src_act);
            v.w *= mvk_act_grad_from_out(y.w, E.src_act);
          }
          csum.x += v.x;
          csum.y += v.y;
          csum.z += v.z;
          csum.w += v.w;
          if (E.res) {
            const float4 rr = *reinterpret_cast<const float4*>(E.res + off);
            v.x = fmaf(E.res_alpha, v.x, rr.x);
            v.y = fmaf(E.res_alpha, v.y, rr.y);
            v.z = fmaf(E.res_alpha, v.z, rr.z);
            v.w = fmaf(E.res_alpha, v.w, rr.w);
          }
          *reinterpret_cast<float4*>(E.out + off) = v;
        }
      }
    }
    if (p + 1 < NPASS) __syncthreads();
  }
  if (E.colsum_part) {
    // fixed-order block reduction of the per-thread column quads: 256/(BN/4) row groups -> one value per column
    constexpr int QN = BN / 4, RG = 256 / QN;
    static_assert(256 % QN == 0 && RG * BN <= LDS_FLOATS, "column-sum scratch");
    __syncthreads();
    *reinterpret_cast<float4*>(lds + (tid / QN) * BN + (tid % QN) * 4) = csum;
    __syncthreads();
    if (tid < BN && n0 + tid < d.N) {
      float t = 0.f;
#pragma unroll 4
      for (int g = 0; g < RG; ++g) t += lds[g * BN + tid];
      E.colsum_part[((long long)blockIdx.z * gridDim.x + blockIdx.x) * d.N + n0 + tid] = t;
    }
  }
  return true;
}

template <int BM, int BN>
__global__ __launch_bounds__(256, MVK_MIN_WAVES) void igemm_kernel(const GemmDesc d) {
  using T = TileCfg<BM, BN>;
  __shared__ __attribute__((aligned(16))) float lds[2 * BK * (T::SA + T::SB)];
  float* As = lds;                    // [2][BK][SA]
  float* Bs = lds + 2 * BK * T::SA;   // [2][BK][SB]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / T::WAVES_N;
  const int wn = wave % T::WAVES_N;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  int ph = 0, pw = 0;
  int kt_begin = 0, kt_end = (d.K + BK - 1) / BK;
  const float* bp = d.b.p;
  if (d.zmode == Z_PARITY) {
    ph = blockIdx.z >> 1;
    pw = blockIdx.z & 1;
    bp += (long long)blockIdx.z * d.b.z_stride;
  } else if (d.zmode == Z_SPLITK) {
    kt_begin = blockIdx.z * d.ksplit_tiles;
    int e = kt_begin + d.ksplit_tiles;
    kt_end = e < kt_end ? e : kt_end;
    if (kt_begin >= kt_end) return;
  }

  f32x16 acc[T::TM][T::TN];
#pragma unroll
  for (int a = 0; a < T::TM; ++a)
#pragma unroll
    for (int b = 0; b < T::TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float sa[T::NA];
  float sb[T::NB];

  const AOperand& A = d.a;
  const BOperand& B = d.b;
  const bool a_gather = A.kind != A_PLAIN;

  // ---------------- hoisted gather decode (fast paths) ----------------
  // FAST_ROW: rows are positions, k = (tap, channel), NHWC, 16-byte loads along the channel, C % BK == 0:
  //           the (n,i,j) of each staged row is decoded once; inside the K loop the tap is block-uniform.
  // FAST_COL: weight-gradient view (rows = (tap, channel), k = position): tap/channel decoded once; the position
  //           decode per tile uses shifts when OH, OW are powers of two.
  const bool fast_row = a_gather && !A.trans && A.vec4 && A.contig_k && (A.C % BK == 0) && A.kind != A_DOWN_NCHW;
  const bool fast_col = a_gather && A.trans && A.vec4 && !A.contig_k && A.kind == A_DOWN;
  int pre_a[T::NA4], pre_b[T::NA4], pre_c[T::NA4];
  if (fast_row) {
#pragma unroll
    for (int u = 0; u < T::NA4; ++u) {
      const int idx = tid + u * 256;
      const int r = m0 + idx / (BK / 4);
      pre_a[u] = -1;
      pre_b[u] = pre_c[u] = 0;
      if (idx < T::UA4 && r < d.M) {
        Pos ps = decode_pos(r, A.OH, A.OW);
        pre_a[u] = ps.n;
        pre_b[u] = ps.i;
        pre_c[u] = ps.j;
      }
    }
  } else if (fast_col) {
#pragma unroll
    for (int u = 0; u < T::NA4; ++u) {
      const int idx = tid + u * 256;
      const int r = m0 + (idx % (BM / 4)) * 4;
      pre_a[u] = -1;
      pre_b[u] = pre_c[u] = 0;
      if (idx < T::UA4 && r < d.M) {
        const int tap = r / A.C;
        pre_a[u] = tap;           // kh*4 + kw
        pre_b[u] = r - tap * A.C; // channel
      }
    }
  }
  const int ow_sh = ((A.OW & (A.OW - 1)) == 0) ? __builtin_ctz(A.OW > 0 ? A.OW : 1) : -1;
  const int oh_sh = ((A.OH & (A.OH - 1)) == 0) ? __builtin_ctz(A.OH > 0 ? A.OH : 1) : -1;

  // ---------------- global -> registers ----------------
  auto load_a = [&](int kt) {
    const int k0 = kt * BK;
    if (fast_row) {
      const int tap = k0 / A.C;
      const int c0 = k0 - tap * A.C;
#pragma unroll
      for (int u = 0; u < T::NA4; ++u) {
        const int idx = tid + u * 256;
        const int c = c0 + (idx % (BK / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pre_a[u] >= 0 && (k0 + (idx % (BK / 4)) * 4) < d.K) {
          int hh, ww;
          if (A.kind == A_UP) {
            hh = pre_b[u] + ph - (tap >> 1);
            ww = pre_c[u] + pw - (tap & 1);
          } else {
            int kh, kw;
            a_tap(A, tap, kh, kw);
            hh = a_mul(A) * pre_b[u] - 1 + kh;
            ww = a_mul(A) * pre_c[u] - 1 + kw;
          }
          if (hh >= 0 && hh < A.H && ww >= 0 && ww < A.W) {
            const long long off = (((long long)pre_a[u] * A.H + hh) * A.W + ww) * A.C + c;
            v = *reinterpret_cast<const float4*>(A.p + off);
            if (A.act_src) {
              float4 y = *reinterpret_cast<const float4*>(A.act_src + off);
              v.x *= mvk_act_grad_from_out(y.x, A.act);
              v.y *= mvk_act_grad_from_out(y.y, A.act);
              v.z *= mvk_act_grad_from_out(y.z, A.act);
              v.w *= mvk_act_grad_from_out(y.w, A.act);
            }
          }
        }
        sa[4 * u + 0] = v.x;
        sa[4 * u + 1] = v.y;
        sa[4 * u + 2] = v.z;
        sa[4 * u + 3] = v.w;
      }
      return;
    }
    if (fast_col) {
#pragma unroll
      for (int u = 0; u < T::NA4; ++u) {
        const int idx = tid + u * 256;
        const int pos = k0 + idx / (BM / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pre_a[u] >= 0 && pos < d.K) {
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
          a_tap(A, pre_a[u], kh, kw);
          const int hh = a_mul(A) * pi - 1 + kh;
          const int ww = a_mul(A) * pj - 1 + kw;
          if (hh >= 0 && hh < A.H && ww >= 0 && ww < A.W) {
            const long long off = (((long long)pn * A.H + hh) * A.W + ww) * A.C + pre_b[u];
            v = *reinterpret_cast<const float4*>(A.p + off);
            if (A.act_src) {
              float4 y = *reinterpret_cast<const float4*>(A.act_src + off);
              v.x *= mvk_act_grad_from_out(y.x, A.act);
              v.y *= mvk_act_grad_from_out(y.y, A.act);
              v.z *= mvk_act_grad_from_out(y.z, A.act);
              v.w *= mvk_act_grad_from_out(y.w, A.act);
            }
          }
        }
        sa[4 * u + 0] = v.x;
        sa[4 * u + 1] = v.y;
        sa[4 * u + 2] = v.z;
        sa[4 * u + 3] = v.w;
      }
      return;
    }
    if (A.contig_k) {
      // units: (row, kvec) ; vec along k
      if (A.vec4) {
#pragma unroll
        for (int u = 0; u < T::NA4; ++u) {
          int idx = tid + u * 256;
          int row = idx / (BK / 4);
          int k = k0 + (idx % (BK / 4)) * 4;
          int r = m0 + row;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (idx < T::UA4 && r < d.M && k < d.K) {
            long long off;
            bool ok = true;
            if (!a_gather) {
              off = r * A.sr + k * A.sk;
            } else {
              Pos ps = decode_pos(A.trans ? k : r, A.OH, A.OW);
              ok = gather_off(A, ps, A.trans ? r : k, ph, pw, off);
            }
            if (ok) {
              v = *reinterpret_cast<const float4*>(A.p + off);
              if (A.act_src) {
                float4 y = *reinterpret_cast<const float4*>(A.act_src + off);
                v.x *= mvk_act_grad_from_out(y.x, A.act);
                v.y *= mvk_act_grad_from_out(y.y, A.act);
                v.z *= mvk_act_grad_from_out(y.z, A.act);
                v.w *= mvk_act_grad_from_out(y.w, A.act);
              }
            }
          }
          sa[4 * u + 0] = v.x;
          sa[4 * u + 1] = v.y;
          sa[4 * u + 2] = v.z;
          sa[4 * u + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int u = 0; u < T::NA1; ++u) {
          int idx = tid + u * 256;
          int row = idx / BK;
          int k = k0 + (idx % BK);
          int r = m0 + row;
          float v = 0.f;
          if (r < d.M && k < d.K) {
            long long off;
            bool ok = true;
            if (!a_gather) {
              off = r * A.sr + k * A.sk;
            } else {
              Pos ps = decode_pos(A.trans ? k : r, A.OH, A.OW);
              ok = gather_off(A, ps, A.trans ? r : k, ph, pw, off);
            }
            if (ok) {
              v = A.p[off];
              if (A.act_src) v *= mvk_act_grad_from_out(A.act_src[off], A.act);
            }
          }
          sa[u] = v;
        }
      }
    } else {
      // contiguous along r: units (k, rvec)
      if (A.vec4) {
#pragma unroll
        for (int u = 0; u < T::NA4; ++u) {
          int idx = tid + u * 256;
          int kk = idx / (BM / 4);
          int row = (idx % (BM / 4)) * 4;
          int k = k0 + kk;
          int r = m0 + row;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (idx < T::UA4 && r < d.M && k < d.K) {
            long long off;
            bool ok = true;
            if (!a_gather) {
              off = r * A.sr + k * A.sk;
            } else {
              Pos ps = decode_pos(A.trans ? k : r, A.OH, A.OW);
              ok = gather_off(A, ps, A.trans ? r : k, ph, pw, off);
            }
            if (ok) {
              v = *reinterpret_cast<const float4*>(A.p + off);
              if (A.act_src) {
                float4 y = *reinterpret_cast<const float4*>(A.act_src + off);
                v.x *= mvk_act_grad_from_out(y.x, A.act);
                v.y *= mvk_act_grad_from_out(y.y, A.act);
                v.z *= mvk_act_grad_from_out(y.z, A.act);
                v.w *= mvk_act_grad_from_out(y.w, A.act);
              }
            }
          }
          sa[4 * u + 0] = v.x;
          sa[4 * u + 1] = v.y;
          sa[4 * u + 2] = v.z;
          sa[4 * u + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int u = 0; u < T::NA1; ++u) {
          int idx = tid + u * 256;
          int kk = idx / BM;
          int row = idx % BM;
          int k = k0 + kk;
          int r = m0 + row;
          float v = 0.f;
          if (r < d.M && k < d.K) {
            long long off;
            bool ok = true;
            if (!a_gather) {
              off = r * A.sr + k * A.sk;
            } else {
              Pos ps = decode_pos(A.trans ? k : r, A.OH, A.OW);
              ok = gather_off(A, ps, A.trans ? r : k, ph, pw, off);
            }
            if (ok) {
              v = A.p[off];
              if (A.act_src) v *= mvk_act_grad_from_out(A.act_src[off], A.act);
            }
          }
          sa[u] = v;
        }
      }
    }
  };

  auto load_b = [&](int kt) {
    const int k0 = kt * BK;
    if (B.contig_k) {
      if (B.vec4) {
#pragma unroll
        for (int u = 0; u < T::NB4; ++u) {
          int idx = tid + u * 256;
          int col = idx / (BK / 4);
          int k = k0 + (idx % (BK / 4)) * 4;
          int n = n0 + col;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (idx < T::UB4 && n < d.N && k < d.K) v = *reinterpret_cast<const float4*>(bp + k * B.sk + n * B.sn);
          sb[4 * u + 0] = v.x;
          sb[4 * u + 1] = v.y;
          sb[4 * u + 2] = v.z;
          sb[4 * u + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int u = 0; u < T::NB1; ++u) {
          int idx = tid + u * 256;
          int col = idx / BK;
          int k = k0 + (idx % BK);
          int n = n0 + col;
          sb[u] = (n < d.N && k < d.K) ? bp[k * B.sk + n * B.sn] : 0.f;
        }
      }
    } else {
      if (B.vec4) {
#pragma unroll
        for (int u = 0; u < T::NB4; ++u) {
          int idx = tid + u * 256;
          int kk = idx / (BN / 4);
          int col = (idx % (BN / 4)) * 4;
          int k = k0 + kk;
          int n = n0 + col;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (idx < T::UB4 && n < d.N && k < d.K) v = *reinterpret_cast<const float4*>(bp + k * B.sk + n * B.sn);
          sb[4 * u + 0] = v.x;
          sb[4 * u + 1] = v.y;
          sb[4 * u + 2] = v.z;
          sb[4 * u + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int u = 0; u < T::NB1; ++u) {
          int idx = tid + u * 256;
          int kk = idx / BN;
          int col = idx % BN;
          int k = k0 + kk;
          int n = n0 + col;
          sb[u] = (n < d.N && k < d.K) ? bp[k * B.sk + n * B.sn] : 0.f;
        }
      }
    }
  };

  // ---------------- registers -> LDS (k-major tiles) ----------------
  auto store_a = [&](int buf) {
    float* dst = As + buf * BK * T::SA;
    if (A.contig_k) {
      if (A.vec4) {
#pragma unroll
        for (int u = 0; u < T::NA4; ++u) {
          int idx = tid + u * 256;
          int row = idx / (BK / 4);
          int kq = (idx % (BK / 4)) * 4;
          if (idx < T::UA4) {
#pragma unroll
            for (int c = 0; c < 4; ++c) dst[(kq + c) * T::SA + row] = sa[4 * u + c];
          }
        }
      } else {
#pragma unroll
        for (int u = 0; u < T::NA1; ++u) {
          int idx = tid + u * 256;
          dst[(idx % BK) * T::SA + idx / BK] = sa[u];
        }
      }
    } else {
      if (A.vec4) {
#pragma unroll
        for (int u = 0; u < T::NA4; ++u) {
          int idx = tid + u * 256;
          int kk = idx / (BM / 4);
          int row = (idx % (BM / 4)) * 4;
          if (idx < T::UA4)
            *reinterpret_cast<float4*>(dst + kk * T::SA + row) =
                make_float4(sa[4 * u], sa[4 * u + 1], sa[4 * u + 2], sa[4 * u + 3]);
        }
      } else {
#pragma unroll
        for (int u = 0; u < T::NA1; ++u) {
          int idx = tid + u * 256;
          dst[(idx / BM) * T::SA + idx % BM] = sa[u];
        }
      }
    }
  };
  auto store_b = [&](int buf) {
    float* dst = Bs + buf * BK * T::SB;
    if (B.contig_k) {
      if (B.vec4) {
#pragma unroll
        for (int u = 0; u < T::NB4; ++u) {
          int idx = tid + u * 256;
          int col = idx / (BK / 4);
          int kq = (idx % (BK / 4)) * 4;
          if (idx < T::UB4) {
#pragma unroll
            for (int c = 0; c < 4; ++c) dst[(kq + c) * T::SB + col] = sb[4 * u + c];
          }
        }
      } else {
#pragma unroll
        for (int u = 0; u < T::NB1; ++u) {
          int idx = tid + u * 256;
          dst[(idx % BK) * T::SB + idx / BK] = sb[u];
        }
      }
    } else {
      if (B.vec4) {
#pragma unroll
        for (int u = 0; u < T::NB4; ++u) {
          int idx = tid + u * 256;
          int kk = idx / (BN / 4);
          int col = (idx % (BN / 4)) * 4;
          if (idx < T::UB4)
            *reinterpret_cast<float4*>(dst + kk * T::SB + col) =
                make_float4(sb[4 * u], sb[4 * u + 1], sb[4 * u + 2], sb[4 * u + 3]);
        }
      } else {
#pragma unroll
        for (int u = 0; u < T::NB1; ++u) {
          int idx = tid + u * 256;
          dst[(idx / BN) * T::SB + idx % BN] = sb[u];
        }
      }
    }
  };

  // ---------------- MFMA over one staged k-tile ----------------
  const int l31 = lane & 31;
  const int lhi = lane >> 5;
  auto compute = [&](int buf) {
    const float* a_s = As + buf * BK * T::SA + wm * T::WTM + l31;
    const float* b_s = Bs + buf * BK * T::SB + wn * T::WTN + l31;
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
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

  // ---------------- main loop: register-staged prefetch, double-buffered LDS ----------------
  load_a(kt_begin);
  load_b(kt_begin);
  store_a(0);
  store_b(0);
  __syncthreads();
  int buf = 0;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const bool more = (kt + 1) < kt_end;
    if (more) {
      load_a(kt + 1);
      load_b(kt + 1);
    }
    compute(buf);
    if (more) {
      store_a(buf ^ 1);
      store_b(buf ^ 1);
    }
    __syncthreads();
    buf ^= 1;
  }

  run_epilogue<T>(d, acc, lds, tid, m0, n0, wm, wn, l31, lhi, ph, pw);
}

// out[map(m,n)] (+)= epilogue(sum_z ws[z * zstride + m*N + n]) — deterministic split-K finish.
// A workgroup owns 32 consecutive (m,n) elements; its 8 z-lanes each sum every 8th slab (128-byte coalesced
// loads), then the 8 partials are combined in a fixed order through LDS.
__device__ __forceinline__ void splitk_store(const Epilogue& E, int N, long long idx, float v) {
  const int m = (int)(idx / N);
  const int n = (int)(idx - (long long)m * N);
  long long off;
  if (E.kind == E_ROWMAJOR) {
    off = (long long)m * E.ld + n;
  } else if (E.kind == E_CONVREF) {
    int tap = m / E.Cu;
    int cu = m - tap * E.Cu;
    off = ((long long)n * E.Cu + cu) * (E.taps > 0 ? E.taps : 16) + tap;
  } else {  // E_UNFLATREF
    int tap = n / E.Cu;
    int co = n - tap * E.Cu;
    off = ((long long)m * E.Cu + co) * 16 + tap;
  }
  if (E.bias) v += E.bias[n % E.bias_mod];
  v = mvk_act(v, E.act);
  if (E.act_src) v *= mvk_act_grad_from_out(E.act_src[off], E.src_act);
  if (E.res) v = fmaf(E.res_alpha, v, E.res[off]);
  if (E.atomic)
    E.out[off] += v;  // "accumulate" semantics; every element is owned by exactly one thread
  else
    E.out[off] = v;
}

__device__ __forceinline__ void splitk_reduce_body(const Epilogue& E, int M, int N, int nz, long long zstride,
                                                   unsigned block, float (*red)[33]) {
  const int e = threadIdx.x & 31;
  const int zl = threadIdx.x >> 5;
  const long long total = (long long)M * N;
  const long long idx = (long long)block * 32 + e;
  float v = 0.f;
  if (idx < total) {
#pragma unroll 4
    for (int z = zl; z < nz; z += 8) v += E.ws[(long long)z * zstride + idx];
  }
  red[zl][e] = v;
  __syncthreads();
  if (zl != 0 || idx >= total) return;
  v = ((red[0][e] + red[1][e]) + (red[2][e] + red[3][e])) + ((red[4][e] + red[5][e]) + (red[6][e] + red[7][e]));
  splitk_store(E, N, idx, v);
}

// The batched finish (mvk_defer_flush): V = 4 floats per thread with 16-byte loads when total and zstride are multiples
// of 4 and the slabs 16-byte aligned; ZL = 8 or 32 z-lanes (32 for long reductions: a thread's serial chain is nz / ZL
// loads).  A workgroup owns (256 / ZL) * V consecutive elements; per element the additions run in a fixed order: every
// z-lane over its slabs z = zl, zl + ZL, ..., then the lanes in groups of 8 as a balanced tree, then the groups in order.
template <int V>
__device__ __forceinline__ void splitk_reduce_bodyv(const Epilogue& E, int M, int N, int nz, long long zstride,
                                                    unsigned block, int zl_bits, float* red /* [ZL][EL*V + 1] */) {
  const int ZL = 1 << zl_bits, EL = 256 >> zl_bits;
  const int e = threadIdx.x & (EL - 1);
  const int zl = threadIdx.x >> (8 - zl_bits);
  const long long total = (long long)M * N;
  const long long idx = ((long long)block * EL + e) * V;
  float v[V];
#pragma unroll
  for (int c = 0; c < V; ++c) v[c] = 0.f;
  if (idx < total) {
#pragma unroll 8
    for (int z = zl; z < nz; z += ZL) {
      const float* src = E.ws + (long long)z * zstride + idx;
      if (V == 4) {
        const float4 t = *reinterpret_cast<const float4*>(src);
        v[0] += t.x;
        v[1 % V] += t.y;
        v[2 % V] += t.z;
        v[3 % V] += t.w;
      } else {
        v[0] += *src;
      }
    }
  }
  const int ld = EL * V + 1;
#pragma unroll
  for (int c = 0; c < V; ++c) red[zl * ld + e * V + c] = v[c];
  __syncthreads();
  if (zl != 0 || idx >= total) return;
#pragma unroll
  for (int c = 0; c < V; ++c) {
    float acc = 0.f;
    for (int g = 0; g < ZL; g += 8) {
      const float* r = red + g * ld + e * V + c;
      acc += ((r[0] + r[ld]) + (r[2 * ld] + r[3 * ld])) + ((r[4 * ld] + r[5 * ld]) + (r[6 * ld] + r[7 * ld]));
    }
    splitk_store(E, N, idx + c, acc);
  }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const Epilogue E, int M, int N, int nz) {
  __shared__ float red[8][33];
  splitk_reduce_body(E, M, N, nz, (long long)M * N, blockIdx.x, red);
}

// The same finish for a whole table of pending reductions in ONE launch (deferred leaf reductions, mvk_defer_begin):
// workgroups [blk0, next blk0) belong to item i.
struct DeferItem {
  Epilogue e;
  int M, N, nz;
  unsigned blk0;
  long long zstride;
  int vec4;     // 16-byte loads, 4 elements per thread
  int zl_bits;  // 3 or 5: 8 or 32 z-lanes
};
// What the finish needs of an item, packed: 40 of them fit the 4 KB kernel-argument limit (the headline step queues ~35: one
// launch instead of two back to back at the very end of the step)
struct DeferItemC {
  float* out;
  const float* ws;
  const float* bias;
  const float* act_src;
  long long ld, zstride;
  int M, N, nz;
  unsigned blk0;
  int Cu, bias_mod;
  short kind, act, src_act, taps;
  unsigned char atomic, vec4, zl_bits, pad;
};
constexpr int DEFER_BATCH = 40;
struct DeferTable {
  int n;
  DeferItemC it[DEFER_BATCH];
};
static_assert(sizeof(DeferTable) <= 4096, "kernel-argument limit");
inline DeferItemC defer_compact(const DeferItem& d) {
  DeferItemC c{};
  c.out = d.e.out, c.ws = d.e.ws, c.bias = d.e.bias, c.act_src = d.e.act_src;
  c.ld = d.e.ld, c.zstride = d.zstride;
  c.M = d.M, c.N = d.N, c.nz = d.nz, c.blk0 = d.blk0;
  c.Cu = d.e.Cu, c.bias_mod = d.e.bias_mod;
  c.kind = (short)d.e.kind, c.act = (short)d.e.act, c.src_act = (short)d.e.src_act, c.taps = (short)d.e.taps;
  c.atomic = (unsigned char)d.e.atomic, c.vec4 = (unsigned char)d.vec4, c.zl_bits = (unsigned char)d.zl_bits;
  return c;
}
__global__ __launch_bounds__(256) void splitk_reduce_batch_kernel(const DeferTable T) {
  __shared__ float red[32 * 33];  // [ZL][EL * V + 1]: (8, 129) or (32, 33) at most
  int i = 0;
  for (int j = 1; j < T.n; ++j)
    if (blockIdx.x >= T.it[j].blk0) i = j;
  const DeferItemC& it = T.it[i];
  Epilogue E{};
  E.out = it.out, E.ws = const_cast<float*>(it.ws), E.bias = it.bias, E.act_src = it.act_src;
  E.ld = it.ld, E.Cu = it.Cu, E.bias_mod = it.bias_mod;
  E.kind = it.kind, E.act = it.act, E.src_act = it.src_act, E.taps = it.taps, E.atomic = it.atomic;
  if (it.vec4)
    splitk_reduce_bodyv<4>(E, it.M, it.N, it.nz, it.zstride, blockIdx.x - it.blk0, it.zl_bits, red);
  else
    splitk_reduce_bodyv<1>(E, it.M, it.N, it.nz, it.zstride, blockIdx.x - it.blk0, it.zl_bits, red);
}

// The same launch with a bounded grid (a workgroup walks the table's workgroup indices grid by grid): for a PARTIAL flush that
// runs beside the step's last dependent chain — 11 k short workgroups take every free CU slot from the launches it runs beside,
// a few hundred long ones leave room (mvk_defer_flush; the final flush stays wide: it IS the chain).
__global__ __launch_bounds__(256) void splitk_reduce_batch_loop_kernel(const DeferTable T, const unsigned total) {
  __shared__ float red[32 * 33];
  for (unsigned b = blockIdx.x; b < total; b += gridDim.x) {
    int i = 0;
    for (int j = 1; j < T.n; ++j)
      if (b >= T.it[j].blk0) i = j;
    const DeferItemC& it = T.it[i];
    Epilogue E{};
    E.out = it.out, E.ws = const_cast<float*>(it.ws), E.bias = it.bias, E.act_src = it.act_src;
    E.ld = it.ld, E.Cu = it.Cu, E.bias_mod = it.bias_mod;
    E.kind = it.kind, E.act = it.act, E.src_act = it.src_act, E.taps = it.taps, E.atomic = it.atomic;
    if (it.vec4)
      splitk_reduce_bodyv<4>(E, it.M, it.N, it.nz, it.zstride, b - it.blk0, it.zl_bits, red);
    else
      splitk_reduce_bodyv<1>(E, it.M, it.N, it.nz, it.zstride, b - it.blk0, it.zl_bits, red);
    __syncthreads();  // `red` is reused by the next index
  }
}

// what the launcher chose (needed to finish fused column sums): rows of one tile; 0 = generic kernel
struct LaunchInfo {
  int bm;
};
int launch_igemm(const GemmDesc& d, int zdim, hipStream_t s, LaunchInfo* info = nullptr);
// db[n] += sum over `rows` partial rows (ordered, deterministic)
int colsum_finish(const float* part, int rows, int N, float* db, hipStream_t s);

int convref_reduce(const float* slab, int nz, int Cu, int Cv, int taps, float* dWref, hipStream_t s, bool deferred = false);
// smallcin.hip: direct kernels for the 4x4/stride-2 convolution of an NCHW image with <= 4 channels
bool smallcin_supported(int Cu, int Cv);
int smallcin_fwd(const float* U, const float* Wdown, const float* bias, float* V, int n, int h, int w, int Cu, int Cv,
                 int act, hipStream_t s);
int smallcin_wgrad(const float* U, const float* dV, float* dWref, int n, int h, int w, int Cu, int Cv, float* ws,
                   long long ws_floats, hipStream_t s);
// split-K launch: picks the slice count, uses the slab workspace when it is large enough (else atomics)
int launch_splitk(GemmDesc& d, float* ws, long long ws_floats, int target_blocks, hipStream_t s);

}  // namespace mvk
