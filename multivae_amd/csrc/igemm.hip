// C-ABI entry points built on the fp32 MFMA implicit-GEMM engine (igemm.hpp).
#include "igemm_bf.hpp"
#include <mutex>
#include <vector>

// ---- deferred leaf reductions ------------------------------------------------------------------------------------------
namespace mvk {
namespace {
struct DeferState {
  std::mutex mu;
  bool active = false;
  float* arena = nullptr;
  long long cap = 0, used = 0;
  long long wanted = 0;  // floats asked for since mvk_defer_begin, granted or not (mvk_defer_wanted: arena sizing)
  const char* g0 = nullptr;
  const char* g1 = nullptr;
  std::vector<DeferItem> items;
  std::vector<hipStream_t> streams;  // every stream a gradient producer ran on since the last flush
  std::vector<hipEvent_t> events;
};
DeferState g_defer;

void defer_note_stream(hipStream_t s) {
  for (hipStream_t t : g_defer.streams)
    if (t == s) return;
  g_defer.streams.push_back(s);
}
}  // namespace

float* defer_scratch(const void* out, long long floats, hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_defer.mu);
  const char* o = static_cast<const char*>(out);
  if (!g_defer.active || !out || o < g_defer.g0 || o >= g_defer.g1 || floats <= 0) return nullptr;
  defer_note_stream(s);  // also when declined: the flush orders itself behind every producer of the gradient buffer
  for (const DeferItem& it : g_defer.items)
    if (it.e.out == out) return nullptr;
  const long long need = (floats + 63) & ~63LL;  // 256-byte aligned regions
  g_defer.wanted += need;
  if (g_defer.used + need > g_defer.cap) return nullptr;
  float* p = g_defer.arena + g_defer.used;
  g_defer.used += need;
  return p;
}

bool defer_free(const void* out) {
  if (!out) return true;
  std::lock_guard<std::mutex> lock(g_defer.mu);
  const char* o = static_cast<const char*>(out);
  if (!g_defer.active || o < g_defer.g0 || o >= g_defer.g1) return false;
  for (const DeferItem& it : g_defer.items)
    if (it.e.out == out) return false;
  return true;
}

static int defer_push(const Epilogue& e, int M, int N, int nz, long long zstride) {
  std::lock_guard<std::mutex> lock(g_defer.mu);
  DeferItem it{};
  it.e = e;
  it.M = M;
  it.N = N;
  it.nz = nz;
  it.zstride = zstride;
  it.vec4 = ((long long)M * N) % 4 == 0 && zstride % 4 == 0 && mvk_aligned16(e.ws);
  static const int zl_env = mvk_tune("MVK_DEFER_ZL") ? atoi(mvk_tune("MVK_DEFER_ZL")) : 5;  // z-lanes of long reductions (A/B)
  it.zl_bits = nz > 64 ? zl_env : 3;
  g_defer.items.push_back(it);
  return MVK_OK;
}

int defer_push_plain(float* out, const float* part, long long count, int nz, long long zstride, hipStream_t) {
  if (count > 0x7fffffffLL) return MVK_EINVAL;
  Epilogue e{};
  e.out = out;
  e.kind = E_ROWMAJOR;
  e.ld = count;
  e.bias_mod = 1;
  e.atomic = 1;
  e.Cu = e.OH = e.OW = 1;
  e.ws = const_cast<float*>(part);
  return defer_push(e, 1, (int)count, nz, zstride);
}
}  // namespace mvk

extern "C" int mvk_defer_begin(float* arena, int64_t arena_floats, const float* grad, int64_t grad_floats) {
  using namespace mvk;
  std::lock_guard<std::mutex> lock(g_defer.mu);
  if (!g_defer.items.empty()) return MVK_EINVAL;  // pending finishes: flush first
  if (!arena || arena_floats <= 0 || !grad || grad_floats <= 0 || !mvk_aligned16(arena)) return MVK_EINVAL;
  g_defer.active = true;
  g_defer.arena = arena;
  g_defer.cap = arena_floats;
  g_defer.used = 0;
  g_defer.wanted = 0;
  g_defer.g0 = reinterpret_cast<const char*>(grad);
  g_defer.g1 = g_defer.g0 + sizeof(float) * (size_t)grad_floats;
  g_defer.streams.clear();
  return MVK_OK;
}

static int defer_flush_locked(hipStream_t s, bool last) {
  using namespace mvk;
  // order this stream behind every stream a gradient producer ran on (autograd replays a branch's backward on the
  // stream of its forward; an earlier partial flush counts as a producer)
  size_t ev = 0;
  for (hipStream_t t : g_defer.streams) {
    if (t == s) continue;
    if (ev == g_defer.events.size()) {
      hipEvent_t e;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return MVK_ELAUNCH;
      g_defer.events.push_back(e);
    }
    if (hipEventRecord(g_defer.events[ev], t) != hipSuccess) return MVK_ELAUNCH;
    if (hipStreamWaitEvent(s, g_defer.events[ev], 0) != hipSuccess) return MVK_ELAUNCH;
    ++ev;
  }
  g_defer.streams.clear();
  int rc = MVK_OK;
  for (size_t i0 = 0; i0 < g_defer.items.size() && rc == MVK_OK; i0 += DEFER_BATCH) {
    DeferTable T{};
    unsigned blocks = 0;
    for (size_t i = i0; i < g_defer.items.size() && i < i0 + DEFER_BATCH; ++i) {
      DeferItem it = g_defer.items[i];
      if (it.e.res || it.e.colsum_part) return MVK_EINVAL;  // not representable in the packed item (never queued)
      it.blk0 = blocks;
      T.it[T.n++] = defer_compact(it);
      const int per = (256 >> it.zl_bits) * (it.vec4 ? 4 : 1);
      blocks += (unsigned)(((long long)it.M * it.N + per - 1) / per);
    }
    // MVK_DEFER_FLUSH_GRID=n: a PARTIAL flush (it runs beside the step's chain) on at most n workgroups (A/B; 0 = one per index)
    static const int part_grid = mvk_tune("MVK_DEFER_FLUSH_GRID") ? atoi(mvk_tune("MVK_DEFER_FLUSH_GRID")) : 0;
    if (!last && part_grid > 0 && blocks > (unsigned)part_grid)
      hipLaunchKernelGGL(splitk_reduce_batch_loop_kernel, dim3(part_grid), dim3(256), 0, s, T, blocks);
    else
      hipLaunchKernelGGL(splitk_reduce_batch_kernel, dim3(blocks), dim3(256), 0, s, T);
    if (hipGetLastError() != hipSuccess) rc = MVK_ELAUNCH;
  }
  g_defer.items.clear();
  if (last)
    g_defer.used = 0;
  else
    defer_note_stream(s);  // the arena regions stay reserved (this launch may still read them) and the next flush waits for it
  return rc;
}

extern "C" int mvk_defer_flush(void* stream) {
  std::lock_guard<std::mutex> lock(mvk::g_defer.mu);
  if (!mvk::g_defer.active) return MVK_OK;
  return defer_flush_locked(mvk_stream(stream), false);
}

extern "C" int mvk_defer_end(void* stream) {
  std::lock_guard<std::mutex> lock(mvk::g_defer.mu);
  if (!mvk::g_defer.active) return MVK_OK;
  const int rc = defer_flush_locked(mvk_stream(stream), true);
  mvk::g_defer.active = false;
  return rc;
}

extern "C" int64_t mvk_defer_wanted(void) {
  std::lock_guard<std::mutex> lock(mvk::g_defer.mu);
  return (int64_t)mvk::g_defer.wanted;
}

extern "C" int mvk_defer_pending(void) {
  std::lock_guard<std::mutex> lock(mvk::g_defer.mu);
  return (int)mvk::g_defer.items.size();
}

#include <cstdlib>

namespace mvk {

template <int BM, int BN>
static int launch_cfg(const GemmDesc& d, int zdim, hipStream_t s) {
  dim3 grid((d.M + BM - 1) / BM, (d.N + BN - 1) / BN, zdim);
  hipLaunchKernelGGL((igemm_kernel<BM, BN>), grid, dim3(256), 0, s, d);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

// ---- mode-specialised kernels (igemm_fast.hpp) -------------------------------------------------------------------
static thread_local int g_last_bm = 0;  // tile rows of the last specialised launch (LaunchInfo)

template <int BM, int BN, int BKT, int AMODE, int BMODE, bool AACT>
static int launch_fast_cfg(const GemmDesc& d, int zdim, hipStream_t s) {
  g_last_bm = BM;
  dim3 grid((d.M + BM - 1) / BM, (d.N + BN - 1) / BN, zdim);
  hipLaunchKernelGGL((igemm_fast_kernel<BM, BN, BKT, AMODE, BMODE, AACT>), grid, dim3(256), 0, s, d);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

template <int BM, int BN, int BKT>
static int launch_fast_tile(const GemmDesc& d, int zdim, hipStream_t s, int amode, int bmode, bool aact) {
  if (amode == AM_PLAIN_K && bmode == BM_K && !aact) return launch_fast_cfg<BM, BN, BKT, AM_PLAIN_K, BM_K, false>(d, zdim, s);
  if (amode == AM_PLAIN_K && bmode == BM_N && !aact) return launch_fast_cfg<BM, BN, BKT, AM_PLAIN_K, BM_N, false>(d, zdim, s);
  if (amode == AM_PLAIN_K && bmode == BM_N && aact) return launch_fast_cfg<BM, BN, BKT, AM_PLAIN_K, BM_N, true>(d, zdim, s);
  if (amode == AM_PLAIN_R && bmode == BM_N && !aact) return launch_fast_cfg<BM, BN, BKT, AM_PLAIN_R, BM_N, false>(d, zdim, s);
  if (amode == AM_PLAIN_R && bmode == BM_N && aact) return launch_fast_cfg<BM, BN, BKT, AM_PLAIN_R, BM_N, true>(d, zdim, s);
  if (amode == AM_ROW && bmode == BM_N && !aact) return launch_fast_cfg<BM, BN, BKT, AM_ROW, BM_N, false>(d, zdim, s);
  if (amode == AM_COL && bmode == BM_N && !aact) return launch_fast_cfg<BM, BN, BKT, AM_COL, BM_N, false>(d, zdim, s);
  return 1;  // combination not instantiated -> generic kernel
}

// ---- split-precision kernels (igemm_bf.hpp) ----------------------------------------------------------------------
template <int BM, int BN, int AMODE, int BMODE, bool AACT>
static int launch_bf_cfg(const GemmDesc& d, int zdim, hipStream_t s) {
  g_last_bm = BM;
  dim3 grid((d.M + BM - 1) / BM, (d.N + BN - 1) / BN, zdim);
  constexpr int DEPTH = 2;
  hipLaunchKernelGGL((igemm_bf_kernel<BM, BN, AMODE, BMODE, AACT, DEPTH>), grid, dim3(256), 0, s, d);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

template <int BM, int BN>
static int launch_bf_tile(const GemmDesc& d, int zdim, hipStream_t s, int amode, int bmode, bool aact) {
  if (BM * BN > 128 * 64) {  // large tile: weight-gradient GEMMs only
    if (amode == AM_COL && bmode == BM_N && !aact) return launch_bf_cfg<BM, BN, AM_COL, BM_N, false>(d, zdim, s);
    return 1;
  }
  if (amode == AM_PLAIN_K && bmode == BM_K && !aact) return launch_bf_cfg<BM, BN, AM_PLAIN_K, BM_K, false>(d, zdim, s);
  if (amode == AM_PLAIN_K && bmode == BM_N && !aact) return launch_bf_cfg<BM, BN, AM_PLAIN_K, BM_N, false>(d, zdim, s);
  if (amode == AM_PLAIN_K && bmode == BM_N && aact) return launch_bf_cfg<BM, BN, AM_PLAIN_K, BM_N, true>(d, zdim, s);
  if (amode == AM_PLAIN_R && bmode == BM_N && !aact) return launch_bf_cfg<BM, BN, AM_PLAIN_R, BM_N, false>(d, zdim, s);
  if (amode == AM_PLAIN_R && bmode == BM_N && aact) return launch_bf_cfg<BM, BN, AM_PLAIN_R, BM_N, true>(d, zdim, s);
  if (amode == AM_ROW && bmode == BM_N && !aact) return launch_bf_cfg<BM, BN, AM_ROW, BM_N, false>(d, zdim, s);
  if (amode == AM_COL && bmode == BM_N && !aact) return launch_bf_cfg<BM, BN, AM_COL, BM_N, false>(d, zdim, s);
  if (amode == AM_ROW3 && bmode == BM_N && !aact) return launch_bf_cfg<BM, BN, AM_ROW3, BM_N, false>(d, zdim, s);
  return 1;
}

static int bf_min_blocks() {  // smallest grid that still goes to the split engine (MVK_BF_MIN_BLOCKS overrides)
  static int v = -1;
  if (v < 0) {
    const char* e = mvk_tune("MVK_BF_MIN_BLOCKS");
    v = e ? atoi(e) : 256;  // measured on the MnistSvhn step: 160-256 best, 384 and < 100 1-2 % slower
  }
  return v;
}

static int g_engine = -1;  // 1 = split-bf16 MFMA (default), 0 = fp32 MFMA (MVK_ENGINE=f32)
static int engine() {
  if (g_engine < 0) {
    const char* e = getenv("MVK_ENGINE");
    g_engine = (e && e[0] == 'f') ? 0 : 1;
  }
  return g_engine;
}

// returns 1 when no specialised kernel applies
static int try_launch_fast(const GemmDesc& d, int zdim, hipStream_t s) {
  const AOperand& A = d.a;
  // buffer loads use 32-bit byte offsets: stay on the generic kernel for operands of 2 GiB or more
  const long long lim = 1ll << 29;
  if (A.kind == A_PLAIN) {
    if ((long long)d.M * d.K >= lim) return 1;
  } else if ((long long)d.a.H * d.a.W * d.a.C * ((long long)d.M / (d.a.OH * d.a.OW > 0 ? d.a.OH * d.a.OW : 1) + 1) >= lim &&
             !d.a.trans) {
    return 1;
  } else if (d.a.trans && (long long)d.a.H * d.a.W * d.a.C * ((long long)d.K / (d.a.OH * d.a.OW > 0 ? d.a.OH * d.a.OW : 1) + 1) >= lim) {
    return 1;
  }
  if ((long long)d.K * d.N >= lim) return 1;
  int amode = 0, bmode = 0;
  const bool aact = A.act_src != nullptr;
  if (A.kind == A_PLAIN) {
    if (!A.vec4) return 1;
    if (aact && !mvk_aligned16(A.act_src)) return 1;
    amode = A.contig_k ? AM_PLAIN_K : AM_PLAIN_R;
  } else if (A.bf3) {
    // pre-split operand: only the split engine reads it (row gather, 32-channel k-tiles, 16-byte aligned planes)
    if (A.trans || !A.contig_k || A.kind == A_DOWN_NCHW || aact || A.C % 32 != 0 || !mvk_aligned16(A.p) ||
        A.plane_bytes % 16 != 0)
      return MVK_EINVAL;
    amode = AM_ROW3;
  } else if (!A.trans && A.vec4 && A.contig_k && A.kind != A_DOWN_NCHW && !aact) {
    amode = AM_ROW;
  } else if (A.trans && A.vec4 && !A.contig_k && A.kind == A_DOWN && !aact) {
    amode = AM_COL;
  } else {
    return 1;
  }
  if (!d.b.vec4) return 1;
  bmode = d.b.contig_k ? BM_K : BM_N;
  if (d.zmode == Z_SPLITK && (d.ksplit_tiles & 1)) return 1;  // BKT = 32 needs 32-aligned slices
  const bool c32 = (amode != AM_ROW) || (A.C % 32 == 0);
  const bool c16 = (amode != AM_ROW) || (A.C % 16 == 0);
  if (amode == AM_ROW3 && !d.b.vec4) return MVK_EINVAL;
  if ((engine() == 1 || amode == AM_ROW3) && c32) {
    auto nb = [&](int bm, int bn) { return (long long)((d.M + bm - 1) / bm) * ((d.N + bn - 1) / bn) * zdim; };
    int rc = 1;
    // measured on the step's convolutions: 128x128 only pays for the weight-gradient GEMMs (both operands staged
    // through the register transpose), 256x32 never beats 128x32 (one MFMA tile per wave, two accumulators)
    if (d.N <= 32) {
      rc = launch_bf_tile<128, 32>(d, zdim, s, amode, bmode, aact);
    } else if (d.N > 64 && amode == AM_COL && nb(128, 128) >= 384) {
      rc = launch_bf_tile<128, 128>(d, zdim, s, amode, bmode, aact);
    } else if (nb(128, 64) >= bf_min_blocks() || amode == AM_ROW3) {
      rc = launch_bf_tile<128, 64>(d, zdim, s, amode, bmode, aact);
    }
    if (rc != 1) return rc;
  }
  if (amode == AM_ROW3) return MVK_EINVAL;  // no exact-fp32 kernel reads pre-split operands
  if (d.N <= 32) {
    // tall-skinny: a 256-row tile lets every wave reuse its B fragment for two MFMA tiles
    if (c16 && (long long)((d.M + 255) / 256) * zdim >= 512) return launch_fast_tile<256, 32, 16>(d, zdim, s, amode, bmode, aact);
    return c32 ? launch_fast_tile<128, 32, 32>(d, zdim, s, amode, bmode, aact) : 1;
  }
  // pick the largest tile that still gives >= 1.5 workgroups per CU (384); small problems take 64x64
  auto nblk = [&](int bm, int bn) { return (long long)((d.M + bm - 1) / bm) * ((d.N + bn - 1) / bn) * zdim; };
  if (d.N > 64 && c16 && nblk(128, 128) >= 384) return launch_fast_tile<128, 128, 16>(d, zdim, s, amode, bmode, aact);
  if (!c32) return 1;
  if (nblk(128, 64) >= 384) return launch_fast_tile<128, 64, 32>(d, zdim, s, amode, bmode, aact);
  return launch_fast_tile<64, 64, 32>(d, zdim, s, amode, bmode, aact);
}

static unsigned long long* g_dbg = nullptr;
static int g_dbg_flags = 0;
// Measurement hooks (declared in mvk.h): only experiment builds (-DMVK_PHASES / -DMVK_EXPER, tools/build_exper.sh,
// tools/build_variants.sh) read what they set; in the shipped library they are inert.
extern "C" void mvk_debug_set_phase_buffer(unsigned long long* p) { g_dbg = p; }
extern "C" void mvk_debug_set_flags(int f) {
  g_dbg_flags = f;
#ifdef MVK_EXPER
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bf_flags), &f, sizeof(int));
#endif
}

int launch_igemm(const GemmDesc& d_in, int zdim, hipStream_t s, LaunchInfo* info) {
  GemmDesc d = d_in;
  if (info) info->bm = 0;
  d.dbg = g_dbg;
  d.dbg_flags = g_dbg_flags;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0) return MVK_OK;
  {
    g_last_bm = 0;
    const int rc = try_launch_fast(d, zdim, s);
    if (rc != 1) {
      if (info) info->bm = g_last_bm;
      return rc;
    }
    d.e.colsum_part = nullptr;  // generic kernels: the caller falls back to a separate column-sum launch
  }
  if (d.N <= 32) return launch_cfg<128, 32>(d, zdim, s);
  if (d.N <= 64) {
    if (d.M <= 64) return launch_cfg<64, 64>(d, zdim, s);
    return launch_cfg<128, 64>(d, zdim, s);
  }
  long long big = (long long)((d.M + 127) / 128) * ((d.N + 127) / 128) * zdim;
  if (big >= 192) return launch_cfg<128, 128>(d, zdim, s);
  long long mid = (long long)((d.M + 127) / 128) * ((d.N + 63) / 64) * zdim;
  if (mid >= 192 || d.M > 4096) return launch_cfg<128, 64>(d, zdim, s);
  return launch_cfg<64, 64>(d, zdim, s);
}

// experiment hook: MVK_SPLITK_TARGET_1024 / _512 replace the two block targets of the split-K launches
// SPLITK_C4 / SPLITK_LIN: the weight gradients of the 4x4/stride-2 convolutions / of the Linear layers (targets of their own since
// round 5: MVK_SPLITK_TARGET_C4, MVK_SPLITK_TARGET_LIN)
enum { SPLITK_C4 = 1025, SPLITK_LIN = 1026 };
static int splitk_target(int t) {
  static int t1024 = -1, t512 = -1, tc4 = -1, tlin = -1;
  if (t1024 < 0) {
    const char* a = mvk_tune("MVK_SPLITK_TARGET_1024");
    const char* b = mvk_tune("MVK_SPLITK_TARGET_512");
    const char* c = mvk_tune("MVK_SPLITK_TARGET_C4");
    const char* l = mvk_tune("MVK_SPLITK_TARGET_LIN");
    t1024 = a ? atoi(a) : 768;  // one full wave of 3 workgroups per CU (A/B in the step: 1.955 vs 1.969 ms at 1024)
    t512 = b ? atoi(b) : 512;
    // 4x4/stride-2 weight gradients at the encoder batch sit on the step's last dependent chain, beside the ordered finish of the
    // decoders' partial sums: 256 workgroups (16 tiles x 16 slices at 64 -> 128 channels, 4 x 64 at 32 -> 64) instead of 768 write a
    // third of the partial tiles and leave the chip to their neighbours — headline 1.0005 -> 0.992 ms, MMVAE MnistSvhn 0.581 ->
    // 0.548 ms (tools/lab/r05/gpu_r05_k2.sh / _m2.sh; 128 and 192 are slower again)
    tc4 = c ? atoi(c) : 256;
    tlin = l ? atoi(l) : t1024;
  }
  return t == 1024 ? t1024 : (t == 512 ? t512 : (t == SPLITK_C4 ? tc4 : (t == SPLITK_LIN ? tlin : t)));
}

int launch_splitk(GemmDesc& d, float* ws, long long ws_floats, int target_blocks, hipStream_t s) {
  target_blocks = splitk_target(target_blocks);
  int tm = (d.M + 127) / 128;
  int tn = (d.N <= 32) ? 1 : (d.N + 63) / 64;
  int tiles = tm * tn;
  int ktiles = (d.K + BK - 1) / BK;
  int splits = target_blocks / (tiles > 0 ? tiles : 1);
  if (splits < 1) splits = 1;
  if (splits > ktiles) splits = ktiles;
  d.zmode = Z_SPLITK;
  d.ksplit_tiles = (ktiles + splits - 1) / splits;
  d.ksplit_tiles += d.ksplit_tiles & 1;  // 32-element aligned slices (BKT = 32 kernels)
  int z = (ktiles + d.ksplit_tiles - 1) / d.ksplit_tiles;
  const long long need = (long long)z * d.M * d.N;
  if (z > 1 && d.e.atomic && d.e.kind != E_UP && d.e.kind != E_UP_NCHW) {
    // a gradient accumulated into the registered buffer: private slabs, the ordered finish is queued (mvk_defer_flush)
    if (float* dws = defer_scratch(d.e.out, need, s)) {
      d.e.ws = dws;
      const int rc = launch_igemm(d, z, s);
      return rc ? rc : defer_push(d.e, d.M, d.N, z, (long long)d.M * d.N);
    }
  }
  if (z > 1 && ws && ws_floats >= need && d.e.kind != E_UP && d.e.kind != E_UP_NCHW) {
    d.e.ws = ws;
    int rc = launch_igemm(d, z, s);
    if (rc) return rc;
    long long total = (long long)d.M * d.N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, s, d.e, d.M, d.N, z);
    MVK_CHECK_LAUNCH();
    return MVK_OK;
  }
  d.e.ws = nullptr;
  return launch_igemm(d, z, s);
}

static void plain_a(AOperand& a, const float* p, long long sr, long long sk, int M, int K) {
  a = AOperand{};
  a.p = p;
  a.kind = A_PLAIN;
  a.sr = sr;
  a.sk = sk;
  a.contig_k = (sk == 1);
  if (sk == 1)
    a.vec4 = (sr % 4 == 0) && (K % 4 == 0) && mvk_aligned16(p);
  else
    a.vec4 = (sr == 1) && (sk % 4 == 0) && (M % 4 == 0) && mvk_aligned16(p);
  a.C = a.H = a.W = a.OH = a.OW = 1;
}

static void plain_b(BOperand& b, const float* p, long long sk, long long sn, int K, int N) {
  b = BOperand{};
  b.p = p;
  b.sk = sk;
  b.sn = sn;
  b.contig_k = (sk == 1);
  if (sk == 1)
    b.vec4 = (sn % 4 == 0) && (K % 4 == 0) && mvk_aligned16(p);
  else
    b.vec4 = (sn == 1) && (sk % 4 == 0) && (N % 4 == 0) && mvk_aligned16(p);
}

static void rowmajor_epi(Epilogue& e, float* out, long long ld) {
  e = Epilogue{};
  e.out = out;
  e.kind = E_ROWMAJOR;
  e.ld = ld;
  e.bias_mod = 1;
  e.Cu = e.OH = e.OW = 1;
}

// ---- small helper kernels ------------------------------------------------------------------------------
// db[n] += sum_m dY[m][n] (* act'(Y)).  Threads = (N/vec column groups) x (row lanes); 16-byte coalesced loads when
// N % 4 == 0; partials combined through LDS.  With `part` every workgroup row writes its partial sums to
// part[blockIdx.y][N] and an ordered finish adds them to db (deterministic); without scratch: one atomicAdd per column
// per workgroup.
template <int VEC>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dY, const float* __restrict__ Y, int act,
                                                     int M, int N, int rows_per_block, float* __restrict__ db,
                                                     float* __restrict__ part) {
  const int ncg = (N + VEC - 1) / VEC;            // column groups
  const int cg_per_blk = ncg < 256 ? ncg : 256;   // column groups handled by this block (blockIdx.x strides them)
  const int lanes_r = 256 / cg_per_blk;           // row lanes
  const int cg = blockIdx.x * cg_per_blk + (threadIdx.x % cg_per_blk);
  const int rl = threadIdx.x / cg_per_blk;
  const int r0 = blockIdx.y * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  float acc[VEC];
#pragma unroll
  for (int c = 0; c < VEC; ++c) acc[c] = 0.f;
  if (cg < ncg && rl < lanes_r) {
#pragma unroll 8
    for (int r = r0 + rl; r < r1; r += lanes_r) {
      const long long o = (long long)r * N + (long long)cg * VEC;
      if (VEC == 4) {
        float4 v = *reinterpret_cast<const float4*>(dY + o);
        if (Y) {
          float4 y = *reinterpret_cast<const float4*>(Y + o);
          v.x *= mvk_act_grad_from_out(y.x, act);
          v.y *= mvk_act_grad_from_out(y.y, act);
          v.z *= mvk_act_grad_from_out(y.z, act);
          v.w *= mvk_act_grad_from_out(y.w, act);
        }
        acc[0] += v.x;
        acc[1 % VEC] += v.y;
        acc[2 % VEC] += v.z;
        acc[3 % VEC] += v.w;
      } else {
        float v = dY[o];
        if (Y) v *= mvk_act_grad_from_out(Y[o], act);
        acc[0] += v;
      }
    }
  }
  __shared__ float red[256 * VEC];
#pragma unroll
  for (int c = 0; c < VEC; ++c) red[threadIdx.x * VEC + c] = acc[c];
  __syncthreads();
  if (rl == 0 && cg < ncg) {
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      float t = 0.f;
      for (int q = 0; q < lanes_r; ++q) t += red[(q * cg_per_blk + (threadIdx.x % cg_per_blk)) * VEC + c];
      const int col = cg * VEC + c;
      if (col < N) {
        if (part) part[(long long)blockIdx.y * N + col] = t;
        else atomicAdd(db + col, t);
      }
    }
  }
}

// db[n] += sum_r part[r][n], r in order: one thread per column (any N)
__global__ __launch_bounds__(256) void colsum_finish_scalar_kernel(const float* __restrict__ part, int rows, int N,
                                                                   float* __restrict__ db) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  float t = 0.f;
  for (int r = 0; r < rows; ++r) t += part[(long long)r * N + c];
  db[c] += t;
}

// db[c] += sum over (n, p) of dY[n,c,p] (* act'(Y)) for NCHW tensors; one block per (c, chunk of n)
__global__ void nchw_channel_sum_kernel(const float* __restrict__ dY, const float* __restrict__ Y, int act, int n,
                                        int c, int hw, int imgs_per_block, float* __restrict__ db,
                                        float* __restrict__ part) {
  const int ch = blockIdx.x;
  const int i0 = blockIdx.y * imgs_per_block;
  int i1 = i0 + imgs_per_block;
  if (i1 > n) i1 = n;
  float s = 0.f;
  for (int img = i0; img < i1; ++img) {
    const long long base = ((long long)img * c + ch) * hw;
    for (int p = threadIdx.x; p < hw; p += blockDim.x) {
      float v = dY[base + p];
      if (Y) v *= mvk_act_grad_from_out(Y[base + p], act);
      s += v;
    }
  }
  s = wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = red[0] + red[1] + red[2] + red[3];
    if (part) part[(long long)blockIdx.y * c + ch] = t;
    else atomicAdd(db + ch, t);
  }
}

int colsum_finish(const float* part, int rows, int N, float* db, hipStream_t s);
static int colsum_finish_any(const float* part, int rows, int N, float* db, hipStream_t s);

// ws: scratch for the per-workgroup partial sums (deterministic ordered finish); without it: fp32 atomics
static int colsum(const float* dY, const float* Y, int act, int M, int N, float* db, float* ws, long long ws_floats,
                  hipStream_t s) {
  if (M <= 0 || N <= 0) return MVK_OK;
  const bool vec = (N % 4 == 0) && mvk_aligned16(dY) && (!Y || mvk_aligned16(Y));
  const int ncg = vec ? N / 4 : N;
  const int cgb = ncg < 256 ? ncg : 256;
  const int gx = (ncg + cgb - 1) / cgb;
  // ~256 workgroups over the rows (few atomics per column), at least 64 rows each
  int rows_per_block = (M + 255) / 256;
  if (rows_per_block < 16) rows_per_block = 16;
  const int gy = (M + rows_per_block - 1) / rows_per_block;
  float* dpart = defer_scratch(db, (long long)gy * N, s);
  float* part = dpart ? dpart : ((ws && (long long)gy * N <= ws_floats) ? ws : nullptr);
  if (vec)
    hipLaunchKernelGGL((colsum_kernel<4>), dim3(gx, gy), dim3(256), 0, s, dY, Y, act, M, N, rows_per_block, db, part);
  else
    hipLaunchKernelGGL((colsum_kernel<1>), dim3(gx, gy), dim3(256), 0, s, dY, Y, act, M, N, rows_per_block, db, part);
  MVK_CHECK_LAUNCH();
  if (dpart) return defer_push_plain(db, dpart, N, gy, N, s);
  return part ? colsum_finish_any(part, gy, N, db, s) : MVK_OK;
}

// db[n] += sum_r part[r][n] in a fixed order: 64 row lanes x 4 column quads per workgroup (float4 loads), LDS tree.
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, int rows, int N,
                                                            float* __restrict__ db) {
  const int q = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int c = (blockIdx.x * 4 + q) * 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < N) {
#pragma unroll 8
    for (int r = rl; r < rows; r += 64) {
      const float4 v = *reinterpret_cast<const float4*>(part + (long long)r * N + c);
      a.x += v.x;
      a.y += v.y;
      a.z += v.z;
      a.w += v.w;
    }
  }
  __shared__ float4 red[64][4];
  red[rl][q] = a;
  __syncthreads();
  for (int st = 32; st >= 1; st >>= 1) {
    if (rl < st) {
      const float4 o = red[rl + st][q];
      float4 m = red[rl][q];
      m.x += o.x;
      m.y += o.y;
      m.z += o.z;
      m.w += o.w;
      red[rl][q] = m;
    }
    __syncthreads();
  }
  if (rl == 0 && c < N) {
    const float4 t = red[0][q];
    db[c] += t.x;
    db[c + 1] += t.y;
    db[c + 2] += t.z;
    db[c + 3] += t.w;
  }
}

int colsum_finish(const float* part, int rows, int N, float* db, hipStream_t s) {
  // N % 4 == 0 and 16-byte aligned partials are guaranteed by epilogue_vec_ok + the scratch allocation
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((N + 15) / 16), dim3(256), 0, s, part, rows, N, db);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

static int colsum_finish_any(const float* part, int rows, int N, float* db, hipStream_t s) {
  if (N % 4 == 0 && mvk_aligned16(part)) return colsum_finish(part, rows, N, db, s);
  hipLaunchKernelGGL(colsum_finish_scalar_kernel, dim3((N + 255) / 256), dim3(256), 0, s, part, rows, N, db);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

int convref_reduce(const float* slab, int nz, int Cu, int Cv, int taps, float* dWref, hipStream_t s, bool deferred) {
  Epilogue e{};
  e.out = dWref;
  e.kind = E_CONVREF;
  e.taps = taps;
  e.bias_mod = 1;
  e.atomic = 1;  // accumulate
  e.Cu = Cu;
  e.OH = e.OW = 1;
  e.ws = const_cast<float*>(slab);
  const int M = taps * Cu;
  if (deferred) return defer_push(e, M, Cv, nz, (long long)M * Cv);  // slab from defer_scratch(dWref, ...)
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((M * Cv + 31) / 32), dim3(256), 0, s, e, M, Cv, nz);
  MVK_CHECK_LAUNCH();
  return MVK_OK;
}

// dPre[m][n] = dY[m][n] * act'(Y[m][n]) and per-workgroup column sums of dPre in ONE pass over dY / Y (the output-layer
// backward of the MLP decoders: the pre-activation gradient is needed by the weight- and data-gradient GEMMs and its
// column sums are the bias gradient).  part[(block)][N]; rows_per_block rows per workgroup; N % 4 == 0.
__global__ void act_bwd_plain_kernel(float* __restrict__ dY, const float* __restrict__ Y, long long n, int act) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dY[i] *= mvk_act_grad_from_out(Y[i], act);
}

__global__ __launch_bounds__(256) void act_bwd_colsum_kernel(const float* __restrict__ dY, const float* __restrict__ Y,
                                                             int act, int M, int N, int rows_per_block,
                                                             float* __restrict__ dPre, float* __restrict__ part) {
  const int nq = N >> 2;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  for (int q = threadIdx.x; q < nq; q += 256) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int r = r0; r < r1; ++r) {
      const long long o = (long long)r * N + 4 * q;
      float4 v = *reinterpret_cast<const float4*>(dY + o);
      const float4 y = *reinterpret_cast<const float4*>(Y + o);
      v.x *= mvk_act_grad_from_out(y.x, act);
      v.y *= mvk_act_grad_from_out(y.y, act);
      v.z *= mvk_act_grad_from_out(y.z, act);
      v.w *= mvk_act_grad_from_out(y.w, act);
      *reinterpret_cast<float4*>(dPre + o) = v;
      acc.x += v.x;
      acc.y += v.y;
      acc.z += v.z;
      acc.w += v.w;
    }
    *reinterpret_cast<float4*>(part + (long long)blockIdx.x * N + 4 * q) = acc;
  }
}

// GEMM launch + `db[n] += column sums of the stored output` (bias gradient of the layer that produced the GEMM's
// input gradient).  Fused into the epilogue when the vectorised epilogue applies and the partials fit in ws;
// otherwise a separate pass over the output.
static int launch_with_colsum(GemmDesc& d, int zdim, float* db, float* ws, int64_t ws_floats, const float* out,
                              long long out_rows, hipStream_t s) {
  if (!db) return launch_igemm(d, zdim, s);
  const long long max_rows = (long long)((d.M + 63) / 64) * zdim;
  float* dpart = epilogue_vec_ok(d.e, d.N) ? defer_scratch(db, max_rows * d.N, s) : nullptr;
  float* cws = dpart ? dpart : ws;
  const bool fuse = cws && mvk_aligned16(cws) && epilogue_vec_ok(d.e, d.N) && (dpart || max_rows * d.N <= ws_floats);
  LaunchInfo info{};
  d.e.colsum_part = fuse ? cws : nullptr;
  int rc = launch_igemm(d, zdim, s, &info);
  d.e.colsum_part = nullptr;
  if (rc != MVK_OK) return rc;
  if (fuse && info.bm > 0) {
    const int rows = ((d.M + info.bm - 1) / info.bm) * zdim;
    return dpart ? defer_push_plain(db, dpart, d.N, rows, d.N, s) : colsum_finish(cws, rows, d.N, db, s);
  }
  return colsum(out, nullptr, 0, (int)out_rows, d.N, db, ws, ws_floats, s);
}

}  // namespace mvk

using namespace mvk;

namespace mvk {
// conv3small.hip: 3x3 convolutions with an image (<= 4 channels) on one side: 1 = shape not covered
int conv3_smallcin(const float* X, const float* Wp, const float* bias, float* Y, int n, int H, int W, int Cin, int Cout, int act,
                   const float* mask_src, int mask_act, hipStream_t s, float* y_amax = nullptr);
int conv3_smallcout(const float* X, const float* Wp, const float* bias, float* Y, int n, int H, int W, int Cin, int Cout, int act,
                    const float* mask_src, int mask_act, hipStream_t s);
int conv3_small_wgrad(const float* X, const float* dY, float* slab, long long slab_floats, int n, int H, int W, int Cin, int Cout,
                      int* nz, hipStream_t s);
// skinny.hip: few-row linear layer (a workgroup per 16 x 16 output tile, its waves split K): 1 = shape not covered
int heads_launch(const float* X, const float* W0, const float* b0, float* Y0, const float* W1, const float* b1, float* Y1,
                 int M, int N, int K, long long w_sk, long long w_sn, int act, hipStream_t s, float* ws = nullptr,
                 long long ws_floats = 0);
// skinny.hip: backward of 1 or 2 narrow heads in one launch (slabs per 128-row group); 1 = shape not covered
int heads_bwd_launch(const float* X, int x_act, const float* dY0, const float* dY1, const float* W0, const float* W1,
                     long long w_sk, long long w_sn, int flat_c, float* dX, float* wslab0, float* wslab1, float* bslab0,
                     float* bslab1, float* pslab, int M, int N, int K, int* nz, hipStream_t s);
// skinny.hip: short-reduction linear layer (K <= 32): 1 = shape not covered
int smallk_fwd(const float* X, const float* W, long long w_sk, long long w_sn, const float* bias, int bias_mod, int act,
               float* Y, int M, int N, int K, hipStream_t s, const float* mask_src = nullptr, int mask_act = 0,
               int accumulate = 0, float* y_amax = nullptr);
// imgconv.hip: register-stationary-weight kernels for the 4x4/stride-2 layer pairs (1 = shape not covered)
int imgconv_up(const float* V, const float* Wup, const void* wfrag, const float* bias, float* U, int n, int h, int w, int Cu,
               int Cv, int act, const float* u_act_src, int u_act, float* colsum_part, int* part_rows, const float* x_amax,
               const float* w_amax, float* y_amax, hipStream_t s);
int imgconv_down(const float* U, const float* Wdown, const void* wfrag, const float* bias, float* V, int n, int h, int w,
                 int Cu, int Cv, int act, const float* v_act_src, int v_act, float* colsum_part, int* part_rows,
                 const float* x_amax, const float* w_amax, float* y_amax, hipStream_t s);
// MVK_IMGCONV=0 disables the kernels, MVK_IMGCONV=<n> sets the smallest batch that takes them (default 256 images);
// mvk_debug_set_flags: bit 0x100 disables them, bit 0x200 takes them for every batch size (tests, A/B probes)
int imgconv_wgrad(const float* U, const float* V, float* slab, long long slab_floats, int n, int h, int w, int Cu, int Cv,
                  int* nz, const float* u_amax, const float* v_amax, hipStream_t s);
// Register-stationary 3x3 convolutions (conv3rs.hip); 1: shape not covered.  mvk_debug_set_flags: bit 0x400 disables them,
// bit 0x800 takes them for every problem size (tests); MVK_C3RS=0 / MVK_C3RS=<min tiles> under MVK_TUNE=1.
int c3rs_conv(const float* X, const float* Wp, const float* bias, float* Y, int n, int H, int W, int Cin, int Cout, int act,
              const float* act_src, int src_act, const float* res, float res_alpha, float* colsum_part, int* part_rows,
              int x_act, float pre_scale, const float* x_amax, const float* w_amax, float* y_amax, hipStream_t s,
              float* y_pre = nullptr, int x_channels = 0, int w_channels = 0, int res_pre = 0);
bool c3rs_shape_ok(int n, int H, int W, int Cin, int Cout, int np);
bool c3rs_wgrad_ok(int n, int H, int W, int Cin, int Cout);
int c3rs_wgrad(const float* X, const float* dY, float* slab, long long slab_floats, float* dbpart, int x_act, float dy_scale,
               int n, int H, int W, int Cin, int Cout, int* nz, const float* x_amax, const float* y_amax, hipStream_t s);
int convref_reduce(const float* slab, int nz, int Cu, int Cv, int taps, float* dWref, hipStream_t s, bool deferred);
static bool c3rs_wgrad_covers(int n, int H, int W, int Cin, int Cout);
static bool c3rs_covers(int n, int H, int W, int Cin, int Cout, int np = 3) {
  static int min_tiles = -1;
  if (min_tiles < 0) {
    const char* e = mvk_tune("MVK_C3RS");
    min_tiles = e ? atoi(e) : 200;  // ~1 tile per worker: 15 us of start-up beat the tiled engine on a 50-us problem
    if (min_tiles == 0) min_tiles = 1 << 30;
  }
  if (g_dbg_flags & 0x400) return false;
  if (!c3rs_shape_ok(n, H, W, Cin, Cout, np)) return false;
  const long long tiles = ((long long)n * (H + 1) * (W + 1) + 31) / 32;
  return (g_dbg_flags & 0x800) ? true : tiles >= min_tiles;
}
static bool c3rs_wgrad_covers(int n, int H, int W, int Cin, int Cout) {
  if ((g_dbg_flags & 0x400) || !c3rs_wgrad_ok(n, H, W, Cin, Cout)) return false;
  if (g_dbg_flags & 0x800) return true;
  return c3rs_covers(n, H, W, 64, 64);  // the same size threshold as the forward kernels
}
static int imgconv_min_images() {
  static int v = -1;
  if (v < 0) {
    const char* e = mvk_tune("MVK_IMGCONV");
    v = e ? atoi(e) : 256;
    if (v == 0) v = 1 << 30;
  }
  if (g_dbg_flags & 0x100) return 1 << 30;
  if (g_dbg_flags & 0x200) return 1;
  return v;
}
static bool imgconv_act_ok(int a) { return a == MVK_ACT_NONE || a == MVK_ACT_RELU; }
}  // namespace mvk

extern "C" {

// tall-skinny / K-serial shapes: too few output tiles to fill 256 CUs -> split the reduction
static int launch_auto(GemmDesc& d, float* ws, int64_t ws_floats, hipStream_t s) {
  long long tiles = (long long)((d.M + 127) / 128) * ((d.N + 63) / 64);
  if (d.e.atomic || (ws && tiles < 128 && d.K >= 256)) return launch_splitk(d, ws, ws_floats, 512, s);
  return launch_igemm(d, 1, s);
}

int mvk_linear_fwd(const float* X, const float* W, const float* b, float* Y, int M, int N, int K, int act,
                   float* ws, int64_t ws_floats, void* stream) {
  if (M == 0) return MVK_OK;  // empty batch: nothing to launch (torch hands out NULL for empty tensors)
  if (!X || !W || !Y || M < 0 || N <= 0 || K <= 0) return MVK_EINVAL;
  if (K <= 32) {
    const int rc = smallk_fwd(X, W, 1, K, b, N, act, Y, M, N, K, mvk_stream(stream));
    if (rc != 1) return rc;
  }
  // a few hundred rows (the encoders' hidden layers at the training batch): too few 128-row tiles for the tiled engine,
  // which then splits K and needs a second launch to reduce (47 us for 512 x 784 -> 400); one 16 x 16 tile per workgroup
  static const int fewrows = mvk_tune("MVK_FEWROWS") ? atoi(mvk_tune("MVK_FEWROWS")) : 1024;
  // (with scratch the kernel splits long reductions over workgroups: 32 rows x 12544 features, the heads of the PolyMNIST ResNet
  // encoders, 57 -> ~10 us; the fat heads of the 64x64 ResNet encoder — 128 x 65536 -> 64, 4 M weights — measured SLOWER there
  // than on the tiled engine, cfg5 8.40 vs 8.34 ms, so the size limit stays)
  if (M <= fewrows && K >= 64 && (long long)N * K <= (1 << 21)) {
    const int rc = heads_launch(X, W, b, Y, nullptr, nullptr, nullptr, M, N, K, 1, K, act, mvk_stream(stream), ws, ws_floats);
    if (rc != 1) return rc;
  }
  GemmDesc d{};
  plain_a(d.a, X, K, 1, M, K);
  plain_b(d.b, W, 1, K, K, N);  // B[k][n] = W[n][k]
  rowmajor_epi(d.e, Y, N);
  d.e.bias = b;
  d.e.bias_mod = N;
  d.e.act = act;
  d.M = M;
  d.N = N;
  d.K = K;
  return launch_auto(d, ws, ws_floats, mvk_stream(stream));
}

int mvk_linear_bwd_data(const float* dY, const float* W, float* dX, int M, int N, int K, const float* y_out,
                        int y_act, const float* prev_out, int prev_act, int accumulate, float* colsum_acc, float* ws,
                        int64_t ws_floats, void* stream) {
  if (M == 0) return MVK_OK;  // empty batch: nothing to launch (torch hands out NULL for empty tensors)
  if (!dY || !W || !dX || M < 0 || N <= 0 || K <= 0 || (colsum_acc && accumulate)) return MVK_EINVAL;
  GemmDesc d{};
  plain_a(d.a, dY, N, 1, M, N);  // reduce over n
  d.a.act_src = y_out;
  d.a.act = y_act;
  plain_b(d.b, W, K, 1, N, K);  // B[kk=n][j=k] = W[n][k]
  rowmajor_epi(d.e, dX, K);
  d.e.act_src = prev_out;
  d.e.src_act = prev_act;
  d.e.atomic = accumulate;
  d.M = M;
  d.N = K;
  d.K = N;
  hipStream_t s = mvk_stream(stream);
  static const int smallk_bwd = mvk_tune("MVK_SMALLK_BWD") ? atoi(mvk_tune("MVK_SMALLK_BWD")) : 0;  // measured +10 us per step on the heads' backward-data: off
  if (smallk_bwd && N <= 32 && !y_out && !colsum_acc) {  // backward-data out of a narrow layer (the encoder heads)
    const int rc = smallk_fwd(dY, W, K, 1, nullptr, 1, MVK_ACT_NONE, dX, M, K, N, s, prev_out, prev_act, accumulate);
    if (rc != 1) return rc;
  }
  if (!colsum_acc) return launch_auto(d, ws, ws_floats, s);
  // bias gradient of the previous layer = column sums of dX: fused into the epilogue unless the launch splits K
  const long long tiles = (long long)((d.M + 127) / 128) * ((d.N + 63) / 64);
  if (!(ws && tiles < 128 && d.K >= 256)) return launch_with_colsum(d, 1, colsum_acc, ws, ws_floats, dX, M, s);
  const int rc = launch_auto(d, ws, ws_floats, s);
  return rc != MVK_OK ? rc : colsum(dX, nullptr, 0, M, K, colsum_acc, ws, ws_floats, s);
}

int mvk_linear_bwd_weight(const float* dY, const float* X, float* dW, float* db, int M, int N, int K,
                          const float* y_out, int y_act, float* ws, int64_t ws_floats, void* stream) {
  if (!dY || !X || !dW || M < 0 || N <= 0 || K <= 0) return MVK_EINVAL;
  hipStream_t s = mvk_stream(stream);
  GemmDesc d{};
  // dW[n][k] += sum_m dY[m][n] X[m][k]:  A'[i=n][kk=m] = dY[m*N + n]
  plain_a(d.a, dY, 1, N, N, M);
  d.a.act_src = y_out;
  d.a.act = y_act;
  plain_b(d.b, X, K, 1, M, K);
  rowmajor_epi(d.e, dW, K);
  d.e.atomic = 1;
  d.M = N;
  d.N = K;
  d.K = M;
  int rc = launch_splitk(d, ws, ws_floats, SPLITK_LIN, s);
  if (rc) return rc;
  if (db) return colsum(dY, y_out, y_act, M, N, db, ws, ws_floats, s);
  return MVK_OK;
}

int mvk_act_bwd_colsum(const float* dY, const float* Y, int act, int M, int N, float* dPre, float* db, float* ws,
                       int64_t ws_floats, void* stream) {
  if (!dY || !Y || !dPre || M < 0 || N <= 0) return MVK_EINVAL;
  if (M == 0) return MVK_OK;
  hipStream_t s = mvk_stream(stream);
  int rpb = (M + 1023) / 1024;  // ~1024 workgroups
  if (rpb < 4) rpb = 4;
  const int blocks = (M + rpb - 1) / rpb;
  const bool fused = (N % 4 == 0) && mvk_aligned16(dY) && mvk_aligned16(Y) && mvk_aligned16(dPre) && ws &&
                     mvk_aligned16(ws) && (!db || (long long)blocks * N <= ws_floats);
  if (!fused) {  // generic route: elementwise pass, then the column-sum kernel
    if (dPre != dY && hipMemcpyAsync(dPre, dY, sizeof(float) * (size_t)M * N, hipMemcpyDeviceToDevice, s) != hipSuccess)
      return MVK_ELAUNCH;
    const long long n = (long long)M * N;
    hipLaunchKernelGGL(act_bwd_plain_kernel, dim3((unsigned)((n + 255) / 256 > 65535 ? 65535 : (n + 255) / 256)), dim3(256),
                       0, s, dPre, Y, n, act);
    MVK_CHECK_LAUNCH();
    return db ? colsum(dPre, nullptr, 0, M, N, db, ws, ws_floats, s) : MVK_OK;
  }
  float* dpart = db ? defer_scratch(db, (long long)blocks * N, s) : nullptr;
  hipLaunchKernelGGL(act_bwd_colsum_kernel, dim3(blocks), dim3(256), 0, s, dY, Y, act, M, N, rpb, dPre, dpart ? dpart : ws);
  MVK_CHECK_LAUNCH();
  if (dpart) return defer_push_plain(db, dpart, N, blocks, N, s);
  return db ? colsum_finish(ws, blocks, N, db, s) : MVK_OK;
}

int mvk_colsum_acc(const float* dY, const float* y_out, int y_act, float* db, int M, int N, float* ws, int64_t ws_floats,
                   void* stream) {
  if (!dY || !db) return MVK_EINVAL;
  return colsum(dY, y_out, y_act, M, N, db, ws, ws_floats, mvk_stream(stream));
}

int mvk_nchw_channel_sum_acc(const float* dY, const float* y_out, int y_act, float* db, int n, int c, int hw,
                             float* ws, int64_t ws_floats, void* stream) {
  if (!dY || !db || n < 0 || c <= 0 || hw <= 0) return MVK_EINVAL;
  if (n == 0) return MVK_OK;
  int ipb = 16;
  const int gy = (n + ipb - 1) / ipb;
  float* dpart = defer_scratch(db, (long long)gy * c, mvk_stream(stream));
  float* part = dpart ? dpart : ((ws && (long long)gy * c <= ws_floats) ? ws : nullptr);
  hipLaunchKernelGGL(nchw_channel_sum_kernel, dim3(c, gy), dim3(256), 0, mvk_stream(stream), dY, y_out, y_act, n, c, hw,
                     ipb, db, part);
  MVK_CHECK_LAUNCH();
  if (dpart) return defer_push_plain(db, dpart, c, gy, c, mvk_stream(stream));
  return part ? colsum_finish_any(part, gy, c, db, mvk_stream(stream)) : MVK_OK;
}

int mvk_gemm(const float* A, const float* B, float* C, int M, int N, int K, int ta, int tb, const float* bias,
             int bias_mod, int act, int accumulate, const float* a_act_src, int a_act, const float* c_act_src,
             int c_act, float* ws, int64_t ws_floats, void* stream) {
  if (!A || !B || !C || M < 0 || N <= 0 || K <= 0) return MVK_EINVAL;
  static const int smallk_bwd = mvk_tune("MVK_SMALLK_BWD") ? atoi(mvk_tune("MVK_SMALLK_BWD")) : 0;  // measured +10 us per step on the heads' backward-data: off
  if (K <= 32 && !ta && !a_act_src && M > 0 && (smallk_bwd || (!accumulate && !c_act_src))) {
    const int rc = smallk_fwd(A, B, tb ? 1 : N, tb ? K : 1, bias, bias_mod, act, C, M, N, K, mvk_stream(stream), c_act_src,
                              c_act, accumulate);
    if (rc != 1) return rc;
  }
  GemmDesc d{};
  if (ta)
    plain_a(d.a, A, 1, M, M, K);
  else
    plain_a(d.a, A, K, 1, M, K);
  d.a.act_src = a_act_src;
  d.a.act = a_act;
  if (tb)
    plain_b(d.b, B, 1, K, K, N);
  else
    plain_b(d.b, B, N, 1, K, N);
  rowmajor_epi(d.e, C, N);
  d.e.bias = bias;
  d.e.bias_mod = bias_mod > 0 ? bias_mod : 1;
  d.e.act = act;
  d.e.act_src = c_act_src;
  d.e.src_act = c_act;
  d.e.atomic = accumulate;
  d.M = M;
  d.N = N;
  d.K = K;
  return launch_auto(d, ws, ws_floats, mvk_stream(stream));
}

// C = act(A B + bias) for a short reduction (K <= 32: the first layer of a decoder) with the published maximum of the result
// (the amax protocol of mvk_conv3x3_s): y_amax receives max |C| by atomic max and must hold 0 before the launch.
// B(k, n) = B[k * N + n] (tb = 0) or B[n * K + k] (tb = 1).  MVK_EINVAL for shapes the short-reduction kernel does not take.
int mvk_gemm_smallk_amax(const float* A, const float* B, float* C, int M, int N, int K, int tb, const float* bias, int bias_mod,
                         int act, float* y_amax, void* stream) {
  if (!A || !B || !C || !y_amax || M < 0 || N <= 0 || K <= 0 || K > 32) return MVK_EINVAL;
  if (M == 0) return MVK_OK;
  const int rc = smallk_fwd(A, B, tb ? 1 : N, tb ? K : 1, bias, bias_mod, act, C, M, N, K, mvk_stream(stream), nullptr, 0, 0, y_amax);
  return rc == 1 ? MVK_EINVAL : rc;
}

// ---------------------------------------------------------------------------------------------------------
// 4x4 / stride 2 / pad 1 pair
// ---------------------------------------------------------------------------------------------------------
static bool imgconv_pair(int h, int w, int Cu, int Cv) { return h == w && ((h == 8 && Cu == 32 && Cv == 64) || (h == 4 && Cu == 64 && Cv == 128)); }

static int conv4s2_down_impl(const float* U, const float* Wdown, const float* bias, float* V, int n, int h, int w, int Cu,
                             int Cv, int act, int u_nchw, const float* u_act_src, int u_act, const float* v_act_src,
                             int v_act, float* colsum_acc, float* ws, int64_t ws_floats, int fmt, const void* wfrag,
                             const float* x_amax, const float* w_amax, float* y_amax, void* stream) {
  if (n == 0) return MVK_OK;  // empty batch: nothing to launch (torch hands out NULL for empty tensors)
  if (!U || !Wdown || !V || n < 0 || h <= 0 || w <= 0 || Cu <= 0 || Cv <= 0 || (!x_amax != !w_amax)) return MVK_EINVAL;
  const bool wants_rs = x_amax || y_amax;  // forms only the register-stationary kernels take
  if ((fmt & ~MVK_FMT_IN_BF3) || ((fmt & MVK_FMT_IN_BF3) && (u_nchw || u_act_src))) return MVK_EINVAL;
  if (u_nchw && !u_act_src && !v_act_src && !colsum_acc) {  // the network-input layer
    if (mvk_conv4s2_small_up_supported(h, w, Cu, Cv))
      return mvk_conv4s2_small_down_fwd(U, Wdown, bias, V, n, h, w, Cu, Cv, act, stream);  // LDS-staged image, MFMA
    if (smallcin_supported(Cu, Cv)) return smallcin_fwd(U, Wdown, bias, V, n, h, w, Cu, Cv, act, mvk_stream(stream));
  }
  if (!u_nchw && !u_act_src && fmt == 0 && n >= imgconv_min_images() && imgconv_act_ok(act) &&
      imgconv_act_ok(v_act) && (!colsum_acc || (ws && ws_floats >= 256 * (int64_t)Cv)) && mvk_aligned16(U)) {
    int rows = 0;
    float* dpart = colsum_acc ? defer_scratch(colsum_acc, 256 * (long long)Cv, mvk_stream(stream)) : nullptr;
    float* cpart = dpart ? dpart : ws;
    const int rc = imgconv_down(U, Wdown, wfrag, bias, V, n, h, w, Cu, Cv, act, v_act_src, v_act, colsum_acc ? cpart : nullptr,
                                &rows, x_amax, w_amax, y_amax, mvk_stream(stream));
    if (rc == MVK_OK && dpart) return defer_push_plain(colsum_acc, dpart, Cv, rows, Cv, mvk_stream(stream));
    if (rc == MVK_OK && colsum_acc) return colsum_finish(ws, rows, Cv, colsum_acc, mvk_stream(stream));
    if (rc != 1) return rc;
  }
  if (wants_rs) return MVK_EINVAL;  // ask mvk_conv4s2_scaled_ok first
  GemmDesc d{};
  d.a = AOperand{};
  d.a.p = U;
  d.a.kind = u_nchw ? A_DOWN_NCHW : A_DOWN;
  d.a.trans = 0;
  d.a.C = Cu;
  d.a.H = 2 * h;
  d.a.W = 2 * w;
  d.a.OH = h;
  d.a.OW = w;
  d.a.contig_k = u_nchw ? 0 : 1;
  d.a.vec4 = (!u_nchw) && (Cu % 4 == 0) && mvk_aligned16(U) && (!u_act_src || mvk_aligned16(u_act_src));
  d.a.act_src = u_act_src;
  d.a.act = u_act;
  d.a.bf3 = (fmt & MVK_FMT_IN_BF3) ? 1 : 0;
  d.a.plane_bytes = (long long)n * 4 * h * w * Cu * 2;
  plain_b(d.b, Wdown, Cv, 1, 16 * Cu, Cv);
  rowmajor_epi(d.e, V, Cv);
  d.e.bias = bias;
  d.e.bias_mod = Cv;
  d.e.act = act;
  d.e.act_src = v_act_src;
  d.e.src_act = v_act;
  d.M = n * h * w;
  d.N = Cv;
  d.K = 16 * Cu;
  return launch_with_colsum(d, 1, colsum_acc, ws, ws_floats, V, (long long)n * h * w, mvk_stream(stream));
}

int mvk_conv4s2_down(const float* U, const float* Wdown, const float* bias, float* V, int n, int h, int w, int Cu,
                     int Cv, int act, int u_nchw, const float* u_act_src, int u_act, const float* v_act_src,
                     int v_act, float* colsum_acc, float* ws, int64_t ws_floats, int fmt, const void* wfrag, void* stream) {
  return conv4s2_down_impl(U, Wdown, bias, V, n, h, w, Cu, Cv, act, u_nchw, u_act_src, u_act, v_act_src, v_act, colsum_acc, ws,
                           ws_floats, fmt, wfrag, nullptr, nullptr, nullptr, stream);
}

// 1 when mvk_conv4s2_down_s / mvk_conv4s2_up_s take this layer at this batch (the register-stationary kernels cover it)
int mvk_conv4s2_scaled_ok(int n, int h, int w, int Cu, int Cv) {
  return n >= imgconv_min_images() && imgconv_pair(h, w, Cu, Cv) && (h != 4 || n % 2 == 0) ? 1 : 0;
}

// mvk_conv4s2_down on the register-stationary kernels with the amax protocol of mvk_conv3x3_s: x_amax + w_amax (both or
// neither) select the scaled-fp16 form (3 MFMAs per product; the weights are converted in the kernel from the fp32 pack),
// y_amax (optional) receives max |V|.  NHWC input, no fused input activation; MVK_EINVAL where mvk_conv4s2_scaled_ok says 0.
int mvk_conv4s2_down_s(const float* U, const float* Wdown, const float* bias, float* V, int n, int h, int w, int Cu, int Cv,
                       int act, const float* v_act_src, int v_act, float* colsum_acc, const float* x_amax, const float* w_amax,
                       float* y_amax, float* ws, int64_t ws_floats, const void* wfrag, void* stream) {
  if (!mvk_conv4s2_scaled_ok(n, h, w, Cu, Cv) || (!x_amax && !y_amax)) return MVK_EINVAL;
  return conv4s2_down_impl(U, Wdown, bias, V, n, h, w, Cu, Cv, act, 0, nullptr, MVK_ACT_NONE, v_act_src, v_act, colsum_acc, ws,
                           ws_floats, 0, wfrag, x_amax, w_amax, y_amax, stream);
}

static int conv4s2_up_impl(const float* V, const float* Wup, const float* bias, float* U, int n, int h, int w, int Cu,
                           int Cv, int act, int u_nchw, const float* u_act_src, int u_act, float* colsum_acc, float* ws,
                           int64_t ws_floats, int fmt, const void* wfrag, const float* x_amax, const float* w_amax,
                           float* y_amax, void* stream) {
  if (n == 0) return MVK_OK;  // empty batch: nothing to launch (torch hands out NULL for empty tensors)
  if (!V || !Wup || !U || n < 0 || h <= 0 || w <= 0 || Cu <= 0 || Cv <= 0 || (colsum_acc && u_nchw) || (!x_amax != !w_amax))
    return MVK_EINVAL;
  const bool wants_rs = x_amax || y_amax;
  if (fmt & ~(MVK_FMT_IN_BF3 | MVK_FMT_TILED)) return MVK_EINVAL;
  if (!u_nchw && fmt == 0 && n >= imgconv_min_images() && imgconv_act_ok(act) && imgconv_act_ok(u_act) &&
      (!colsum_acc || (ws && ws_floats >= 256 * (int64_t)Cu)) && mvk_aligned16(V)) {
    int rows = 0;
    float* dpart = colsum_acc ? defer_scratch(colsum_acc, 256 * (long long)Cu, mvk_stream(stream)) : nullptr;
    float* cpart = dpart ? dpart : ws;
    const int rc = imgconv_up(V, Wup, wfrag, bias, U, n, h, w, Cu, Cv, act, u_act_src, u_act, colsum_acc ? cpart : nullptr,
                              &rows, x_amax, w_amax, y_amax, mvk_stream(stream));
    if (rc == MVK_OK && dpart) return defer_push_plain(colsum_acc, dpart, Cu, rows, Cu, mvk_stream(stream));
    if (rc == MVK_OK && colsum_acc) return colsum_finish(ws, rows, Cu, colsum_acc, mvk_stream(stream));
    if (rc != 1) return rc;
  }
  if (wants_rs) return MVK_EINVAL;
  GemmDesc d{};
  d.a = AOperand{};
  d.a.p = V;
  d.a.kind = A_UP;
  d.a.trans = 0;
  d.a.C = Cv;
  d.a.H = h;
  d.a.W = w;
  d.a.OH = h;
  d.a.OW = w;
  d.a.contig_k = 1;
  d.a.vec4 = (Cv % 4 == 0) && mvk_aligned16(V);
  d.a.bf3 = (fmt & MVK_FMT_IN_BF3) ? 1 : 0;
  d.a.plane_bytes = (long long)n * h * w * Cv * 2;
  plain_b(d.b, Wup, Cu, 1, 4 * Cv, Cu);
  d.b.z_stride = (long long)4 * Cv * Cu;
  d.e = Epilogue{};
  d.e.out = U;
  d.e.kind = u_nchw ? E_UP_NCHW : E_UP;
  d.e.ld = Cu;
  d.e.bias = bias;
  d.e.bias_mod = Cu;
  d.e.act = act;
  d.e.act_src = u_act_src;
  d.e.src_act = u_act;
  d.e.Cu = Cu;
  d.e.OH = h;
  d.e.OW = w;
  d.M = n * h * w;
  d.N = Cu;
  d.K = 4 * Cv;
  d.zmode = Z_PARITY;
  return launch_with_colsum(d, 4, colsum_acc, ws, ws_floats, U, (long long)n * 4 * h * w, mvk_stream(stream));
}

int mvk_conv4s2_up(const float* V, const float* Wup, const float* bias, float* U, int n, int h, int w, int Cu,
                   int Cv, int act, int u_nchw, const float* u_act_src, int u_act, float* colsum_acc, float* ws,
                   int64_t ws_floats, int fmt, const void* wfrag, void* stream) {
  return conv4s2_up_impl(V, Wup, bias, U, n, h, w, Cu, Cv, act, u_nchw, u_act_src, u_act, colsum_acc, ws, ws_floats, fmt, wfrag,
                         nullptr, nullptr, nullptr, stream);
}

// mvk_conv4s2_up with the amax protocol (see mvk_conv4s2_down_s); NHWC output
int mvk_conv4s2_up_s(const float* V, const float* Wup, const float* bias, float* U, int n, int h, int w, int Cu, int Cv, int act,
                     const float* u_act_src, int u_act, float* colsum_acc, const float* x_amax, const float* w_amax,
                     float* y_amax, float* ws, int64_t ws_floats, const void* wfrag, void* stream) {
  if (!mvk_conv4s2_scaled_ok(n, h, w, Cu, Cv) || (!x_amax && !y_amax)) return MVK_EINVAL;
  return conv4s2_up_impl(V, Wup, bias, U, n, h, w, Cu, Cv, act, 0, u_act_src, u_act, colsum_acc, ws, ws_floats, 0, wfrag, x_amax,
                         w_amax, y_amax, stream);
}

// ---- 3x3 / stride 1 / pad 1 convolution on NHWC activations (ResNet blocks) --------------------------------------
// Y[n,H,W,Cout] = act(conv3x3(X[n,H,W,Cin]) + b) (* src_act'(y_act_src));  Wp[(kh*3+kw)*Cin + ci][co].
// The same launch is the backward-data pass when it is fed the output gradient and the flipped / transposed pack.
static int conv3x3_any(const float* X, const float* Wp, const float* bias, float* Y, int n, int H, int W, int Cin, int Cout,
                       int act, const float* y_act_src, int y_src_act, float* colsum_acc, const float* res, float res_alpha,
                       float* ws, int64_t ws_floats, void* stream, int x_act = MVK_ACT_NONE, float pre_scale = 1.f,
                       const float* x_amax = nullptr, const float* w_amax = nullptr, float* y_amax = nullptr,
                       float* y_pre = nullptr) {
  // forms only the register-stationary kernels take
  // (y_amax alone with an image on the input side: the direct kernel of conv3small.hip publishes it, mvk_conv3x3_y)
  const bool img_y = y_amax && !x_amax && !w_amax && Cin <= 4 && !res && x_act == MVK_ACT_NONE && pre_scale == 1.f;
  const bool fused = !img_y && (x_act != MVK_ACT_NONE || pre_scale != 1.f || x_amax || w_amax || y_amax || y_pre);
  if (y_pre && !(res && x_amax && w_amax && !y_act_src && !colsum_acc)) return MVK_EINVAL;  // the second store: scaled residual form
  const int np = x_amax && w_amax ? 2 : 3;
  if (!X || !Wp || !Y || n < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (!x_amax != !w_amax)) return MVK_EINVAL;
  if (fused && !(n > 0 && ws && ws_floats >= 256ll * Cout && !(res && colsum_acc) && mvk_aligned16(X) &&
                 (!colsum_acc || y_act_src) && c3rs_covers(n, H, W, Cin, Cout, np)))
    return MVK_EINVAL;  // ask mvk_conv3x3_fused_ok / mvk_conv3x3_scaled_ok first
  if (n > 0 && Cin <= 4 && !res) {  // the image-consuming layer (or the backward-data pass of the image-producing one)
    const int rc = conv3_smallcin(X, Wp, bias, Y, n, H, W, Cin, Cout, act, y_act_src, y_src_act, mvk_stream(stream), y_amax);
    if (rc == MVK_OK && colsum_acc)
      return colsum(Y, nullptr, 0, n * H * W, Cout, colsum_acc, ws, ws_floats, mvk_stream(stream));
    if (rc != 1) return rc;
  }
  if (img_y) return MVK_EINVAL;  // mvk_conv3x3_y: only the direct image kernel publishes without the scaled operands
  if (n > 0 && Cout <= 4 && !res) {  // the image-producing layer
    const int rc = conv3_smallcout(X, Wp, bias, Y, n, H, W, Cin, Cout, act, y_act_src, y_src_act, mvk_stream(stream));
    if (rc == MVK_OK && colsum_acc)
      return colsum(Y, nullptr, 0, n * H * W, Cout, colsum_acc, ws, ws_floats, mvk_stream(stream));
    if (rc != 1) return rc;
  }
  if (n > 0 && ws && ws_floats >= 256ll * Cout && !(res && colsum_acc) && mvk_aligned16(X) && act != MVK_ACT_SIGMOID &&
      !(y_act_src && y_src_act == MVK_ACT_SIGMOID) && (!colsum_acc || y_act_src) && c3rs_covers(n, H, W, Cin, Cout, np)) {
    int rows = 0;
    float* dpart = colsum_acc ? defer_scratch(colsum_acc, 256ll * Cout, mvk_stream(stream)) : nullptr;
    const int rc = c3rs_conv(X, Wp, bias, Y, n, H, W, Cin, Cout, act, y_act_src, y_src_act, res, res_alpha,
                             colsum_acc ? (dpart ? dpart : ws) : nullptr, &rows, x_act, pre_scale, x_amax, w_amax, y_amax,
                             mvk_stream(stream), y_pre);
    if (rc == MVK_OK && dpart) return defer_push_plain(colsum_acc, dpart, Cout, rows, Cout, mvk_stream(stream));
    if (rc == MVK_OK && colsum_acc) return colsum_finish(ws, rows, Cout, colsum_acc, mvk_stream(stream));
    if (rc != 1) return rc;
    if (dpart) return MVK_EINVAL;  // covered shapes never decline after taking arena space
  }
  GemmDesc d{};
  d.a = AOperand{};
  d.a.p = X;
  d.a.kind = A_DOWN;
  d.a.tw = 3;
  d.a.mul = 1;
  d.a.trans = 0;
  d.a.C = Cin;
  d.a.H = H;
  d.a.W = W;
  d.a.OH = H;
  d.a.OW = W;
  d.a.contig_k = 1;
  d.a.vec4 = (Cin % 4 == 0) && mvk_aligned16(X);
  plain_b(d.b, Wp, Cout, 1, 9 * Cin, Cout);
  rowmajor_epi(d.e, Y, Cout);
  d.e.bias = bias;
  d.e.bias_mod = Cout;
  d.e.act = act;
  d.e.act_src = y_act_src;
  d.e.src_act = y_src_act;
  d.e.res = res;
  d.e.res_alpha = res_alpha;
  d.M = n * H * W;
  d.N = Cout;
  d.K = 9 * Cin;
  if (res && colsum_acc && !epilogue_vec_ok(d.e, d.N)) return MVK_EINVAL;  // the sums would include the residual
  return launch_with_colsum(d, 1, colsum_acc, ws, ws_floats, Y, (long long)n * H * W, mvk_stream(stream));
}

int mvk_conv3x3(const float* X, const float* Wp, const float* bias, float* Y, int n, int H, int W, int Cin, int Cout,
                int act, const float* y_act_src, int y_src_act, float* colsum_acc, float* ws, int64_t ws_floats,
                void* stream) {
  return conv3x3_any(X, Wp, bias, Y, n, H, W, Cin, Cout, act, y_act_src, y_src_act, colsum_acc, nullptr, 0.f, ws, ws_floats,
                     stream);
}

// Y = res + res_alpha * (act(conv3x3(X) + b) * src_act'(y_act_src)): the residual sum of a ResNet block in the
// convolution's epilogue (res may alias nothing the launch reads through X)
int mvk_conv3x3_res(const float* X, const float* Wp, const float* bias, float* Y, int n, int H, int W, int Cin, int Cout,
                    int act, const float* y_act_src, int y_src_act, const float* res, float res_alpha, float* ws,
                    int64_t ws_floats, void* stream) {
  if (!res) return MVK_EINVAL;
  return conv3x3_any(X, Wp, bias, Y, n, H, W, Cin, Cout, act, y_act_src, y_src_act, nullptr, res, res_alpha, ws, ws_floats,
                     stream);
}

// 1 when the fused forms mvk_conv3x3_f / mvk_conv3x3_wgrad_f take this problem (the register-stationary kernels cover both)
int mvk_conv3x3_fused_ok(int n, int H, int W, int Cin, int Cout) {
  return n > 0 && c3rs_covers(n, H, W, Cin, Cout) && c3rs_wgrad_covers(n, H, W, Cin, Cout) ? 1 : 0;
}

// Y = [res + res_alpha *] (act(pre_scale * conv(x_act(X)) + bias) * src_act'(y_act_src)); colsum_acc as in mvk_conv3x3
int mvk_conv3x3_f(const float* X, const float* Wp, const float* bias, float* Y, int n, int H, int W, int Cin, int Cout, int act,
                  const float* y_act_src, int y_src_act, const float* res, float res_alpha, float* colsum_acc, int x_act,
                  float pre_scale, float* ws, int64_t ws_floats, void* stream) {
  return conv3x3_any(X, Wp, bias, Y, n, H, W, Cin, Cout, act, y_act_src, y_src_act, colsum_acc, res, res_alpha, ws, ws_floats,
                     stream, x_act, pre_scale);
}

// mvk_conv3x3 for an IMAGE on the input side (Cin <= 4: conv_img of the ResNet encoders, the backward-data pass of the decoders'
// conv_img) that also publishes max |Y| (amax protocol): the stack behind it takes the scaled-fp16 form without a pass over Y.
int mvk_conv3x3_y(const float* X, const float* Wp, const float* bias, float* Y, int n, int H, int W, int Cin, int Cout, int act,
                  const float* y_act_src, int y_src_act, float* colsum_acc, float* y_amax, float* ws, int64_t ws_floats,
                  void* stream) {
  if (!y_amax || Cin > 4) return MVK_EINVAL;
  return conv3x3_any(X, Wp, bias, Y, n, H, W, Cin, Cout, act, y_act_src, y_src_act, colsum_acc, nullptr, 0.f, ws, ws_floats, stream,
                     MVK_ACT_NONE, 1.f, nullptr, nullptr, y_amax);
}

// The scaled-fp16 form of mvk_conv3x3_f (3 MFMAs per product instead of 6, csrc/bf3.hpp): x_amax / w_amax = device scalars
// holding upper bounds of max |X| and max |Wp| (mvk_amax, the `amax` slot of mvk_pack_weights, or the y_amax of the launch
// that produced X); y_amax (optional) receives max |Y| by atomic max and must hold 0 before the launch.
int mvk_conv3x3_scaled_ok(int n, int H, int W, int Cin, int Cout) { return n > 0 && c3rs_covers(n, H, W, Cin, Cout, 2) ? 1 : 0; }
int mvk_conv3x3_s(const float* X, const float* Wp, const float* bias, float* Y, int n, int H, int W, int Cin, int Cout, int act,
                  const float* y_act_src, int y_src_act, const float* res, float res_alpha, float* colsum_acc, int x_act,
                  float pre_scale, const float* x_amax, const float* w_amax, float* y_amax, float* ws, int64_t ws_floats,
                  void* stream) {
  if (!x_amax || !w_amax) return MVK_EINVAL;
  return conv3x3_any(X, Wp, bias, Y, n, H, W, Cin, Cout, act, y_act_src, y_src_act, colsum_acc, res, res_alpha, ws, ws_floats,
                     stream, x_act, pre_scale, x_amax, w_amax, y_amax);
}

/* mvk_conv3x3_s in its residual form with a SECOND store: y_pre <- act(conv + bias) (what res + res_alpha * (.) is formed from).
 * A post-activation ResNet block (models/nn/mmnist.py:229-246: x_s + 0.1 * lrelu(conv2(.))) keeps that tensor for its backward;
 * one launch instead of a convolution and an elementwise pass over three tensors. */
int mvk_conv3x3_s2(const float* X, const float* Wp, const float* bias, float* Y, float* y_pre, int n, int H, int W, int Cin, int Cout,
                   int act, const float* res, float res_alpha, const float* x_amax, const float* w_amax, float* y_amax, float* ws,
                   int64_t ws_floats, void* stream) {
  if (!x_amax || !w_amax || !res || !y_pre || !mvk_conv3x3_scaled_ok(n, H, W, Cin, Cout)) return MVK_EINVAL;
  return conv3x3_any(X, Wp, bias, Y, n, H, W, Cin, Cout, act, nullptr, MVK_ACT_NONE, nullptr, res, res_alpha, ws, ws_floats, stream,
                     MVK_ACT_NONE, 1.f, x_amax, w_amax, y_amax, y_pre);
}

/* One launch of mvk_conv3x3_s over a SLICE of the layer's input channels: X [n][H][W][x_channels], the slice starts at channel
 * x_off and has Cin channels; Wp is the layer's whole pack [9 x_channels][Cout].  res_pre != 0: Y = act(conv + bias + res) — `res`
 * is the partial sum the other slice's launch left (act = none, no bias there).  A 256-channel layer (the register-stationary
 * kernels hold at most 128 input channels' weights) = two such launches (models/nn/mmnist.py:345-352, cub.py:233-240). */
int mvk_conv3x3_s_part(const float* X, int x_channels, int x_off, const float* Wp, const float* bias, float* Y, int n, int H, int W,
                       int Cin, int Cout, int act, const float* y_act_src, int y_src_act, const float* res, int res_pre, int x_act,
                       float pre_scale, const float* x_amax, const float* w_amax, float* y_amax, void* stream) {
  if (!X || !Wp || !Y || !x_amax || !w_amax || n <= 0 || x_off < 0 || x_off + Cin > x_channels || x_off % 4 != 0 ||
      (res_pre && !res) || (res && !res_pre) || !mvk_conv3x3_scaled_ok(n, H, W, Cin, Cout))
    return MVK_EINVAL;
  const int rc = c3rs_conv(X + x_off, Wp + (long long)x_off * Cout, bias, Y, n, H, W, Cin, Cout, act, y_act_src, y_src_act, res, 1.f,
                           nullptr, nullptr, x_act, pre_scale, x_amax, w_amax, y_amax, mvk_stream(stream), nullptr, x_channels,
                           x_channels, res_pre);
  return rc == 1 ? MVK_EINVAL : rc;
}

static int conv3x3_wgrad_any(const float* X, const float* dY, float* dWref, float* db, int x_act, float dy_scale, int n, int H,
                             int W, int Cin, int Cout, float* ws, int64_t ws_floats, void* stream,
                             const float* x_amax = nullptr, const float* dy_amax = nullptr);

// scaled-fp16 form of mvk_conv3x3_wgrad_f (3 MFMAs per product; x_amax / dy_amax as in mvk_conv3x3_s)
int mvk_conv3x3_wgrad_scaled_ok(int n, int H, int W, int Cin, int Cout) {
  return n > 0 && c3rs_wgrad_covers(n, H, W, Cin, Cout) ? 1 : 0;
}
int mvk_conv3x3_wgrad_s(const float* X, const float* dY, float* dWref, float* db, int n, int H, int W, int Cin, int Cout,
                        int x_act, float dy_scale, const float* x_amax, const float* dy_amax, float* ws, int64_t ws_floats,
                        void* stream) {
  if (!x_amax || !dy_amax || !mvk_conv3x3_wgrad_scaled_ok(n, H, W, Cin, Cout)) return MVK_EINVAL;
  return conv3x3_wgrad_any(X, dY, dWref, db, x_act, dy_scale, n, H, W, Cin, Cout, ws, ws_floats, stream, x_amax, dy_amax);
}

// dWref[Cout][Cin][3][3] += dy_scale * sum_pos x_act(X)(gathered) dY;  db[Cout] += dy_scale * sum_pos dY (db may be null)
int mvk_conv3x3_wgrad_f(const float* X, const float* dY, float* dWref, float* db, int n, int H, int W, int Cin, int Cout,
                        int x_act, float dy_scale, float* ws, int64_t ws_floats, void* stream) {
  if (!mvk_conv3x3_fused_ok(n, H, W, Cin, Cout)) return MVK_EINVAL;
  return conv3x3_wgrad_any(X, dY, dWref, db, x_act, dy_scale, n, H, W, Cin, Cout, ws, ws_floats, stream);
}

// dWref[Cout][Cin][3][3] += sum_pos X(gathered) dY
int mvk_conv3x3_wgrad(const float* X, const float* dY, float* dWref, int n, int H, int W, int Cin, int Cout, float* ws,
                      int64_t ws_floats, void* stream) {
  return conv3x3_wgrad_any(X, dY, dWref, nullptr, MVK_ACT_NONE, 1.f, n, H, W, Cin, Cout, ws, ws_floats, stream);
}

static int conv3x3_wgrad_any(const float* X, const float* dY, float* dWref, float* db, int x_act, float dy_scale, int n, int H,
                             int W, int Cin, int Cout, float* ws, int64_t ws_floats, void* stream, const float* x_amax,
                             const float* dy_amax) {
  if (!X || !dY || !dWref || n < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return MVK_EINVAL;
  if (n > 0 && (Cin <= 4 || Cout <= 4)) {  // an image on one side: per-workgroup slabs in dWref order + ordered finish
    const long long total = 9ll * Cin * Cout;
    float* dslab = defer_scratch(dWref, 1024 * total, mvk_stream(stream));
    int nz = 0;
    const int rc = conv3_small_wgrad(X, dY, dslab ? dslab : ws, dslab ? 1024 * total : ws_floats, n, H, W, Cin, Cout, &nz,
                                     mvk_stream(stream));
    if (rc == MVK_OK)
      return dslab ? defer_push_plain(dWref, dslab, total, nz, total, mvk_stream(stream))
                   : colsum_finish_any(ws, nz, (int)total, dWref, mvk_stream(stream));
    if (rc != 1) return rc;
  }
  if (n > 0 && mvk_aligned16(X) && mvk_aligned16(dY) && c3rs_wgrad_covers(n, H, W, Cin, Cout)) {
    const int types = (Cin / 64) * (Cout / 64), workers = 256 / types;
    const long long slab_floats = (long long)workers * 9 * Cin * Cout, db_floats = db ? ((long long)workers * Cout + 3) & ~3ll : 0;
    float* dslab = defer_scratch(dWref, slab_floats, mvk_stream(stream));
    float* dbslab = db ? defer_scratch(db, db_floats, mvk_stream(stream)) : nullptr;
    const long long ws_need = (dslab ? 0 : slab_floats) + ((db && !dbslab) ? db_floats : 0);
    if (ws_need == 0 || (ws && ws_floats >= ws_need && mvk_aligned16(ws))) {
      float* wslab = dslab ? dslab : ws;
      float* bslab = !db ? nullptr : (dbslab ? dbslab : ws + (dslab ? 0 : slab_floats));
      int nz = 0;
      const int rc = c3rs_wgrad(X, dY, wslab, slab_floats, bslab, x_act, dy_scale, n, H, W, Cin, Cout, &nz, x_amax, dy_amax,
                                mvk_stream(stream));
      if (rc != MVK_OK) return rc == 1 ? MVK_EINVAL : rc;  // covered shapes never decline
      if (db) {
        const int r = dbslab ? defer_push_plain(db, dbslab, Cout, nz, Cout, mvk_stream(stream))
                             : colsum_finish_any(bslab, nz, Cout, db, mvk_stream(stream));
        if (r != MVK_OK) return r;
      }
      return convref_reduce(wslab, nz, Cin, Cout, 9, dWref, mvk_stream(stream), dslab != nullptr);
    }
    if (dslab || dbslab) return MVK_EINVAL;
  }
  if (db || x_act != MVK_ACT_NONE || dy_scale != 1.f || x_amax || dy_amax) return MVK_EINVAL;  // fused forms: register-stationary kernel only
  GemmDesc d{};
  d.a = AOperand{};
  d.a.p = X;
  d.a.kind = A_DOWN;
  d.a.tw = 3;
  d.a.mul = 1;
  d.a.trans = 1;  // GEMM row = (tap, ci), reduction index = position
  d.a.C = Cin;
  d.a.H = H;
  d.a.W = W;
  d.a.OH = H;
  d.a.OW = W;
  d.a.contig_k = 0;
  d.a.vec4 = (Cin % 4 == 0) && mvk_aligned16(X);
  plain_b(d.b, dY, Cout, 1, n * H * W, Cout);
  d.e = Epilogue{};
  d.e.out = dWref;
  d.e.kind = E_CONVREF;
  d.e.taps = 9;
  d.e.bias_mod = 1;
  d.e.atomic = 1;
  d.e.Cu = Cin;
  d.e.OH = H;
  d.e.OW = W;
  d.M = 9 * Cin;
  d.N = Cout;
  d.K = n * H * W;
  return launch_splitk(d, ws, ws_floats, 1024, mvk_stream(stream));
}

// the implicit GEMM of a 4x4 / stride-2 weight gradient: dWref[cv][cu][4][4] += sum over positions U(window)^T V
static GemmDesc conv4s2_wgrad_desc(const float* U, const float* V, float* dWref, int n, int h, int w, int Cu, int Cv, int u_nchw,
                                   const float* u_act_src, int u_act) {
  GemmDesc d{};
  d.a = AOperand{};
  d.a.p = U;
  d.a.kind = u_nchw ? A_DOWN_NCHW : A_DOWN;
  d.a.trans = 1;  // GEMM row = (tap, cu), reduction index = position
  d.a.C = Cu;
  d.a.H = 2 * h;
  d.a.W = 2 * w;
  d.a.OH = h;
  d.a.OW = w;
  d.a.contig_k = u_nchw ? 1 : 0;
  d.a.vec4 = (!u_nchw) && (Cu % 4 == 0) && mvk_aligned16(U) && (!u_act_src || mvk_aligned16(u_act_src));
  d.a.act_src = u_act_src;
  d.a.act = u_act;
  plain_b(d.b, V, Cv, 1, n * h * w, Cv);
  d.e = Epilogue{};
  d.e.out = dWref;
  d.e.kind = E_CONVREF;
  d.e.bias_mod = 1;
  d.e.atomic = 1;
  d.e.Cu = Cu;
  d.e.OH = h;
  d.e.OW = w;
  d.M = 16 * Cu;
  d.N = Cv;
  d.K = n * h * w;
  return d;
}

static int conv4s2_wgrad_impl(const float* U, const float* V, float* dWref, int n, int h, int w, int Cu, int Cv,
                              int u_nchw, const float* u_act_src, int u_act, float* ws, int64_t ws_floats,
                              const float* u_amax, const float* v_amax, void* stream) {
  if (!U || !V || !dWref || n < 0 || h <= 0 || w <= 0 || Cu <= 0 || Cv <= 0 || (!u_amax != !v_amax)) return MVK_EINVAL;
  if (u_nchw && !u_act_src && smallcin_supported(Cu, Cv)) {
    const int rc = smallcin_wgrad(U, V, dWref, n, h, w, Cu, Cv, ws, ws_floats, mvk_stream(stream));
    if (rc != 1) return rc;  // 1: scratch too small -> the implicit GEMM below
  }
  // the output-stationary kernel writes 33.5 MB of per-worker slabs whatever n is: from 1024 images (MVK_IMGWGRAD_MIN for A/B)
  static const int wg_min = mvk_tune("MVK_IMGWGRAD_MIN") ? atoi(mvk_tune("MVK_IMGWGRAD_MIN")) : 0;
  const bool wg_ok = wg_min > 0 ? (n >= wg_min && !(g_dbg_flags & 0x100)) : n / 4 >= imgconv_min_images();
  if (!u_nchw && !u_act_src && ws && wg_ok && mvk_aligned16(U) && mvk_aligned16(V) &&
      mvk_aligned16(ws)) {
    int nz = 0;
    const long long slab_floats = 256ll * 16 * Cu * Cv;
    float* dslab = defer_scratch(dWref, slab_floats, mvk_stream(stream));
    const int rc = imgconv_wgrad(U, V, dslab ? dslab : ws, dslab ? slab_floats : ws_floats, n, h, w, Cu, Cv, &nz, u_amax,
                                 v_amax, mvk_stream(stream));
    if (rc == MVK_OK) return convref_reduce(dslab ? dslab : ws, nz, Cu, Cv, 16, dWref, mvk_stream(stream), dslab != nullptr);
    if (rc != 1) return rc;
  }
  if (u_amax) return MVK_EINVAL;  // the scaled form exists in the register-stationary kernel only (mvk_conv4s2_wgrad_scaled_ok)
  GemmDesc d = conv4s2_wgrad_desc(U, V, dWref, n, h, w, Cu, Cv, u_nchw, u_act_src, u_act);
  return launch_splitk(d, ws, ws_floats, SPLITK_C4, mvk_stream(stream));
}

// Two weight gradients of 4x4 / stride-2 layers at a small batch in ONE launch (igemm_bf_pair_kernel): both must take the
// implicit-GEMM split-K path with the 128 x 64 split-bf16 tile and land in the deferred arena (their ordered finishes are queued
// like those of two separate launches: same slabs, same sums, bit-identical gradients); anything else = two launches.
int mvk_conv4s2_wgrad_pair(const float* U0, const float* V0, float* dW0, int h0, int w0, int Cu0, int Cv0, const float* U1,
                           const float* V1, float* dW1, int h1, int w1, int Cu1, int Cv1, int n, float* ws, int64_t ws_floats,
                           void* stream) {
  if (!U0 || !V0 || !dW0 || !U1 || !V1 || !dW1 || n < 0) return MVK_EINVAL;
  hipStream_t s = mvk_stream(stream);
  static const bool off = mvk_tune("MVK_WGRAD_PAIR") && atoi(mvk_tune("MVK_WGRAD_PAIR")) == 0;  // A/B: two launches
  GemmDesc d[2] = {conv4s2_wgrad_desc(U0, V0, dW0, n, h0, w0, Cu0, Cv0, 0, nullptr, MVK_ACT_NONE),
                   conv4s2_wgrad_desc(U1, V1, dW1, n, h1, w1, Cu1, Cv1, 0, nullptr, MVK_ACT_NONE)};
  int z[2] = {0, 0};
  bool ok = !off && engine() == 1 && n > 0 && !(n / 4 >= imgconv_min_images());  // (large batches: the output-stationary kernel)
  const long long lim = 1ll << 29;
  for (int i = 0; i < 2 && ok; ++i) {
    GemmDesc& g = d[i];
    ok = g.a.vec4 && g.b.vec4 && !g.b.contig_k && g.N > 32 && g.N <= 128 && g.N % 64 == 0 &&
         (long long)g.a.H * g.a.W * g.a.C * ((long long)g.K / (g.a.OH * g.a.OW) + 1) < lim && (long long)g.K * g.N < lim;
    if (!ok) break;
    const int tiles = ((g.M + 127) / 128) * ((g.N + 63) / 64), ktiles = (g.K + BK - 1) / BK;
    int splits = splitk_target(SPLITK_C4) / tiles;
    splits = splits < 1 ? 1 : (splits > ktiles ? ktiles : splits);
    g.zmode = Z_SPLITK;
    g.ksplit_tiles = (ktiles + splits - 1) / splits;
    g.ksplit_tiles += g.ksplit_tiles & 1;
    z[i] = (ktiles + g.ksplit_tiles - 1) / g.ksplit_tiles;
    ok = z[i] > 1 && (long long)tiles * z[i] >= bf_min_blocks() && defer_free(g.e.out);
  }
  if (ok) {
    float* slab0 = defer_scratch(d[0].e.out, (long long)z[0] * d[0].M * d[0].N, s);
    float* slab1 = slab0 ? defer_scratch(d[1].e.out, (long long)z[1] * d[1].M * d[1].N, s) : nullptr;
    if (slab0 && slab1) {
      d[0].e.ws = slab0;
      d[1].e.ws = slab1;
      for (int i = 0; i < 2; ++i) d[i].dbg = g_dbg, d[i].dbg_flags = g_dbg_flags;
      const unsigned gx0 = (d[0].M + 127) / 128, gy0 = (d[0].N + 63) / 64, gx1 = (d[1].M + 127) / 128, gy1 = (d[1].N + 63) / 64;
      const unsigned n0 = gx0 * gy0 * z[0], n1 = gx1 * gy1 * z[1];
      hipLaunchKernelGGL((igemm_bf_pair_kernel<128, 64, AM_COL, BM_N, false, 2>), dim3(n0 + n1), dim3(256), 0, s, d[0], d[1], n0,
                         gx0, gy0, gx1, gy1);
      MVK_CHECK_LAUNCH();
      int rc = defer_push(d[0].e, d[0].M, d[0].N, z[0], (long long)d[0].M * d[0].N);
      if (rc == MVK_OK) rc = defer_push(d[1].e, d[1].M, d[1].N, z[1], (long long)d[1].M * d[1].N);
      return rc;
    }
    if (slab0) {
      // the arena (still growing in the first steps of a process) had room for the first region only: two launches, the first
      // into the region it was granted, the second on whatever path mvk_conv4s2_wgrad takes without one
      d[0].e.ws = slab0;
      int rc = launch_igemm(d[0], z[0], s);
      if (rc == MVK_OK) rc = defer_push(d[0].e, d[0].M, d[0].N, z[0], (long long)d[0].M * d[0].N);
      if (rc == MVK_OK)
        rc = conv4s2_wgrad_impl(U1, V1, dW1, n, h1, w1, Cu1, Cv1, 0, nullptr, MVK_ACT_NONE, ws, ws_floats, nullptr, nullptr, stream);
      return rc;
    }
  }
  int rc = conv4s2_wgrad_impl(U0, V0, dW0, n, h0, w0, Cu0, Cv0, 0, nullptr, MVK_ACT_NONE, ws, ws_floats, nullptr, nullptr, stream);
  if (rc == MVK_OK)
    rc = conv4s2_wgrad_impl(U1, V1, dW1, n, h1, w1, Cu1, Cv1, 0, nullptr, MVK_ACT_NONE, ws, ws_floats, nullptr, nullptr, stream);
  return rc;
}

int mvk_conv4s2_wgrad(const float* U, const float* V, float* dWref, int n, int h, int w, int Cu, int Cv,
                      int u_nchw, const float* u_act_src, int u_act, float* ws, int64_t ws_floats, void* stream) {
  return conv4s2_wgrad_impl(U, V, dWref, n, h, w, Cu, Cv, u_nchw, u_act_src, u_act, ws, ws_floats, nullptr, nullptr, stream);
}

// mvk_conv4s2_wgrad on scaled fp16 pairs (imgwgrad_kernel NP = 2: one accumulator per tile, V carries the 2^11 in a third
// piece); u_amax / v_amax: device scalars >= max |U| / max |V|.  NHWC U, no fused activation.
int mvk_conv4s2_wgrad_scaled_ok(int n, int h, int w, int Cu, int Cv) {
  return n / 4 >= imgconv_min_images() && imgconv_pair(h, w, Cu, Cv) && (h != 4 || n % 2 == 0) ? 1 : 0;
}
int mvk_conv4s2_wgrad_s(const float* U, const float* V, float* dWref, int n, int h, int w, int Cu, int Cv, const float* u_amax,
                        const float* v_amax, float* ws, int64_t ws_floats, void* stream) {
  if (!u_amax || !v_amax || !ws || !mvk_conv4s2_wgrad_scaled_ok(n, h, w, Cu, Cv)) return MVK_EINVAL;
  return conv4s2_wgrad_impl(U, V, dWref, n, h, w, Cu, Cv, 0, nullptr, MVK_ACT_NONE, ws, ws_floats, u_amax, v_amax, stream);
}

// dWref[ci][co][4][4] += z[n,ci]^T dY[n,(tap,co)]  (ConvTranspose2d(L,C,4,1,0) on a 1x1 input)
int mvk_unflatten_wgrad(const float* Z, const float* dY, float* dWref, int n, int Cin, int Cout, float* ws,
                        int64_t ws_floats, void* stream) {
  if (!Z || !dY || !dWref || n < 0 || Cin <= 0 || Cout <= 0) return MVK_EINVAL;
  GemmDesc d{};
  plain_a(d.a, Z, 1, Cin, Cin, n);  // A'[i=ci][kk=row] = Z[row*Cin + ci]
  plain_b(d.b, dY, 16 * Cout, 1, n, 16 * Cout);
  d.e = Epilogue{};
  d.e.out = dWref;
  d.e.kind = E_UNFLATREF;
  d.e.bias_mod = 1;
  d.e.atomic = 1;
  d.e.Cu = Cout;
  d.e.OH = d.e.OW = 1;
  d.M = Cin;
  d.N = 16 * Cout;
  d.K = n;
  return launch_splitk(d, ws, ws_floats, 512, mvk_stream(stream));
}

// Backward of the (embedding, log-covariance) heads in one launch + the ordered finishes of its row-group slabs.
int mvk_heads_bwd(const float* X, int x_act, const float* dY0, const float* dY1, const float* W0, const float* W1,
                  int64_t w_sk, int64_t w_sn, int flat_c, float* dX, float* dW0, float* dW1, float* db0, float* db1,
                  float* dbprev, int M, int N, int K, float* ws, int64_t ws_floats, void* stream) {
  if (M == 0) return MVK_OK;
  if (!X || !dY0 || !W0 || !dW0 || (dY1 && (!W1 || !dW1)) || M < 0 || N <= 0 || K <= 0) return MVK_EINVAL;
  hipStream_t s = mvk_stream(stream);
  const int nh = dY1 ? 2 : 1, rgs = (M + 127) / 128;
  const long long wtot = (long long)N * K;
  // slab regions: the deferred arena where the target is part of the flat gradient buffer, else the caller's workspace
  float* outs[5] = {dW0, dW1, db0, db1, dbprev};
  // with flat_c the layer below is a convolution: its bias gradient has flat_c channels, and the [rg][K] slab of column
  // sums IS [rg * taps][flat_c] partial rows
  const int pc = flat_c > 0 ? flat_c : K, prow = K / pc;
  if (flat_c > 0 && K % flat_c != 0) return MVK_EINVAL;
  const long long cnt[5] = {wtot, wtot, N, N, pc};
  float* slab[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  bool deferred[5] = {false, false, false, false, false};
  float* dummy = nullptr;  // bias slabs are always written by the kernel: without a target they go to the workspace
  long long ws_used = 0;
  for (int i = 0; i < 5; ++i) {
    const bool head1 = (i == 1 || i == 3);
    if (head1 && nh == 1) continue;
    if (i == 4 && !dbprev) continue;
    const long long need = (cnt[i] * rgs * (i == 4 ? prow : 1) + 3) & ~3LL;
    if (outs[i]) slab[i] = defer_scratch(outs[i], need, s);
    deferred[i] = slab[i] != nullptr;
    if (!slab[i]) {
      if (!ws || ws_used + need > ws_floats) return MVK_EINVAL;
      slab[i] = ws + ws_used;
      ws_used += need;
    }
  }
  (void)dummy;
  int nz = 0;
  const int rc = heads_bwd_launch(X, x_act, dY0, dY1, W0, W1, w_sk, w_sn, flat_c, dX, slab[0], slab[1], slab[2], slab[3],
                                  slab[4], M, N, K, &nz, s);
  if (rc != MVK_OK) return rc == 1 ? MVK_EINVAL : rc;
  for (int i = 0; i < 5; ++i) {
    if (!slab[i] || !outs[i]) continue;
    const int nrows = nz * (i == 4 ? prow : 1);
    const int r = deferred[i] ? defer_push_plain(outs[i], slab[i], cnt[i], nrows, cnt[i], s)
                              : colsum_finish_any(slab[i], nrows, (int)cnt[i], outs[i], s);
    if (r != MVK_OK) return r;
  }
  return MVK_OK;
}

// dWref[cv][cu][4][4] += H[n,(tap,cu)]^T dY[n,cv]  (Conv2d(C,L,4,2,0) heads on a 4x4 input)
int mvk_flatten_wgrad(const float* H, const float* dY, float* dWref, int n, int Cu, int Cv, float* ws,
                      int64_t ws_floats, void* stream) {
  if (!H || !dY || !dWref || n < 0 || Cu <= 0 || Cv <= 0) return MVK_EINVAL;
  GemmDesc d{};
  plain_a(d.a, H, 1, 16 * Cu, 16 * Cu, n);
  plain_b(d.b, dY, Cv, 1, n, Cv);
  d.e = Epilogue{};
  d.e.out = dWref;
  d.e.kind = E_CONVREF;
  d.e.bias_mod = 1;
  d.e.atomic = 1;
  d.e.Cu = Cu;
  d.e.OH = d.e.OW = 1;
  d.M = 16 * Cu;
  d.N = Cv;
  d.K = n;
  return launch_splitk(d, ws, ws_floats, 512, mvk_stream(stream));
}

}  // extern "C"

extern "C" int64_t mvk_splitk_workspace_floats(int rows, int cols, int reduce_len) {
  // at most `target` (<= 1024) slices, never more slices than k-tiles
  long long ktiles = ((long long)reduce_len + mvk::BK - 1) / mvk::BK;
  long long z = ktiles < 1024 ? ktiles : 1024;
  return (int64_t)(z * (long long)rows * cols);
}
