// Shared device/host helpers for libmvk (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mvk.h"

#define MVK_WAVE 64

#define MVK_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return MVK_ELAUNCH;               \
  } while (0)

static inline hipStream_t mvk_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Experiment switches (the MVK_* A/B knobs of DESIGN.md section 9) are read only when MVK_TUNE=1 is set: a process without it
// runs ONE configuration, the shipped one.  (MVK_ENGINE, the documented and tested engine selector, is read directly.)
#include <cstdlib>
#include <cstring>
static inline const char* mvk_tune(const char* name) {
  static const bool on = getenv("MVK_TUNE") && strcmp(getenv("MVK_TUNE"), "1") == 0;
  return on ? getenv(name) : nullptr;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// The same total without the LDS crossbar: four DPP steps sum every 16-lane row into each of its lanes (quad permutes, half-row and
// row mirrors: register-to-register, no ds_bpermute), then the four rows are added in a fixed order from scalar reads.  __shfl_xor is
// six DEPENDENT ds_bpermute_b32 (~700 cycles per call): per image in the fused decoder tail that was ~3 us of a 55-us launch.
// Deterministic; the order differs from wave_sum's, so the two are not interchangeable where bit-identity between paths matters.
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define MVK_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, false))
  MVK_DPP_ADD(0xB1);   // quad_perm [1, 0, 3, 2]
  MVK_DPP_ADD(0x4E);   // quad_perm [2, 3, 0, 1]
  MVK_DPP_ADD(0x141);  // row_half_mirror
  MVK_DPP_ADD(0x140);  // row_mirror
#undef MVK_DPP_ADD
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
}

// the maximum of a NON-NEGATIVE value over the wave as a wave-uniform bit pattern (non-negative floats order like unsigned
// integers): the same register-to-register steps, four scalar reads, scalar maxima — a running maximum kept this way lives in a
// scalar register (the fused decoder tail has no vector register to spare).
__device__ __forceinline__ unsigned wave_max_dpp_bits(float v) {
#define MVK_DPP_MAX(ctrl) v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, false)))
  MVK_DPP_MAX(0xB1);
  MVK_DPP_MAX(0x4E);
  MVK_DPP_MAX(0x141);
  MVK_DPP_MAX(0x140);
#undef MVK_DPP_MAX
  const int b = __builtin_bit_cast(int, v);
  const unsigned r0 = (unsigned)__builtin_amdgcn_readlane(b, 0), r1 = (unsigned)__builtin_amdgcn_readlane(b, 16);
  const unsigned r2 = (unsigned)__builtin_amdgcn_readlane(b, 32), r3 = (unsigned)__builtin_amdgcn_readlane(b, 48);
  const unsigned m01 = r0 > r1 ? r0 : r1, m23 = r2 > r3 ? r2 : r3;
  return m01 > m23 ? m01 : m23;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// activation and its derivative expressed through the OUTPUT y (what the forward saved)
__device__ __forceinline__ float mvk_act(float v, int act) {
  if (act == MVK_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == MVK_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
  if (act == MVK_ACT_LEAKY02) return v > 0.f ? v : 0.2f * v;
  return v;
}
__device__ __forceinline__ float mvk_act_grad_from_out(float y, int act) {
  if (act == MVK_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == MVK_ACT_SIGMOID) return y * (1.f - y);
  if (act == MVK_ACT_LEAKY02) return y > 0.f ? 1.f : 0.2f;  // sign(y) == sign(pre-activation)
  return 1.f;
}

// 1 / (1 + e^-v) on the 1-ulp transcendental instructions: e^-v = 2^t (1 + tl ln 2) with -v log2(e) = t + tl carried as a
// two-term product (the rounding of a plain -v * log2(e) alone is a relative error of |v| 2^-24 in the exponential)
__device__ __forceinline__ float mvk_fast_sigmoid(float v) {
  constexpr float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-08f, LN2 = 0.693147180559945309f;
  const float nv = -v, t = nv * L2E_HI;
  const float tl = fmaf(nv, L2E_HI, -t) + nv * L2E_LO;
  const float e0 = __builtin_amdgcn_exp2f(fminf(t, 126.f));  // 2^126 (1 + tl ln 2) is finite: sigmoid(-87) = 0 to fp32
  const float e = fmaf(e0 * LN2, tl, e0);
  return __builtin_amdgcn_rcpf(1.f + e);
}

__device__ __forceinline__ bool mvk_dev_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool mvk_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- deferred leaf reductions (mvk_defer_begin, mvk.h; implemented in igemm.hip) ---------------------------------------
// Parameter gradients are leaves of the backward pass: nobody reads them before the optimizer.  While deferral is on, a
// producer whose result accumulates into the registered gradient buffer writes its partial results (split-K slabs,
// per-workgroup column sums, weight-gradient slabs) into a private region of the caller's arena instead of the shared
// scratch and queues the ordered finish; mvk_defer_flush runs every queued finish in one launch.
namespace mvk {
// arena region of `floats` floats for a producer whose finish targets `out` on stream `s`; nullptr = not deferred
// (deferral off, `out` outside the gradient buffer, another queued finish targets `out`, arena full): finish immediately
float* defer_scratch(const void* out, long long floats, hipStream_t s);
// true when `out` is null, or could be the target of a queued finish right now (no scratch is taken)
bool defer_free(const void* out);
// queue out[i] += sum_{z < nz} part[z * zstride + i], i < count (z in a fixed order); part from defer_scratch
int defer_push_plain(float* out, const float* part, long long count, int nz, long long zstride, hipStream_t s);
}  // namespace mvk

// ---- device-timestamp profiler (mvk_prof_enable, mvk.h) --------------------------------------------------------------
// An instrumented launch takes the next record of the caller's device array; thread 0 of every workgroup stamps the
// constant-rate clock (s_memrealtime) on entry (min) and exit (max) into one of 32 entries of the record (64 bytes apart:
// the ~1000 workgroups of a streaming kernel leave within a few microseconds, one address would serialise their atomics
// for longer than that); a one-wave fold kernel behind the launch adds max(end) - min(start) (first workgroup's first
// instruction to last workgroup's last one: what a kernel trace reports) to the record's running sum and re-arms it, so a
// record captured into a hipGraph accumulates one duration per replay.  A null record costs one scalar compare.
#define MVK_PROF_ENTRIES 32
#define MVK_PROF_SLOT_U64 (8 + 2 * 8 * MVK_PROF_ENTRIES)  // header {sum, count, ...} + start entries + end entries
struct mvk_prof_slot {
  unsigned long long w[MVK_PROF_SLOT_U64];
};
namespace mvk {
mvk_prof_slot* prof_next(int kind, double work);      // host side (misc.hip): nullptr while the profiler is off
void prof_fold(mvk_prof_slot* slot, hipStream_t s);    // enqueue the fold kernel behind an instrumented launch
}
__device__ __forceinline__ void mvk_prof_begin(mvk_prof_slot* s) {
  if (s && threadIdx.x == 0) {
    const unsigned long long t = (unsigned long long)wall_clock64();
    unsigned long long* e = &s->w[8 + 8 * ((blockIdx.x + blockIdx.y + blockIdx.z) & (MVK_PROF_ENTRIES - 1))];
    if (t < __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(e, t);
  }
}
__device__ __forceinline__ void mvk_prof_end(mvk_prof_slot* s) {
  if (s) {
    __syncthreads();
    if (threadIdx.x == 0)
      atomicMax(&s->w[8 + 8 * MVK_PROF_ENTRIES + 8 * ((blockIdx.x + blockIdx.y + blockIdx.z) & (MVK_PROF_ENTRIES - 1))],
                (unsigned long long)wall_clock64());
  }
}
// barrier-free exit stamp for kernels whose waves leave at different points (one atomic per wave)
__device__ __forceinline__ void mvk_prof_end_wave(mvk_prof_slot* s) {
  if (s && (threadIdx.x & 63) == 0)
    atomicMax(&s->w[8 + 8 * MVK_PROF_ENTRIES + 8 * ((blockIdx.x + blockIdx.y + blockIdx.z) & (MVK_PROF_ENTRIES - 1))],
              (unsigned long long)wall_clock64());
}
