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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
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

__device__ __forceinline__ bool mvk_dev_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool mvk_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
