// fp32 -> three bf16 pieces (x = x0 + x1 + x2, round-to-nearest, |x - (x0+x1+x2)| <= 2^-26 |x|): shared by the
// split-precision GEMM engine (igemm_bf.hpp) and the register-stationary convolution kernels (imgconv.hip).
#pragma once
#include "common.hpp"

namespace mvk {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// two fp32 values -> three dwords, each holding the (lo = x, hi = y) pair of one bf16 piece.
// Scalar subtractions on purpose (built with -fno-slp-vectorize): v_pk_add_f32 needs aligned register pairs, which
// costs copies of freshly loaded registers (and the waits that go with them) beside the MFMAs.
__device__ __forceinline__ void bf3_split(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
  const bf16x2 a0 = __builtin_convertvector(f32x2{x, y}, bf16x2);
  p0 = __builtin_bit_cast(unsigned, a0);
  const float rx1 = x - __uint_as_float(p0 << 16), ry1 = y - __uint_as_float(p0 & 0xffff0000u);
  const bf16x2 a1 = __builtin_convertvector(f32x2{rx1, ry1}, bf16x2);
  p1 = __builtin_bit_cast(unsigned, a1);
  const float rx2 = rx1 - __uint_as_float(p1 << 16), ry2 = ry1 - __uint_as_float(p1 & 0xffff0000u);
  const bf16x2 a2 = __builtin_convertvector(f32x2{rx2, ry2}, bf16x2);
  p2 = __builtin_bit_cast(unsigned, a2);
}

// ---- scaled fp16 pairs (2 pieces, 3 products): Ootomo & Yokota, "Recovering single precision accuracy from Tensor Cores
// while surpassing the FP32 theoretical peak performance" (2022) --------------------------------------------------------------
// x * s = hi + lo / 2048 with hi = fp16(x s), lo = fp16((x s - hi) 2048): |x s - (hi + lo / 2048)| <= 2^-23 |x s|, and the
// product of two such pairs is hi hi' + (hi lo' + lo hi') / 2048 up to 2^-22 (the lo lo' term): three fp16 MFMAs instead of
// the six bf16 ones, the two cross terms in an accumulator of their own.  fp16 has 5 exponent bits, so every operand
// TENSOR carries a power-of-two scale s that puts its largest magnitude in [2^13, 2^14): an element then keeps full
// precision down to 2^-28 of the tensor's maximum (hi and the pre-scaled lo are both normal fp16 numbers there) and degrades
// gradually below that — an absolute error of 2^-39 of the maximum, nothing in a dot product that contains the maximum's
// order of magnitude.  The scale needs an upper bound of max |x| BEFORE the operand is converted: the producers of the
// tensors publish it (atomic max in their epilogues, `amax` arguments of the C-ABI), mvk_amax computes it for the others.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// power of two s with amax * s in [2^13, 2^14) (clamped to [2^-126, 2^126]: amax = 0 or denormal gives 2^126)
__device__ __forceinline__ float f16_scale_of(float amax) {
  const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu);
  int se = 267 - e;
  se = se < 1 ? 1 : (se > 253 ? 253 : se);
  return __uint_as_float((unsigned)se << 23);
}
__device__ __forceinline__ float f16_inv_scale(float s) { return __uint_as_float((254u << 23) - __float_as_uint(s)); }

// two SCALED fp32 values -> (hi pair, lo pair), each dword = (low half: x, high half: y)
__device__ __forceinline__ void f16_split(float x, float y, unsigned& hi, unsigned& lo) {
  const f16x2 h = __builtin_convertvector(f32x2{x, y}, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  const float rx = (x - (float)h[0]) * 2048.f, ry = (y - (float)h[1]) * 2048.f;
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{rx, ry}, f16x2));
}

// workgroup-wide maximum of a non-negative value (wave shuffles, then `red[waves]` in LDS), then ONE atomic per workgroup: the
// bit patterns of non-negative floats order like unsigned integers (+inf above every finite value); *dst must hold 0 (or a
// bound to keep) before the launch.  Atomics on one address serialise at ~10 ns each: one per wave of a 2048-workgroup launch
// was 80 us.  Every thread of the workgroup must call it (it synchronises).
__device__ __forceinline__ void amax_publish(float v, float* dst, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  const int wave = threadIdx.x >> 6, waves = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = red[0];
    for (int i = 1; i < waves; ++i) m = fmaxf(m, red[i]);
    // Only a workgroup that would RAISE the published value issues the atomic: same-address atomics serialise at ~10 ns each
    // (the 1280 workgroups of the decoder's first layer spent 12 of their 28 us there), a relaxed agent-scope load does not,
    // and after the first few workgroups almost none exceeds what is already there.
    unsigned* const d = reinterpret_cast<unsigned*>(dst);
    const unsigned mu = __float_as_uint(m);
    if (mu > __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(d, mu);
  }
}

}  // namespace mvk
