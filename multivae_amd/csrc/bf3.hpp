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

// The same split from UNSCALED values and a power-of-two scale, on the mixed-precision FMA instructions: v_fma_mixlo/hi_f16
// round x s straight into one half of a register (x s is exact, so this is the fp16 the two-step conversion gives),
// v_fma_mix_f32 reads that half back for the residual, and the 2^11 of the low piece rides on the third pair: 3 instructions
// per element instead of 4 (scale, half a packed conversion, residual, 2^11, half a packed conversion) — bit-identical pieces.
// The staging of the register-stationary kernels runs this 8-16 times per thread and tile in a loop whose budget is ~7 issue
// slots per MFMA (one wave per SIMD).  _su: the scale (and the 2^11) in scalar registers; _sv: a per-lane scale.
__device__ __forceinline__ void f16_split_su(float x, float y, float s, unsigned& hi, unsigned& lo) {
#ifndef MVK_MIX
  return f16_split(x * s, y * s, hi, lo);
#endif
  unsigned h, l;
  float rx, ry;
  const float k = 2048.f;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x), "s"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(y), "s"(s));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(rx) : "v"(x), "s"(s), "v"(h));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(ry) : "v"(y), "s"(s), "v"(h));
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(l) : "v"(rx), "s"(k));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(l) : "v"(ry), "s"(k));
  hi = h;
  lo = l;
}
__device__ __forceinline__ void f16_split_sv(float x, float y, float sx, float sy, unsigned& hi, unsigned& lo) {
#ifndef MVK_MIX
  return f16_split(x * sx, y * sy, hi, lo);
#endif
  unsigned h, l;
  float rx, ry;
  const float k = 2048.f;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x), "v"(sx));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(y), "v"(sy));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(rx) : "v"(x), "v"(sx), "v"(h));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(ry) : "v"(y), "v"(sy), "v"(h));
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(l) : "v"(rx), "s"(k));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(l) : "v"(ry), "s"(k));
  hi = h;
  lo = l;
}

// the three planes of a weight gradient's dY operand (h = fp16(y s), H = fp16(y s 2^11), l = fp16(y s 2^11 - H)) the same way:
// 3 instructions per element instead of 4, bit-identical (y s and y s 2^11 are exact products)
__device__ __forceinline__ void f16_split3_su(float x, float y, float s, float s11, unsigned& h, unsigned& H, unsigned& l) {
#ifndef MVK_MIX
  {
    const f16x2 hh = __builtin_convertvector(f32x2{x * s, y * s}, f16x2);
    const float q0 = x * s11, q1 = y * s11;
    const f16x2 HH = __builtin_convertvector(f32x2{q0, q1}, f16x2);
    const f16x2 ll = __builtin_convertvector(f32x2{q0 - (float)HH[0], q1 - (float)HH[1]}, f16x2);
    h = __builtin_bit_cast(unsigned, hh);
    H = __builtin_bit_cast(unsigned, HH);
    l = __builtin_bit_cast(unsigned, ll);
    return;
  }
#endif
  unsigned a, b, c;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(a) : "v"(x), "s"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(a) : "v"(y), "s"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(b) : "v"(x), "s"(s11));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(b) : "v"(y), "s"(s11));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(c) : "v"(x), "s"(s11), "v"(b));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(c) : "v"(y), "s"(s11), "v"(b));
  h = a;
  H = b;
  l = c;
}

// workgroup-wide maximum of a non-negative value (wave shuffles, then `red[waves]` in LDS), then ONE atomic per workgroup: the
// bit patterns of non-negative floats order like unsigned integers (+inf above every finite value); *dst must hold 0 (or a
// bound to keep) before the launch.  Atomics on one address serialise at ~10 ns each: one per wave of a 2048-workgroup launch
// was 80 us.  Every thread of the workgroup must call it (it synchronises).
// amax_publish_wave: v already is the maximum of its wave (wave-uniform) — no lane shuffles (their address registers would stay
// live across a whole kernel that shuffled in its prologue).
__device__ __forceinline__ void amax_publish_wave(float v, float* dst, float* red);
__device__ __forceinline__ void amax_publish(float v, float* dst, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  amax_publish_wave(v, dst, red);
}
__device__ __forceinline__ void amax_publish_wave(float v, float* dst, float* red) {
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
