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

}  // namespace mvk
