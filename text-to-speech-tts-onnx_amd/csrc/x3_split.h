// x3_split.h — exact three-way bf16 split of fp32 values (gemm_x3.hip, attention.hip): a = a1 + a2 + a3 with
// a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2) (round to nearest even; the subtractions are exact in fp32).
#pragma once
#include "common.h"

namespace mi {

typedef unsigned int x3_u4 __attribute__((ext_vector_type(4)));
typedef float x3_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 x3_b2 __attribute__((ext_vector_type(2)));

// two fp32 values -> three packed bf16 pairs (low half = x, high half = y)
__device__ __forceinline__ void x3_split_pair(float x, float y, unsigned& p1, unsigned& p2, unsigned& p3) {
    const x3_b2 b1 = __builtin_convertvector(x3_f2{x, y}, x3_b2);
    p1 = __builtin_bit_cast(unsigned, b1);
    const float rx = x - __uint_as_float(p1 << 16), ry = y - __uint_as_float(p1 & 0xffff0000u);
    const x3_b2 b2 = __builtin_convertvector(x3_f2{rx, ry}, x3_b2);
    p2 = __builtin_bit_cast(unsigned, b2);
    const float sx = rx - __uint_as_float(p2 << 16), sy = ry - __uint_as_float(p2 & 0xffff0000u);
    const x3_b2 b3 = __builtin_convertvector(x3_f2{sx, sy}, x3_b2);
    p3 = __builtin_bit_cast(unsigned, b3);
}

}  // namespace mi
