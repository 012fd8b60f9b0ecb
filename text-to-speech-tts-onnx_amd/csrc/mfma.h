// mfma.h — per-dtype MFMA 32x32 tile traits (gfx950).
//   f32  : v_mfma_f32_32x32x2_f32   (exact fp32, K=2 per instruction, one float per operand lane)
//   f16  : v_mfma_f32_32x32x16_f16  (K=16, eight halfs per operand lane)
//   bf16 : v_mfma_f32_32x32x16_bf16
// Operand lane map (both forms): lane l supplies A[i = l&31][k = (l>>5)*KP + e] and
// B[k = (l>>5)*KP + e][j = l&31], e < KP.  C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
#pragma once
#include "common.h"

namespace mi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct Mfma;
template <> struct Mfma<float> {
    static constexpr int KP = 1;
    using Frag = float;
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma<f16> {
    static constexpr int KP = 8;
    using Frag = f16x8;
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma<bf16> {
    static constexpr int KP = 8;
    using Frag = bf16x8;
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

}  // namespace mi
