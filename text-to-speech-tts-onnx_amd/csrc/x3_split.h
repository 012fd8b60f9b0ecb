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

// Two-way fp16 split: a ~= hi + lo * 2^-11 with hi = fp16(a), lo = fp16((a - hi) * 2^11) — 22 significant bits, the low
// part stored scaled so that it lives in the same exponent range as the high part (no fp16 underflow of the residual: the
// absolute error floor is 2^-36, the relative error 2^-23).  Finite |a| beyond the fp16 range saturates at 65504 (it would
// otherwise turn into inf - inf); the clamp is a compare-select, so a NaN stays a NaN (fminf / fmaxf would hide it).
// Two fp32 values -> packed pairs (low half = x, high half = y).
typedef _Float16 x2_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void x2_split_pair_raw(float x, float y, unsigned& ph, unsigned& pl) {
    const x2_h2 h = __builtin_convertvector(x3_f2{x, y}, x2_h2);
    ph = __builtin_bit_cast(unsigned, h);
    const float rx = (x - (float)h.x) * 2048.f, ry = (y - (float)h.y) * 2048.f;       // exact
    const x2_h2 l = __builtin_convertvector(x3_f2{rx, ry}, x2_h2);
    pl = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void x2_split_pair(float x, float y, unsigned& ph, unsigned& pl) {
    x = x > 65504.f ? 65504.f : x; x = x < -65504.f ? -65504.f : x;
    y = y > 65504.f ? 65504.f : y; y = y < -65504.f ? -65504.f : y;
    x2_split_pair_raw(x, y, ph, pl);
}
// The same pair with the low part UNSCALED: lo = fp16(a - hi).  Both parts then multiply into ONE accumulator (no 2^-11 fold),
// at the price of fp16's subnormal spacing for the residual: absolute operand error 2^-25 instead of 2^-36.  For operands of
// order one whose products are summed into a softmax argument or a probability-weighted mean (attention: q, k, v, p) that is
// below the fp32 rounding of the sum; the linear layers keep the scaled form (arbitrary operand magnitudes).
__device__ __forceinline__ void x2u_split_pair_raw(float x, float y, unsigned& ph, unsigned& pl) {
    const x2_h2 h = __builtin_convertvector(x3_f2{x, y}, x2_h2);
    ph = __builtin_bit_cast(unsigned, h);
    const x2_h2 l = __builtin_convertvector(x3_f2{x - (float)h.x, y - (float)h.y}, x2_h2);
    pl = __builtin_bit_cast(unsigned, l);
}
// The same values with the residuals formed by v_fma_mix_f32 (an FMA whose first operand is read as one fp16 HALF of a register):
// x - (float)h.x is one instruction instead of a v_cvt_f32_f16 and a subtraction — per pair 3 VALU issues instead of 6, in the loop
// whose limit is VALU issue (attention.hip: the probabilities of every 32 x 32 tile).  Exact as before: bit-identical pieces.
__device__ __forceinline__ void x2u_split_pair_raw_mix(float x, float y, unsigned& ph, unsigned& pl) {
#if defined(__HIP_DEVICE_COMPILE__)
    const x2_h2 h = __builtin_convertvector(x3_f2{x, y}, x2_h2);
    ph = __builtin_bit_cast(unsigned, h);
    // v_fma_mixlo / mixhi_f16: fma(fp16 half of ph, -1, fp32) rounded straight into one half of the destination — the exact residual
    // x - hi with ONE rounding to fp16, the same value as cvt(x - (float)hi); the pair costs 3 VALU issues (v_cvt_pk + these two)
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(pl) : "v"(ph), "v"(x), "v"(y));
#else
    x2u_split_pair_raw(x, y, ph, pl);
#endif
}
__device__ __forceinline__ void x2u_split_pair(float x, float y, unsigned& ph, unsigned& pl) {
    x = x > 65504.f ? 65504.f : x; x = x < -65504.f ? -65504.f : x;
    y = y > 65504.f ? 65504.f : y; y = y < -65504.f ? -65504.f : y;
    x2u_split_pair_raw(x, y, ph, pl);
}
// Range watch of the fp16-pair split: non-zero when either half of a packed HIGH word sits at the fp16 limit or beyond
// (|a| >= 65504 after the clamp, inf, nan) — 0x7bff + 0x0401 carries into the sign position of its half.  Producers OR it
// over everything they split and raise the engine's flag word once (sat_publish): a clamped operand is no longer silent.
__device__ __forceinline__ unsigned x2_sat_word(unsigned ph) { return ((ph & 0x7fff7fffu) + 0x04010401u) & 0x80008000u; }
__device__ __forceinline__ void sat_publish(int* flag, unsigned sat) {
    if (flag && sat) atomicOr(flag, 1);          // only lanes that saw a saturated operand get here: rare by construction
}
// non-zero if one of eight fp32 values does not fit fp16 (|v| > 65504 or inf): the range watch of the f16 engines' fold operand
__device__ __forceinline__ unsigned f16_range_word(const float (&v)[8]) {
    const float m = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))),
                          fmaxf(fmaxf(fabsf(v[4]), fabsf(v[5])), fmaxf(fabsf(v[6]), fabsf(v[7]))));
    return m > 65504.0f ? 1u : 0u;              // (a NaN operand is not a range event: it propagates and the parity gates see it)
}
// eight consecutive values -> the 16-byte slot of each plane
template <int NP> __device__ __forceinline__ void xnp_split8(const float (&v)[8], x3_u4 (&pl)[NP]) {
    unsigned w[NP][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (NP == 3) x3_split_pair(v[2 * i], v[2 * i + 1], w[0][i], w[1][i], w[2][i]);
        else x2_split_pair(v[2 * i], v[2 * i + 1], w[0][i], w[1][i]);
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) pl[q] = x3_u4{w[q][0], w[q][1], w[q][2], w[q][3]};
}
// ... and the range watch of the pair format folded in (three bf16 planes cover the whole fp32 exponent range: nothing to watch)
template <int NP> __device__ __forceinline__ void xnp_split8_sat(const float (&v)[8], x3_u4 (&pl)[NP], unsigned& sat) {
    xnp_split8<NP>(v, pl);
    if constexpr (NP == 2) sat |= x2_sat_word(pl[0].x) | x2_sat_word(pl[0].y) | x2_sat_word(pl[0].z) | x2_sat_word(pl[0].w);
}

}  // namespace mi
