// wave_reduce.h — wave-wide (64-lane) sum / max without LDS-crossbar round trips.
#pragma once
#include "common.h"

namespace mi {

// wave-wide sum / max on the DPP path (no LDS crossbar round trips: six dependent ds_bpermute cost ~0.2 us per
// reduction, a fifth of a decode-step GEMV's life): xor-1 and xor-2 inside the quads, mirrored halves and rows, then the
// four row totals through SGPRs in a fixed order.  Every lane returns the same value.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_value(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);     // row_half_mirror
    v += dpp_mov<0x140>(v);     // row_mirror
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}

// ---- row statistics of the AdaLN fold (gemm_epilogue.h, f5_kernels.hip) ------------------------------------------------------
// A residual row's LayerNorm statistics travel as one (sum, M2) pair per 32-column block, M2 = the squared deviations about the
// BLOCK's own mean, and are merged Chan-style about a reference point (the first block's mean) — the variance never comes from
// E[x^2] - mean^2, whose cancellation error grows as eps * mean^2 / var for rows with a DC offset (nn.LayerNorm is
// two-pass).  Producers: a quad of lanes holds one block as 8 consecutive columns each.  Fixed orders throughout: the pairs are
// bit-identical whichever kernel wrote them, and a consumer epilogue and ln_finalize_kernel finish them to the same bits.
__device__ __forceinline__ void ln_block_stats(const float (&x)[8], float& s1, float& m2) {
    s1 = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
    s1 += dpp_mov<0xB1>(s1);        // lane ^ 1
    s1 += dpp_mov<0x4E>(s1);        // lane ^ 2
    const float mb = s1 * (1.0f / 32.0f);
    float d[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) d[q] = x[q] - mb;
    m2 = ((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) + ((d[4] * d[4] + d[5] * d[5]) + (d[6] * d[6] + d[7] * d[7]));
    m2 += dpp_mov<0xB1>(m2);
    m2 += dpp_mov<0x4E>(m2);
}
// running merge of block pairs about the reference mean m0: a = sum (mb - m0), b = sum (mb - m0)^2, c = sum M2_b
struct LnMerge { float a = 0.f, b = 0.f, c = 0.f; };
__device__ __forceinline__ void ln_merge_add(LnMerge& t, float s1, float m2, float m0) {
    const float dm = s1 * (1.0f / 32.0f) - m0;
    t.a += dm; t.b += dm * dm; t.c += m2;
}
// (rstd, mean * rstd) of a row of nb blocks from the merged sums (lo: blocks 0 .. nb/2 - 1, hi: the rest; lo + hi in that order)
__device__ __forceinline__ void ln_merge_finish(const LnMerge& lo, const LnMerge& hi, float m0, int nb, float eps, float& rstd, float& mrstd) {
    const float a = lo.a + hi.a, b = lo.b + hi.b, c = lo.c + hi.c;
    const float inv_nb = 1.0f / (float)nb;
    const float mean = m0 + a * inv_nb;
    float between = b - a * a * inv_nb;                       // sum (mb - mean)^2: the cancellation here is relative to the spread of the block means only
    between = between > 0.f ? between : 0.f;
    const float var = (c + 32.0f * between) * (inv_nb * (1.0f / 32.0f));       // biased variance
    rstd = 1.0f / sqrtf(var + eps);
    mrstd = mean * rstd;
}
}  // namespace mi
