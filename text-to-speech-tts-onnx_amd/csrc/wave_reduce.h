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
}  // namespace mi
