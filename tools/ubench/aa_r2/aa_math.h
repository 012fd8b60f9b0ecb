// ROUND-2 FORM, kept for the determinism A/B of tools/ubench/aa_race.sh only (the product uses csrc/aa_math.h).
// aa_math.h — register-level anti-aliased SnakeBeta on one channel, R consecutive outputs per call.
//
// Same polyphase math as aa_act.hip's header comment (reference: act.py:25-29, resample.py:30-34,
// filter.py:94-98), arranged for the VALU:
//   * the (even, odd) up-sampled pair and the two halves of every down-FIR tap pair are carried as float2 and
//     advanced with packed fp32 FMAs (v_pk_fma_f32: two FMAs per lane per issue slot);
//   * the x2 of the up-sampler is folded into the up taps, and for 16-bit storage 1/(2*pi) is folded into alpha so
//     sin^2 comes from v_sin_f32 (argument in revolutions) — the fp32 parity path keeps libm sinf;
//   * the "is this up-sampled index inside [lo, hi)" select is only compiled into the EDGE variant; interior tiles
//     (all indices valid) skip it;
//   * R = 16 outputs per run: 21 up-sampled pairs per 16 outputs (1.31x halo redundancy instead of 1.63x at R = 8).
#pragma once
#include "common.h"

namespace mi {

typedef float aa_f2 __attribute__((ext_vector_type(2)));

struct AATaps {            // built once per kernel from the 12 kaiser-sinc taps h[]
    aa_f2 up[6];           // (2*h[2e+1], 2*h[2e])   -> (U_even, U_odd) += up[e] * x
    aa_f2 dn[6];           // (h[2t+1],   h[2t])     -> y += dn[t] . (S_even, S_odd)
};

__device__ __forceinline__ AATaps aa_make_taps(const float* h) {
    AATaps t;
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        t.up[e] = aa_f2{2.f * h[2 * e + 1], 2.f * h[2 * e]};
        t.dn[e] = aa_f2{h[2 * e + 1], h[2 * e]};
    }
    return t;
}

// xv[j] = x[mp - 5 + j], j in [0, R+10).  out[r] = y[mp + r].  al = alpha (or alpha/(2*pi) when FAST), ib = 1/(beta+eps).
// S index i' of the pair e: even = 2*(mp+e-2), odd = even - 1 ... valid iff lo <= i' < hi (only checked when EDGE).
template <int R, bool FAST, bool EDGE>
__device__ __forceinline__ void aa_run(const float (&xv)[R + 10], float (&out)[R], const AATaps& tp, float al, float ib,
                                       int mp, int lo, int hi) {
    aa_f2 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = aa_f2{0.f, 0.f};
#pragma unroll
    for (int e = 0; e < R + 5; ++e) {
        aa_f2 u = aa_f2{0.f, 0.f};
#pragma unroll
        for (int ee = 0; ee < 6; ++ee) {
            const float x = xv[e + 5 - ee];
#if defined(AA_R2_FORCE_HI)
            // experiment (tools/ubench/aa_race.sh): sample AA_R2_FORCE_HI is read through the encoding hipcc chose for
            // xv[1] — the HIGH register of an aligned pair broadcast to both halves, `op_sel:[0,1,0]` — with a recognisable
            // value in the LOW register; every other sample keeps the compiler's choice
            if (e + 5 - ee == AA_R2_FORCE_HI) {
                aa_f2 xp = aa_f2{1000.f, x};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(u) : "v"(tp.up[ee]), "v"(xp));
                continue;
            }
#endif
            u = __builtin_elementwise_fma(tp.up[ee], aa_f2{x, x}, u);
        }
        aa_f2 sn;
        if constexpr (FAST) {
            sn = aa_f2{__builtin_amdgcn_sinf(al * u.x), __builtin_amdgcn_sinf(al * u.y)};
        } else {
            sn = aa_f2{sinf(al * u.x), sinf(al * u.y)};
        }
        aa_f2 s = __builtin_elementwise_fma(sn * sn, aa_f2{ib, ib}, u);
        if constexpr (EDGE) {
            const int ie = 2 * (mp + e - 2), io = ie - 1;
            if (ie < lo || ie >= hi) s.x = 0.f;
            if (io < lo || io >= hi) s.y = 0.f;
        }
#pragma unroll
        for (int tt = 0; tt < 6; ++tt) {
            const int r = e - tt;
            if (r >= 0 && r < R) acc[r] = __builtin_elementwise_fma(tp.dn[tt], s, acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) out[r] = acc[r].x + acc[r].y;
}

}  // namespace mi
