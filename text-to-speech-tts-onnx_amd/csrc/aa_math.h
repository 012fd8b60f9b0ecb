// aa_math.h — register-level anti-aliased SnakeBeta on a PAIR of adjacent channels, R consecutive outputs per call.
//
// Same polyphase math as aa_act.hip's header comment (reference: act.py:25-29, resample.py:30-34,
// filter.py:94-98), arranged for the VALU:
//   * the two channels of a pair are the two halves of every float2, advanced with packed fp32 FMAs (v_pk_fma_f32: two
//     FMAs per lane per issue slot).  The (even, odd) up-sampled phases and the two halves of every down-FIR tap pair are
//     SEPARATE chains whose only scalar operands are the filter taps — wave-uniform, so they sit in SGPRs;
//   * the x2 of the up-sampler is folded into the up taps, and for 16-bit storage 1/(2*pi) is folded into alpha so
//     sin^2 comes from v_sin_f32 (argument in revolutions) — the fp32 parity path keeps libm sinf;
//   * the "is this up-sampled index inside [lo, hi)" select is only compiled into the EDGE variant; interior tiles
//     (all indices valid) skip it;
//   * R = 16 outputs per run: 21 up-sampled pairs per 16 outputs (1.31x halo redundancy instead of 1.63x at R = 8).
//
// Round 3 — why channel pairs and not (even, odd) pairs of ONE channel, as rounds 1-2 had it.  The old form broadcast
// each input sample x to both halves of the packed FMA; hipcc keeps the samples in arbitrary VGPRs and, for a sample that
// landed in the ODD register of an aligned pair, emits `v_pk_fma_f32 vD, vTap, v[2n:2n+1], vD op_sel:[0,1,0]` (both halves
// read the HIGH register).  In the fused AA+conv kernel exactly one sample per channel got that encoding (xv[1]: it feeds
// up-sampled pairs 0 and 1, i.e. outputs 0 and 1 of a 16-row run), and exactly those outputs were not run-to-run
// reproducible while a co-resident workgroup's waves ran MFMAs on the same SIMD (profiles/r3/aa_conv_opsel_*.txt:
// annotated ISA, the A/B that moves the differing rows with the encoding, and the fix).  With channel pairs no VGPR operand
// is ever broadcast: every VGPR source is a whole aligned pair, and the broadcast operands are SGPRs.
// The arithmetic is unchanged, operation for operation (same FMA chains in the same order, the even and odd
// accumulators of an output added last), so results are bit-identical to the old form.
#pragma once
#include "common.h"

namespace mi {

typedef float aa_f2 __attribute__((ext_vector_type(2)));

struct AATaps {            // built once per kernel from the 12 kaiser-sinc taps h[] (wave-uniform: SGPRs)
    float ue[6], uo[6];    // U_even += ue[e] * x, U_odd += uo[e] * x        (ue[e] = 2*h[2e+1], uo[e] = 2*h[2e])
    float de[6], dO[6];    // y += de[t] * S_even + dO[t] * S_odd            (de[t] = h[2t+1],   dO[t] = h[2t])
};

__device__ __forceinline__ AATaps aa_make_taps(const float* h) {
    AATaps t;
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        t.ue[e] = 2.f * h[2 * e + 1]; t.uo[e] = 2.f * h[2 * e];
        t.de[e] = h[2 * e + 1];       t.dO[e] = h[2 * e];
    }
    return t;
}

__device__ __forceinline__ aa_f2 aa_splat(float s) { return aa_f2{s, s}; }

// xv[j] = (x_c0, x_c1)[mp - 5 + j], j in [0, R+10).  out[r] = (y_c0, y_c1)[mp + r].
// al = alpha per channel (alpha/(2*pi) when FAST), ib = 1/(beta+eps) per channel.
// S index i' of the pair e: even = 2*(mp+e-2), odd = even - 1 ... valid iff lo <= i' < hi (only checked when EDGE).
template <int R, bool FAST, bool EDGE>
__device__ __forceinline__ void aa_run(const aa_f2 (&xv)[R + 10], aa_f2 (&out)[R], const AATaps& tp, aa_f2 al, aa_f2 ib,
                                       int mp, int lo, int hi) {
    aa_f2 ae[R], ao[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { ae[r] = aa_f2{0.f, 0.f}; ao[r] = aa_f2{0.f, 0.f}; }
#pragma unroll
    for (int e = 0; e < R + 5; ++e) {
        aa_f2 ue = aa_f2{0.f, 0.f}, uo = aa_f2{0.f, 0.f};
#pragma unroll
        for (int ee = 0; ee < 6; ++ee) {
            const aa_f2 x = xv[e + 5 - ee];
            ue = __builtin_elementwise_fma(aa_splat(tp.ue[ee]), x, ue);
            uo = __builtin_elementwise_fma(aa_splat(tp.uo[ee]), x, uo);
        }
        const aa_f2 pe = al * ue, po = al * uo;
        aa_f2 sne, sno;
        if constexpr (FAST) {
            sne = aa_f2{__builtin_amdgcn_sinf(pe.x), __builtin_amdgcn_sinf(pe.y)};
            sno = aa_f2{__builtin_amdgcn_sinf(po.x), __builtin_amdgcn_sinf(po.y)};
        } else {
            sne = aa_f2{sinf(pe.x), sinf(pe.y)};
            sno = aa_f2{sinf(po.x), sinf(po.y)};
        }
        aa_f2 se = __builtin_elementwise_fma(sne * sne, ib, ue);
        aa_f2 so = __builtin_elementwise_fma(sno * sno, ib, uo);
        if constexpr (EDGE) {
            const int ie = 2 * (mp + e - 2), io = ie - 1;
            if (ie < lo || ie >= hi) se = aa_f2{0.f, 0.f};
            if (io < lo || io >= hi) so = aa_f2{0.f, 0.f};
        }
#pragma unroll
        for (int tt = 0; tt < 6; ++tt) {
            const int r = e - tt;
            if (r >= 0 && r < R) {
                ae[r] = __builtin_elementwise_fma(aa_splat(tp.de[tt]), se, ae[r]);
                ao[r] = __builtin_elementwise_fma(aa_splat(tp.dO[tt]), so, ao[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) out[r] = ae[r] + ao[r];
}

// The same run with the inputs fetched and the outputs handed over AS THEY ARE NEEDED / FINISHED (ld(j) -> x pair j of the run,
// st(r, y) <- output r): the live state is a window of six inputs and six pairs of accumulators whatever R is, so a run of 16
// (1.31 up-sampler + snake evaluations per output instead of 1.63 at R = 8) fits the register budget of three waves per SIMD.
// Same operations in the same order per output as aa_run: bit-identical results.
template <int R, bool FAST, bool EDGE, typename LD, typename ST>
__device__ __forceinline__ void aa_run_stream(LD&& ld, ST&& st, const AATaps& tp, aa_f2 al, aa_f2 ib, int mp, int lo, int hi) {
    aa_f2 xw[6];                       // xw[k] = x[e + 5 - k] at step e
    aa_f2 ae[6], ao[6];                // accumulators of outputs e - 5 .. e, slot = output index mod 6
#pragma unroll
    for (int k = 0; k < 5; ++k) xw[k + 1] = ld(4 - k);      // x[0..4]: xw[1] = x[4] ... xw[5] = x[0]
#pragma unroll
    for (int e = 0; e < R + 5; ++e) {
        xw[0] = ld(e + 5);
        aa_f2 ue = aa_f2{0.f, 0.f}, uo = aa_f2{0.f, 0.f};
#pragma unroll
        for (int ee = 0; ee < 6; ++ee) {
            ue = __builtin_elementwise_fma(aa_splat(tp.ue[ee]), xw[ee], ue);
            uo = __builtin_elementwise_fma(aa_splat(tp.uo[ee]), xw[ee], uo);
        }
        const aa_f2 pe = al * ue, po = al * uo;
        aa_f2 sne, sno;
        if constexpr (FAST) {
            sne = aa_f2{__builtin_amdgcn_sinf(pe.x), __builtin_amdgcn_sinf(pe.y)};
            sno = aa_f2{__builtin_amdgcn_sinf(po.x), __builtin_amdgcn_sinf(po.y)};
        } else {
            sne = aa_f2{sinf(pe.x), sinf(pe.y)};
            sno = aa_f2{sinf(po.x), sinf(po.y)};
        }
        aa_f2 se = __builtin_elementwise_fma(sne * sne, ib, ue);
        aa_f2 so = __builtin_elementwise_fma(sno * sno, ib, uo);
        if constexpr (EDGE) {
            const int ie = 2 * (mp + e - 2), io = ie - 1;
            if (ie < lo || ie >= hi) se = aa_f2{0.f, 0.f};
            if (io < lo || io >= hi) so = aa_f2{0.f, 0.f};
        }
        if (e < R) { ae[e % 6] = aa_f2{0.f, 0.f}; ao[e % 6] = aa_f2{0.f, 0.f}; }       // output e opens at step e (tap 0)
#pragma unroll
        for (int tt = 0; tt < 6; ++tt) {
            const int r = e - tt;
            if (r >= 0 && r < R) {
                ae[r % 6] = __builtin_elementwise_fma(aa_splat(tp.de[tt]), se, ae[r % 6]);
                ao[r % 6] = __builtin_elementwise_fma(aa_splat(tp.dO[tt]), so, ao[r % 6]);
            }
        }
        if (e >= 5) st(e - 5, ae[(e - 5) % 6] + ao[(e - 5) % 6]);                       // output e - 5 closed with tap 5
#pragma unroll
        for (int k = 5; k > 0; --k) xw[k] = xw[k - 1];
    }
}

}  // namespace mi
