// gemm_epilogue.h — device-side argument block and the shared epilogues of the implicit-GEMM kernels
// (gemm_conv.hip: register-staged and LDS-DMA kernels; gemm_dma3.hip: the 8-wave 256-row kernels).
#pragma once
#include "common.h"
#include <type_traits>
#include "mfma.h"
#include "x3_split.h"
#include "wave_reduce.h"

namespace mi {

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case ACT_GELU_TANH: {
            const float k0 = 0.7978845608028654f, k1 = 0.044715f;
            return 0.5f * v * (1.f + tanhf(k0 * (v + k1 * v * v * v)));
        }
        case ACT_GELU_ERF: return 0.5f * v * (1.f + erff(v * 0.7071067811865476f));
        case ACT_MISH: {
            float sp = v > 20.f ? v : log1pf(expf(v));
            return v * tanhf(sp);
        }
        case ACT_SILU: return v / (1.f + expf(-v));
        default: return v;
    }
}

struct ConvGemmDev {
    const void* x; const void* w; const float* bias; void* out; const void* res; const float* gate;
    long gate_bstride;
    int G, T_in, M, N, Cin, K, dil, pad;
    long x_bstride, x_rstride, out_bstride, out_rstride, x_goff;
    int act; float alpha; int accumulate; int epi;
    int u, Cout, padT, T_out;
    const float* rope_cos; const float* rope_sin; const void* rope_pack; int heads, head_dim; void* out2; void* out3;
    long v_ld; int Mb;
    int m_off = 0;   // EPI_QKV_ROPE: this launch's row 0 is row m_off of the flattened [batch item][token] axis (launch_conv_gemm's row split)
    const void* zero;      // >= 16 bytes of zeros: source for out-of-range / K-tail vectors of the LDS-DMA path
    int dbg;               // tuning only: 1 = no DMA in the main loop, 2 = no ds_read/MFMA in the main loop
    int lds_epi;           // 1: outputs leave through the LDS-staged, 16-byte-store epilogue (alignment checked on the host)
    int use_buf;           // 128x128 DMA kernel: 1 = LDS-DMA through buffer descriptors (whole K chunks, offsets fit 31 bits)
    int Tm, Tn, RT, RC;    // XCD-aware tile order (DMA kernel): M-tiles per batch item, N-tiles, row tiles (B*Tm), rows per XCD
    float* sk_ws; int* sk_flags; int sk_slots;   // stream-K (gemm_sk.hip): 64 KB partial-tile slot + flag per persistent workgroup
    const void* w3;                              // gemm_x3.hip: weight planes [3][N][K] bf16 (null: not available)
    const void* xp; const void* w3p; int np;     // gemm_x3p.hip: A and B as panel planes (null: not available), np planes each (3 bf16 | 2 fp16)
    void* out_planes;                            // gemm_x3p.hip: output as panel planes of an [M][N] matrix (null: rows in `out`)
    int kv_planes; long k_ld;                    // EPI_QKV_ROPE, fp32: K and V^T leave pre-split (attention.hip KVP), np = kv_planes planes (3 bf16 | 2 fp16 pairs with the low part unscaled, x2u_split_pair; 1 = 3): out2 = [bh][np][k_ld][64], out3 = [bh][np][64][v_ld]
    int tail_tiles, tail_split;                  // gemm_ph8.hip: the last tail_tiles tiles are cut into tail_split K slices (0 / 1: none)
    // AdaLN fold (ConvGemm, common.h): producer side (ln_stats_out) / consumer side (ln_stats_in)
    const float* ln_scale = nullptr; void* ln_out = nullptr; float* ln_stats_out = nullptr; int ln_out_np = 0;
    const float* ln_stats_in = nullptr; const float* ln_p = nullptr; const float* ln_c = nullptr; int ln_dim = 0; float ln_eps = 0.f;
    int ln_final = 0;                            // ln_stats_in = finished (rstd, mean * rstd) per row instead of partial sums
    int* sat = nullptr;                          // fp16-pair producers raise bit 0 when an operand met the fp16 range limit
};

// Shared epilogue: 32x32 accumulator tiles -> bias / activation / gate / residual / alpha / accumulate -> HBM,
// with the ConvTranspose1d index map and the fused QKV bias+RoPE+head-scatter variants.
// All wave-uniform decisions (activation kind, residual, accumulate, index map) are taken ONCE per 32x32 tile, not
// per element: the first version branched per element and cost ~5 us per tile (~20 us fixed per launch).
// 16-bit outputs: tanh-form GELU as x * sigmoid(2t), t = k0 * (x + k1 * x^3), on v_exp_f32 / v_rcp_f32 (two
// transcendental issues and five FMAs instead of libm tanhf's ~30-instruction expansion; the error, ~1e-6 relative, is
// far below the half / bf16 rounding of the stored value).  Measured on the DiT FF1 layer (18016 x 2048 outputs): the
// libm epilogue cost ~45 us of a 155 us launch.  fp32 outputs keep libm (parity path).
__device__ __forceinline__ float gelu_tanh_fast(float v) {
    const float c0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;       // -2 * k0 * log2(e)
    const float c1 = c0 * 0.044715f;
    const float u = v * v;
    const float e = __builtin_amdgcn_exp2f(v * __builtin_fmaf(c1, u, c0));       // exp(-2t)
    return v * __builtin_amdgcn_rcpf(1.f + e);
}

template <int ACT, bool FAST = false>
__device__ __forceinline__ void act16(float (&v)[16]) {
    if constexpr (FAST && ACT == ACT_GELU_TANH) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = gelu_tanh_fast(v[r]);
        return;
    }
    // four values at a time: letting the scheduler interleave all 16 transcendental expansions costs ~90 VGPRs
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[q * 4 + r] = act_apply(v[q * 4 + r], ACT);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <typename TO, int TM, int TN, int WM, int WN>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[TM][TN], const ConvGemmDev& p, int m0, int n0, int b, int g,
                                              int wm, int wn, int lr, int lk) {
    if (p.epi == EPI_QKV_ROPE) {
        // fused bias + interleaved-pair RoPE + head scatter (AttnProcessor, modules.py:459-466, 421-438):
        //   column n -> (which = q|k|v, head, d) ; q,k: z*cos + rot(z)*sin with rot(z)[2j] = -z[2j+1],
        //   rot(z)[2j+1] = z[2j] (the pair partner lives in lane^1 of the accumulator tile) ;
        //   destination layout [b*H + head][token][head_dim] for the attention kernel.
        const int dm = p.heads * p.head_dim;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WN + j * 32 + lr;        // N % 32 == 0 is required: no lane drops out
            const int which = n / dm;
            const int rem = n - which * dm;
            const int hh = rem / p.head_dim, dd = rem - hh * p.head_dim;
            const float bv = p.bias ? p.bias[n] : 0.f;
            const float sgn = (dd & 1) ? 1.f : -1.f;
            const bool vt = which == 2 && p.v_ld > 0;         // V transposed: [bh][d][key]
            TO* base = (TO*)(which == 0 ? p.out : which == 1 ? p.out2 : p.out3);
            const long mstride = vt ? 1 : p.head_dim;
            const int Mb = p.Mb > 0 ? p.Mb : p.M;             // tokens per batch item (batch may be flattened into M)
            const long item_stride = vt ? (long)p.heads * p.head_dim * p.v_ld : (long)p.heads * Mb * p.head_dim;
            const long head_off = vt ? ((long)hh * p.head_dim + dd) * p.v_ld : (long)hh * Mb * p.head_dim + dd;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mb = p.m_off + m0 + wm * WM + i * 32 + 4 * lk;
                const int bi0 = p.Mb > 0 ? mb / Mb : 0;                    // one division per 32-row tile (Mb >= 32)
                const int mloc0 = mb - bi0 * Mb;
                TO* dst0 = base + ((long)b + bi0) * item_stride + head_off;        // this tile touches at most two items
                TO* dst1 = dst0 + item_stride;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int off = (r & 3) + 8 * (r >> 2);
                    const int mg = mb + off;                               // row in the (possibly flattened) M axis
                    float v = acc[i][j][r] + bv;
                    const float partner = __shfl_xor(v, 1);
                    int m = mloc0 + off, bi = bi0;                         // token position inside its batch item
                    if (m >= Mb) { m -= Mb; ++bi; }
                    if (mg - p.m_off >= p.M) { m = 0; bi = 0; }
                    const float c = which < 2 ? p.rope_cos[(long)m * p.head_dim + dd] : 1.f;
                    const float sn = which < 2 ? p.rope_sin[(long)m * p.head_dim + dd] : 0.f;
                    v = v * c + sgn * partner * sn;
                    if (mg - p.m_off < p.M) (bi == bi0 ? dst0 : dst1)[(long)m * mstride] = from_f32<TO>(v);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        return;
    }
    TO* outp = (TO*)p.out + (long)b * p.out_bstride;
    const TO* resp = p.res ? (const TO*)p.res + (long)b * p.out_bstride : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + lr;
        const bool nok = n < p.N;
        const int nc = nok ? n : 0;
        int col, ph = 0;
        if (p.epi == EPI_CONVT) { ph = nc / p.Cout; col = nc - ph * p.Cout; }
        else col = g * p.N + nc;
        const float bv = p.bias ? p.bias[col] : 0.f;
        const float gv = p.gate ? p.gate[(long)b * p.gate_bstride + col] : 1.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * WM + i * 32 + 4 * lk;
            float v[16];
            // element r -> output row; recomputed where needed instead of kept in 32 registers
            auto row_of = [&](int r, bool& o) -> long {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                long row = m;
                o = nok && m < p.M;
                if (p.epi == EPI_CONVT) { row = (long)m * p.u + ph - p.padT; o = o && row >= 0 && row < p.T_out; }
                return (o ? row : 0) * p.out_rstride + col;
            };
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] + bv;
            switch (p.act) {                                  // wave-uniform, once per tile
                case ACT_GELU_TANH: act16<ACT_GELU_TANH, sizeof(TO) == 2>(v); break;
                case ACT_GELU_ERF: act16<ACT_GELU_ERF>(v); break;
                case ACT_MISH: act16<ACT_MISH>(v); break;
                case ACT_SILU: act16<ACT_SILU>(v); break;
                default: break;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] *= gv;
            if (resp) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { bool o; const long ix = row_of(r, o); v[r] += o ? to_f32(resp[ix]) : 0.f; }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] *= p.alpha;
            if (p.accumulate) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { bool o; const long ix = row_of(r, o); v[r] += o ? to_f32(outp[ix]) : 0.f; }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { bool o; const long ix = row_of(r, o); if (o) outp[ix] = from_f32<TO>(v[r]); }
            __builtin_amdgcn_sched_barrier(0);                // keep one tile's addresses live at a time
        }
    }
}

// LDS-staged epilogue for the DMA kernels (EPI_PLAIN / EPI_CONVT): the accumulator layout gives every lane ONE output
// column, so the direct epilogue above leaves the chip as 2- or 4-byte stores (a 256x256 bf16 tile = 128 store
// instructions of 128 bytes per wave; measured: 25-35 us of fixed cost per block wave, more than the K loop of a
// K = 1024 GEMM).  Here each wave parks 32*RT rows x WN columns of fp32 results (bias / activation / gate applied) in
// its own slice of the now idle operand ring, then reads them back row-wise and leaves with 16-byte stores (residual
// and accumulate operands are fetched with 16-byte loads in the same layout).  Wave-local: no block barrier.
template <typename TO, int TM, int TN, int WM, int WN>
__device__ __forceinline__ void gemm_epilogue_lds(f32x16 (&acc)[TM][TN], const ConvGemmDev& p, int m0, int n0, int b, int g,
                                                  int wm, int wn, int lr, int lk, float* stage, int m_end = -1) {
    const int Mlim = m_end >= 0 ? m_end : p.M;
    constexpr int CH = sizeof(TO) == 2 ? 8 : 4;            // columns per lane: one 16-byte store
    constexpr int LPR = WN / CH;                            // lanes per output row
    constexpr int RPI = 64 / LPR;                           // rows per wave instruction
    constexpr int RT = (WN <= 64 && TM % 2 == 0) ? 2 : 1;   // 32-row tiles per pass (<= 16 KB of fp32 per wave)
    const int lane = lk * 32 + lr;
    TO* outp = (TO*)p.out + (long)b * p.out_bstride;
    const TO* resp = p.res ? (const TO*)p.res + (long)b * p.out_bstride : nullptr;
    const int cl = (lane % LPR) * CH, rl0 = lane / LPR;
    const int nchunk = n0 + wn * WN + cl;                   // first of this lane's CH columns
    int ccol = g * p.N + nchunk, cph = 0;
    if (p.epi == EPI_CONVT) { cph = nchunk / p.Cout; ccol = nchunk - cph * p.Cout; }
    const bool cok = nchunk < p.N && lane < RPI * LPR;
#pragma unroll
    for (int i0 = 0; i0 < TM; i0 += RT) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WN + j * 32 + lr;
            const int nc = n < p.N ? n : 0;
            const int col = p.epi == EPI_CONVT ? nc % p.Cout : g * p.N + nc;
            const float bv = p.bias ? p.bias[col] : 0.f;
            const float gv = p.gate ? p.gate[(long)b * p.gate_bstride + col] : 1.f;
#pragma unroll
            for (int ii = 0; ii < RT; ++ii) {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = acc[i0 + ii][j][r] + bv;
                switch (p.act) {
                    case ACT_GELU_TANH: act16<ACT_GELU_TANH, sizeof(TO) == 2>(v); break;
                    case ACT_GELU_ERF: act16<ACT_GELU_ERF>(v); break;
                    case ACT_MISH: act16<ACT_MISH>(v); break;
                    case ACT_SILU: act16<ACT_SILU>(v); break;
                    default: break;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stage[(ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * WN + j * 32 + lr] = v[r] * gv;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // Rows in groups of GRP: all residual / accumulate operands of a group are requested before the first is used, and
        // out-of-range rows are handled by clamping the address and masking the store, not by branching.  (Before, every
        // row was its own branch with load -> wait -> store inside: one 16-byte request in flight per lane, and a workgroup
        // alone on its CU — the 256x256 kernel — spent longer in this loop than in its K loop: the DiT O projection of 8
        // utterances ran 102 us in the model against 64 us without the residual.)
        constexpr int NIT = (RT * 32 + RPI - 1) / RPI, GRP = NIT % 4 == 0 ? 4 : (NIT % 2 == 0 ? 2 : 1);     // 4 rows: 8 more registers per operand kind (8 rows spilled in the 256x256 kernel)
        struct alignas(16) Pk { TO v[CH]; };
#pragma unroll
        for (int it0 = 0; it0 < NIT; it0 += GRP) {
            Pk rv[GRP], ov[GRP];
            long ixv[GRP];
            bool okv[GRP];
#pragma unroll
            for (int gi = 0; gi < GRP; ++gi) {
                const int rr = rl0 + (it0 + gi) * RPI;
                const int m = m0 + wm * WM + i0 * 32 + rr;
                long row = m;
                bool ok = cok && rr < RT * 32 && m < Mlim;                // (RPI need not divide the pass: 96-wide wave tiles)
                if (p.epi == EPI_CONVT) { row = (long)m * p.u + cph - p.padT; ok = ok && row >= 0 && row < p.T_out; }
                okv[gi] = ok;
                ixv[gi] = ok ? row * p.out_rstride + ccol : 0;       // masked lanes read element 0 (always there) and store nothing
            }
            if (resp) {
#pragma unroll
                for (int gi = 0; gi < GRP; ++gi) rv[gi] = *reinterpret_cast<const Pk*>(resp + ixv[gi]);
            }
            if (p.accumulate) {
#pragma unroll
                for (int gi = 0; gi < GRP; ++gi) ov[gi] = *reinterpret_cast<const Pk*>(outp + ixv[gi]);
            }
#pragma unroll
            for (int gi = 0; gi < GRP; ++gi) {
                const int rr = min(rl0 + (it0 + gi) * RPI, RT * 32 - 1);
                float x[CH];
#pragma unroll
                for (int q = 0; q < CH; q += 4) {
                    const float4 t = *reinterpret_cast<const float4*>(&stage[rr * WN + cl + q]);
                    x[q] = t.x; x[q + 1] = t.y; x[q + 2] = t.z; x[q + 3] = t.w;
                }
                if (resp) {
#pragma unroll
                    for (int q = 0; q < CH; ++q) x[q] += to_f32(rv[gi].v[q]);
                }
#pragma unroll
                for (int q = 0; q < CH; ++q) x[q] *= p.alpha;
                if (p.accumulate) {
#pragma unroll
                    for (int q = 0; q < CH; ++q) x[q] += to_f32(ov[gi].v[q]);
                }
                Pk o;
#pragma unroll
                for (int q = 0; q < CH; ++q) o.v[q] = from_f32<TO>(x[q]);
                if (okv[gi]) *reinterpret_cast<Pk*>(outp + ixv[gi]) = o;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// AdaLN fold (ConvGemm::ln_* in common.h; AdaLayerNorm.forward modules.py:301-305, DiTBlock.forward :599-613).
// The LayerNorm between the residual stream and the QKV / FF1 projections has no launch of its own: the producer of the
// residual row (O / FF2 epilogue) leaves x o (1 + scale) as the next GEMM's A operand plus per-row partial (sum, M2 about the block mean)
// over 32-column blocks (merged Chan-style, wave_reduce.h: no E[x^2] - mean^2 cancellation); the consumer's epilogue finishes  rstd * acc - (mean * rstd) * (W (1 + scale)) + (W shift + b).
// Every partial is a plain store and every sum runs in a fixed order: results are bit-reproducible and do not depend on
// which kernel (tile shape) produced the partials.
// ---------------------------------------------------------------------------------------------------------------------------

// (rstd, mean * rstd) of the 32 rows row0 .. row0 + 31 (clamped to row_last): lane l and lane l + 32 both return row (l & 31).
// The two half-waves each add one half of the row's D / 32 partial pairs in index order, then low half + high half.
__device__ __forceinline__ void ln_rows32(const ConvGemmDev& p, long row0, long row_last, int lr, int lk, float& rstd, float& mrstd) {
    const int nb = p.ln_dim / LN_BLK;                        // a multiple of 4 (ln_dim % 128 == 0, checked on the host)
    long row = row0 + lr;
    row = row < row_last ? row : row_last;
    if (p.ln_final) {           // finished by ln_finalize_kernel (16-bit engines: every column tile would otherwise redo the 32-partial sum)
        const float2 t = reinterpret_cast<const float2*>(p.ln_stats_in)[row];
        rstd = t.x; mrstd = t.y;
        return;
    }
    const float* rowp = p.ln_stats_in + row * (long)(nb * 2);
    const float m0 = rowp[0] * (1.0f / 32.0f);               // reference point of the merge: the first block's mean (wave_reduce.h)
    const float4* sp = reinterpret_cast<const float4*>(rowp) + lk * (nb >> 2);
    LnMerge t;
    for (int i = 0; i < (nb >> 2); ++i) {
        const float4 v = sp[i];
        ln_merge_add(t, v.x, v.y, m0);
        ln_merge_add(t, v.z, v.w, m0);
    }
    LnMerge o;
    o.a = __shfl_xor(t.a, 32); o.b = __shfl_xor(t.b, 32); o.c = __shfl_xor(t.c, 32);
    if (lk) ln_merge_finish(o, t, m0, nb, p.ln_eps, rstd, mrstd);     // low half + high half on both lanes
    else ln_merge_finish(t, o, m0, nb, p.ln_eps, rstd, mrstd);
}

// LDS-staged variant of the fused QKV epilogue for the 128x128 DMA kernel (head_dim 64, one (q|k|v, head) slice per
// 64-column wave tile).  q / k: rows leave as 16-byte stores into [b*H + h][token][64] with the interleaved-pair RoPE
// applied on the 8-column chunk a lane holds.  V (transposed for the attention kernel, [b*H + h][d][key]): the tile is
// read back column-wise with lane = key, so every store instruction writes 32 consecutive keys of one d (64 contiguous
// bytes) instead of 64 two-byte writes to 64 different rows.
// TMQ: 32-row blocks of the wave tile (2: 64 x 64 per wave ; 1: 32 x 64, the eight-wave layout of gemm_x3.hip)
// LN: consumer side of the AdaLN fold (see below): the accumulators carry W (x o (1 + scale)); bias and the LayerNorm's
// mean / rstd enter on the row-wise read-back as rstd * acc - (mean * rstd) * ln_p[col] + ln_c[col]
// PRE (with LN): (rstd, mean * rstd) of the wave's rows were fetched by the caller (at the start of the tile, under its main loop)
template <typename TO, int TMQ = 2, bool LN = false, bool PRE = false>
__device__ __forceinline__ void gemm_epilogue_qkv_lds(f32x16 (&acc)[TMQ][2], const ConvGemmDev& p, int m0, int n0, int b,
                                                      int wm, int wn, int lr, int lk, float* stage,
                                                      const float* pre_rs = nullptr, const float* pre_mr = nullptr, int m_end = -1) {
    // m_end >= 0: rows of this launch at or beyond m_end are not this caller's (gemm_x3d.hip: a 144-row tile ends inside a 32-row block)
    const int Mlim = m_end >= 0 ? m_end : p.M;
    const int lane = lk * 32 + lr;
    const int dm = p.heads * 64;
    const int nbase = n0 + wn * 64;
    const int which = nbase / dm, hh = (nbase - which * dm) >> 6;           // wave-uniform
    const int Mb = p.Mb > 0 ? p.Mb : p.M;
    const int mrow = m0 + wm * (32 * TMQ);                                   // row of this launch
    const int mbase = mrow + p.m_off;                                        // ... of the flattened [batch item][token] axis
    const int bi0 = mbase / Mb, mloc0 = mbase - bi0 * Mb;
    const bool vt = which == 2 && p.v_ld > 0;
    TO* base = (TO*)(which == 0 ? p.out : which == 1 ? p.out2 : p.out3);
    struct alignas(16) Pk { TO v[8]; };
    unsigned qkv_sat = 0;            // fp32 engines with fp16-pair attention operands: range watch (x3_split.h)
    float ln_rs[TMQ], ln_mr[TMQ];
    if constexpr (LN) {
#pragma unroll
        for (int i = 0; i < TMQ; ++i) {
            if constexpr (PRE) { ln_rs[i] = pre_rs[i]; ln_mr[i] = pre_mr[i]; }
            else ln_rows32(p, (long)mbase + i * 32, (long)p.m_off + p.M - 1, lr, lk, ln_rs[i], ln_mr[i]);
        }
    }
    if (!vt) {
        float ln_pv[8], ln_cv[8];
        if constexpr (LN) {
            const int cc = nbase + (lane & 7) * 8;
            const float4 a0 = *reinterpret_cast<const float4*>(p.ln_p + cc), a1 = *reinterpret_cast<const float4*>(p.ln_p + cc + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(p.ln_c + cc), b1 = *reinterpret_cast<const float4*>(p.ln_c + cc + 4);
            ln_pv[0] = a0.x; ln_pv[1] = a0.y; ln_pv[2] = a0.z; ln_pv[3] = a0.w; ln_pv[4] = a1.x; ln_pv[5] = a1.y; ln_pv[6] = a1.z; ln_pv[7] = a1.w;
            ln_cv[0] = b0.x; ln_cv[1] = b0.y; ln_cv[2] = b0.z; ln_cv[3] = b0.w; ln_cv[4] = b1.x; ln_cv[5] = b1.y; ln_cv[6] = b1.z; ln_cv[7] = b1.w;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float bv = (!LN && p.bias) ? p.bias[nbase + j * 32 + lr] : 0.f;
#pragma unroll
            for (int i = 0; i < TMQ; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stage[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * 64 + j * 32 + lr] = acc[i][j][r] + bv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int c8 = (lane & 7) * 8;
        // rows in groups of four: the RoPE table rows of a group are requested before the first is used, out-of-range rows
        // are clamped and their store masked (see gemm_epilogue_lds)
        struct alignas(16) H8 { f16 v[8]; };
        constexpr int GRP = 4;
#pragma unroll
        for (int it0 = 0; it0 < 4 * TMQ; it0 += GRP) {
            int mv[GRP], biv[GRP];
            bool okv[GRP];
            H8 cs[GRP];
#pragma unroll
            for (int gi = 0; gi < GRP; ++gi) {
                const int rr = (lane >> 3) + (it0 + gi) * 8;
                okv[gi] = mrow + rr < Mlim;
                int m = mloc0 + rr, bi = bi0;
                if (m >= Mb) { m -= Mb; ++bi; }
                if (!okv[gi]) { m = 0; bi = bi0; }
                mv[gi] = m; biv[gi] = bi;
            }
            if (which < 2 && p.rope_pack) {
                // four (cos, sin) half pairs = one 16-byte load for the lane's eight columns
#pragma unroll
                for (int gi = 0; gi < GRP; ++gi) cs[gi] = *reinterpret_cast<const H8*>((const f16*)p.rope_pack + (long)mv[gi] * 64 + c8);
            }
#pragma unroll
            for (int gi = 0; gi < GRP; ++gi) {
                const int rr = (lane >> 3) + (it0 + gi) * 8;
                float x[8];
#pragma unroll
                for (int q = 0; q < 8; q += 4) {
                    const float4 t = *reinterpret_cast<const float4*>(&stage[rr * 64 + c8 + q]);
                    x[q] = t.x; x[q + 1] = t.y; x[q + 2] = t.z; x[q + 3] = t.w;
                }
                if constexpr (LN) {
                    const float rs = __shfl(ln_rs[(it0 + gi) >> 2], rr & 31), mr = __shfl(ln_mr[(it0 + gi) >> 2], rr & 31);
#pragma unroll
                    for (int q = 0; q < 8; ++q) x[q] = __builtin_fmaf(rs, x[q], __builtin_fmaf(-mr, ln_pv[q], ln_cv[q]));
                }
                if (which < 2 && p.rope_pack) {
#pragma unroll
                    for (int q = 0; q < 8; q += 2) {
                        const float cc = (float)cs[gi].v[q], ss = (float)cs[gi].v[q + 1];
                        const float e = x[q], o = x[q + 1];
                        x[q] = e * cc - o * ss;
                        x[q + 1] = o * cc + e * ss;
                    }
                } else if (which < 2) {                          // fp32 tables (no packed table given): loaded row by row
                    float c[8], sn[8];
#pragma unroll
                    for (int q = 0; q < 8; q += 4) {
                        const float4 tc = *reinterpret_cast<const float4*>(p.rope_cos + (long)mv[gi] * 64 + c8 + q);
                        const float4 ts = *reinterpret_cast<const float4*>(p.rope_sin + (long)mv[gi] * 64 + c8 + q);
                        c[q] = tc.x; c[q + 1] = tc.y; c[q + 2] = tc.z; c[q + 3] = tc.w;
                        sn[q] = ts.x; sn[q + 1] = ts.y; sn[q + 2] = ts.z; sn[q + 3] = ts.w;
                    }
#pragma unroll
                    for (int q = 0; q < 8; q += 2) {
                        const float e = x[q], o = x[q + 1];
                        x[q] = e * c[q] - o * sn[q];
                        x[q + 1] = o * c[q + 1] + e * sn[q + 1];
                    }
                }
                if constexpr (sizeof(TO) == 4) if (p.kv_planes && which == 1) {
                    // K for the fp32 attention kernel with pre-split operands: the three bf16 pieces of the row's eight values,
                    // one 16-byte store per plane ([bh][plane][key][64])
                    if (p.kv_planes == 2) {          // fp16 {hi, lo} pairs: [bh][2][key][64]
                        x3_u4 pl[2];                  // unscaled low part: the attention kernel sums both parts into one accumulator
                        {
                            unsigned wh[4], wl[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) x2u_split_pair(x[2 * q], x[2 * q + 1], wh[q], wl[q]);
                            pl[0] = x3_u4{wh[0], wh[1], wh[2], wh[3]}; pl[1] = x3_u4{wl[0], wl[1], wl[2], wl[3]};
                            qkv_sat |= x2_sat_word(wh[0]) | x2_sat_word(wh[1]) | x2_sat_word(wh[2]) | x2_sat_word(wh[3]);
                        }
                        bf16* kp = (bf16*)p.out2 + ((((long)b + biv[gi]) * p.heads + hh) * 2 * p.k_ld + mv[gi]) * 64 + c8;
                        if (okv[gi]) {
                            *reinterpret_cast<x3_u4*>(kp) = pl[0];
                            *reinterpret_cast<x3_u4*>(kp + p.k_ld * 64) = pl[1];
                        }
                        continue;
                    }
                    unsigned p1[4], p2[4], p3[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) x3_split_pair(x[2 * q], x[2 * q + 1], p1[q], p2[q], p3[q]);
                    bf16* kp = (bf16*)p.out2 + ((((long)b + biv[gi]) * p.heads + hh) * 3 * p.k_ld + mv[gi]) * 64 + c8;
                    if (okv[gi]) {
                        *reinterpret_cast<x3_u4*>(kp) = x3_u4{p1[0], p1[1], p1[2], p1[3]};
                        *reinterpret_cast<x3_u4*>(kp + p.k_ld * 64) = x3_u4{p2[0], p2[1], p2[2], p2[3]};
                        *reinterpret_cast<x3_u4*>(kp + 2 * p.k_ld * 64) = x3_u4{p3[0], p3[1], p3[2], p3[3]};
                    }
                    continue;
                }
                if constexpr (sizeof(TO) == 4) if (p.kv_planes == 2 && which == 0) {
                    // q leaves as fp32 rows and is split into fp16 pairs (times log2 e) inside the attention kernel: watch its range here
#pragma unroll
                    for (int q = 0; q < 8; ++q) qkv_sat |= (fabsf(x[q]) * 1.4426950408889634f >= 65504.f || x[q] != x[q]) ? 1u : 0u;
                }
                Pk o8;
#pragma unroll
                for (int q = 0; q < 8; ++q) o8.v[q] = from_f32<TO>(x[q]);
                if (okv[gi]) *reinterpret_cast<Pk*>(base + (((long)b + biv[gi]) * p.heads + hh) * Mb * 64 + (long)mv[gi] * 64 + c8) = o8;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        const int row = lane & 31, dh = lane >> 5;
        // LN: the two per-column vectors of the wave's 64 columns, one column per lane (fetched through a lane shuffle below: a
        // global load per column inside the read-back loop was a chain of 32 dependent round trips per block)
        float ln_pl = 0.f, ln_cl = 0.f;
        if constexpr (LN) { ln_pl = p.ln_p[nbase + lane]; ln_cl = p.ln_c[nbase + lane]; }
#pragma unroll
        for (int i = 0; i < TMQ; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float bv = (!LN && p.bias) ? p.bias[nbase + j * 32 + lr] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stage[((r & 3) + 8 * (r >> 2) + 4 * lk) * 65 + j * 32 + lr] = acc[i][j][r] + bv;
            }
            // the value of (row, column dd) on the read-back; LN: this lane's row is row `lane & 31` = lr of block i
            auto vt_val = [&](int dd) __attribute__((always_inline)) -> float {
                const float a = stage[row * 65 + dd];
                if constexpr (LN) return __builtin_fmaf(ln_rs[i], a, __builtin_fmaf(-ln_mr[i], __shfl(ln_pl, dd), __shfl(ln_cl, dd)));
                else return a;
            };
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int rr = i * 32 + row;
            const bool ok = mrow + rr < Mlim;
            int m = mloc0 + rr, bi = bi0;
            if (m >= Mb) { m -= Mb; ++bi; }
            TO* dst = base + (((long)b + bi) * p.heads + hh) * 64 * p.v_ld + m;
            if (sizeof(TO) == 4 && p.kv_planes) {   // (fp32 instantiations only reach this with kv_planes set)
                // V^T as three bf16 planes [bh][plane][d][v_ld]: 2-byte stores, 32 consecutive keys per instruction
                const long pstride = 64 * p.v_ld;
                if (p.kv_planes == 2) {              // fp16 {hi, lo} pairs: [bh][2][d][v_ld]
                    unsigned short* vp = (unsigned short*)p.out3 + (((long)b + bi) * p.heads + hh) * 2 * 64 * p.v_ld + m;
#pragma unroll 8
                    for (int d = 0; d < 32; ++d) {
                        const int dd = dh * 32 + d;
                        unsigned ph, pl;
                        x2u_split_pair(vt_val(dd), 0.f, ph, pl);
                        qkv_sat |= x2_sat_word(ph);
                        if (ok) {
                            vp[(long)dd * p.v_ld] = (unsigned short)ph;
                            vp[pstride + (long)dd * p.v_ld] = (unsigned short)pl;
                        }
                    }
                } else {
                unsigned short* vp = (unsigned short*)p.out3 + (((long)b + bi) * p.heads + hh) * 3 * 64 * p.v_ld + m;
#pragma unroll 8
                for (int d = 0; d < 32; ++d) {
                    const int dd = dh * 32 + d;
                    unsigned p1, p2, p3;
                    x3_split_pair(vt_val(dd), 0.f, p1, p2, p3);
                    if (ok) {
                        vp[(long)dd * p.v_ld] = (unsigned short)p1;
                        vp[pstride + (long)dd * p.v_ld] = (unsigned short)p2;
                        vp[2 * pstride + (long)dd * p.v_ld] = (unsigned short)p3;
                    }
                }
                }
            } else {
#pragma unroll 8
                for (int d = 0; d < 32; ++d) {
                    const int dd = dh * 32 + d;
                    const float v = vt_val(dd);
                    if (ok) dst[(long)dd * p.v_ld] = from_f32<TO>(v);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if constexpr (sizeof(TO) == 4) sat_publish(p.sat, qkv_sat);
}


template <typename TO> __device__ __forceinline__ void ln_act8(float (&x)[8], int act) {
    switch (act) {                                            // wave-uniform
        case ACT_GELU_TANH:
            // x * sigmoid(2t) on v_exp_f32 / v_rcp_f32 for every output type (round 4: fp32 too).  The form has no cancellation
            // (0.5 x (1 + tanh t) loses the digits of 1 + tanh t for negative t in fp32, libm or not); its error is the ~1 ulp of
            // each of the two instructions plus the rounding of the exponent, <= 3e-7 |x| — the size of the reference formula's own
            // fp32 rounding.  The libm tanhf expansion cost ~6 us of the FF1 launch's epilogue (profiles/r4/x3p_epilogue_cost.txt).
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = gelu_tanh_fast(x[q]);
            break;
        case ACT_GELU_ERF:
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = act_apply(x[q], ACT_GELU_ERF);
            break;
        case ACT_MISH:
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = act_apply(x[q], ACT_MISH);
            break;
        case ACT_SILU:
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = act_apply(x[q], ACT_SILU);
            break;
        default: break;
    }
}

// CONSUMER with a plain epilogue (FF1): v = act(rstd * acc - (mean * rstd) * ln_p[col] + ln_c[col]).  The wave tile is TM
// 32-row blocks x WN = 32 * TN columns, first row `mw` of this launch, first column `nc0`; it leaves as rows of TO (16-bit
// engines: p.out) or as panel planes of NP planes (fp32 engines: p.out_planes), one 16-byte store per lane and plane.
template <typename TO, int TM, int TN, int NP, bool PRE = false>
__device__ __forceinline__ void gemm_epilogue_ln_in(f32x16 (&acc)[TM][TN], const ConvGemmDev& p, int mw, int nc0, int lr, int lk, float* stage,
                                                    const float* pre_rs = nullptr, const float* pre_mr = nullptr, int m_end = -1) {
    const int Mlim = m_end >= 0 ? m_end : p.M;
    constexpr int WN = 32 * TN, LPR = WN / 8, RPI = 64 / LPR, NIT = 32 / RPI;
    const int lane = lk * 32 + lr;
    const int c8 = (lane % LPR) * 8, rl0 = lane / LPR;
    const int col = nc0 + c8;
    float pv[8], cv[8];
    {
        const float4 a0 = *reinterpret_cast<const float4*>(p.ln_p + col), a1 = *reinterpret_cast<const float4*>(p.ln_p + col + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(p.ln_c + col), b1 = *reinterpret_cast<const float4*>(p.ln_c + col + 4);
        pv[0] = a0.x; pv[1] = a0.y; pv[2] = a0.z; pv[3] = a0.w; pv[4] = a1.x; pv[5] = a1.y; pv[6] = a1.z; pv[7] = a1.w;
        cv[0] = b0.x; cv[1] = b0.y; cv[2] = b0.z; cv[3] = b0.w; cv[4] = b1.x; cv[5] = b1.y; cv[6] = b1.z; cv[7] = b1.w;
    }
    unsigned sat = 0;
    struct alignas(16) Pk { TO v[8]; };
    // the statistics of ALL the tile's row blocks are requested before the first block is staged: one L2 round trip per tile, not
    // one per 32-row block (measured on the 256x256 kernel: a dependent load in front of each of a wave's four blocks cost ~10 us per launch)
    float rs_all[TM], mr_all[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if constexpr (PRE) { rs_all[i] = pre_rs[i]; mr_all[i] = pre_mr[i]; }
        else ln_rows32(p, (long)mw + i * 32, (long)p.M - 1, lr, lk, rs_all[i], mr_all[i]);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const float rstd = rs_all[i], mrstd = mr_all[i];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * lk) * WN + j * 32 + lr] = acc[i][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int rr = rl0 + it * RPI;
            const int m = mw + i * 32 + rr;
            const float rs = __shfl(rstd, rr), mr = __shfl(mrstd, rr);
            const float4 t0 = *reinterpret_cast<const float4*>(&stage[rr * WN + c8]);
            const float4 t1 = *reinterpret_cast<const float4*>(&stage[rr * WN + c8 + 4]);
            float x[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = __builtin_fmaf(rs, x[q], __builtin_fmaf(-mr, pv[q], cv[q]));
            ln_act8<TO>(x, p.act);
            if constexpr (sizeof(TO) == 4) {
                x3_u4 pl[NP];
                xnp_split8_sat<NP>(x, pl, sat);
                if (m < Mlim) {
                    unsigned char* dst = (unsigned char*)p.out_planes + x3p_slot_offset(m, col >> 3, p.N >> 5, NP);
#pragma unroll
                    for (int q = 0; q < NP; ++q) *reinterpret_cast<x3_u4*>(dst + q * X3P_PLANE) = pl[q];
                }
            } else {
                Pk o;
#pragma unroll
                for (int q = 0; q < 8; ++q) o.v[q] = from_f32<TO>(x[q]);
                if (m < Mlim) *reinterpret_cast<Pk*>((TO*)p.out + (long)m * p.out_rstride + col) = o;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (sizeof(TO) == 4 && NP == 2) sat_publish(p.sat, sat);
}

// PRODUCER (O / FF2 projections): x_new = res + gate * (acc + bias) -> fp32 rows in p.out (the residual stream); the same rows
// o (1 + ln_scale) -> the next GEMM's A operand (TA = float: panel planes of NP planes; else rows of TA, [M][N]); partial
// (sum, M2 about the block mean) of x_new per 32-column block -> ln_stats_out (wave_reduce.h).  Wave tile as above.
template <typename TA, int TM, int TN, int NP>
__device__ __forceinline__ void gemm_epilogue_resid_ln(f32x16 (&acc)[TM][TN], const ConvGemmDev& p, int mw, int nc0, int lr, int lk, float* stage, int m_end = -1) {
    const int Mlim = m_end >= 0 ? m_end : p.M;
    constexpr int WN = 32 * TN, LPR = WN / 8, RPI = 64 / LPR, NIT = 32 / RPI;
    const int lane = lk * 32 + lr;
    const int c8 = (lane % LPR) * 8, rl0 = lane / LPR;
    const int col = nc0 + c8;
    const int nb = p.N / LN_BLK;
    float* outp = (float*)p.out;
    const float* resp = (const float*)p.res;
    float gsc[8];
    {
        const float4 a0 = *reinterpret_cast<const float4*>(p.ln_scale + col), a1 = *reinterpret_cast<const float4*>(p.ln_scale + col + 4);
        gsc[0] = 1.f + a0.x; gsc[1] = 1.f + a0.y; gsc[2] = 1.f + a0.z; gsc[3] = 1.f + a0.w;
        gsc[4] = 1.f + a1.x; gsc[5] = 1.f + a1.y; gsc[6] = 1.f + a1.z; gsc[7] = 1.f + a1.w;
    }
    unsigned sat = 0;
    struct alignas(16) Pk { TA v[8]; };
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = nc0 + j * 32 + lr;
            const float bv = p.bias ? p.bias[n] : 0.f;
            const float gv = p.gate ? p.gate[n] : 1.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * lk) * WN + j * 32 + lr] = (acc[i][j][r] + bv) * gv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // all residual rows of the block are requested before the first is used (clamped addresses, masked stores)
        float4 r0[NIT], r1[NIT];
        long ix[NIT];
        bool ok[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int m = mw + i * 32 + rl0 + it * RPI;
            ok[it] = m < Mlim;
            ix[it] = ok[it] ? (long)m * p.out_rstride + col : 0;
            r0[it] = *reinterpret_cast<const float4*>(resp + ix[it]);
            r1[it] = *reinterpret_cast<const float4*>(resp + ix[it] + 4);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int rr = rl0 + it * RPI;
            const int m = mw + i * 32 + rr;
            const float4 t0 = *reinterpret_cast<const float4*>(&stage[rr * WN + c8]);
            const float4 t1 = *reinterpret_cast<const float4*>(&stage[rr * WN + c8 + 4]);
            float x[8] = {t0.x + r0[it].x, t0.y + r0[it].y, t0.z + r0[it].z, t0.w + r0[it].w,
                          t1.x + r1[it].x, t1.y + r1[it].y, t1.z + r1[it].z, t1.w + r1[it].w};
            if (ok[it]) {
                *reinterpret_cast<float4*>(outp + ix[it]) = make_float4(x[0], x[1], x[2], x[3]);
                *reinterpret_cast<float4*>(outp + ix[it] + 4) = make_float4(x[4], x[5], x[6], x[7]);
            }
            // partial statistics of the 32-column block this lane's quad covers (8 columns per lane, 4 lanes): fixed order
            float s1, s2;
            ln_block_stats(x, s1, s2);                             // (sum, M2 about the block mean): wave_reduce.h
            if (ok[it] && (lane & 3) == 0)
                *reinterpret_cast<float2*>(p.ln_stats_out + ((long)m * nb + (col >> 5)) * 2) = make_float2(s1, s2);
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] *= gsc[q];
            if constexpr (sizeof(TA) == 4) {
                x3_u4 pl[NP];
                xnp_split8_sat<NP>(x, pl, sat);
                if (ok[it]) {
                    unsigned char* dst = (unsigned char*)p.ln_out + x3p_slot_offset(m, col >> 3, p.N >> 5, NP);
#pragma unroll
                    for (int q = 0; q < NP; ++q) *reinterpret_cast<x3_u4*>(dst + q * X3P_PLANE) = pl[q];
                }
            } else {
                Pk o;
#pragma unroll
                for (int q = 0; q < 8; ++q) o.v[q] = from_f32<TA>(x[q]);
                if (ok[it]) *reinterpret_cast<Pk*>((TA*)p.ln_out + (long)m * p.N + col) = o;
                // f16 engines: the fold's A operand is the UNNORMALISED residual row o (1 + scale) — not bounded by ~sqrt(d) like
                // LN(x) (1 + scale) + shift is.  A value beyond the fp16 range raises the engine's flag: the call is re-run on the
                // row-norm path (capi.hip f5_run_checked)
                if constexpr (std::is_same<TA, f16>::value) sat |= ok[it] ? f16_range_word(x) : 0u;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr ((sizeof(TA) == 4 && NP == 2) || std::is_same<TA, f16>::value) sat_publish(p.sat, sat);
}

// Epilogue with the OUTPUT as panel planes of the [M][N] result (the A operand of the next linear layer; FF1 -> FF2):
// bias + activation on the accumulators, the 64x64 wave tile through LDS, then every lane takes 8 consecutive columns of
// a row (one 16-byte k-slot of the next GEMM), splits them three ways and stores 16 bytes per plane.
template <int TM, int TN, int NP>
__device__ __forceinline__ void x3p_epilogue_planes(f32x16 (&acc)[TM][TN], const ConvGemmDev& p, int m0, int n0, int wm, int wn,
                                                    int lr, int lk, float* stage, int m_end = -1) {
    const int Mlim = m_end >= 0 ? m_end : p.M;
    const int lane = lk * 32 + lr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const float bv = p.bias ? p.bias[n0 + wn * 64 + j * 32 + lr] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] + bv;
            switch (p.act) {
                case ACT_GELU_TANH: act16<ACT_GELU_TANH, true>(v); break;      // x * sigmoid(2t): see ln_act8 (gemm_epilogue.h)
                case ACT_GELU_ERF: act16<ACT_GELU_ERF>(v); break;
                case ACT_MISH: act16<ACT_MISH>(v); break;
                case ACT_SILU: act16<ACT_SILU>(v); break;
                default: break;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) stage[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * 64 + j * 32 + lr] = v[r];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int nch_out = p.N >> 5;
    const int c8 = (lane & 7) * 8;
    const int s8 = (n0 + wn * 64 + c8) >> 3;
    unsigned char* planes = (unsigned char*)p.out_planes;
    unsigned sat = 0;
#pragma unroll
    for (int it = 0; it < 4 * TM; ++it) {
        const int rr = (lane >> 3) + it * 8;
        const int m = m0 + wm * (32 * TM) + rr;                 // wm counts (32 * TM)-row blocks
        const float4 t0 = *reinterpret_cast<const float4*>(&stage[rr * 64 + c8]);
        const float4 t1 = *reinterpret_cast<const float4*>(&stage[rr * 64 + c8 + 4]);
        const float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        x3_u4 pl[NP];
        xnp_split8_sat<NP>(v, pl, sat);
        if (m < Mlim) {
            unsigned char* dst = planes + x3p_slot_offset(m, s8, nch_out, NP);
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<x3_u4*>(dst + q * X3P_PLANE) = pl[q];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if constexpr (NP == 2) sat_publish(p.sat, sat);
}

typedef __attribute__((address_space(3))) void lds_void;

// gemm_dma3.hip: the 8-wave 256-row kernels (bn = 192 / 256: buffer-descriptor DMA, bn = 128: three-stage ring)
template <typename T, typename TO> void launch_conv_gemm_dma3(const ConvGemmDev& e, int bn, hipStream_t s);
// gemm_sk.hip: stream-K 128x128 kernel for plain linear layers (stages: 0 = automatic)
template <typename T, typename TO> void launch_linear_sk(const ConvGemmDev& e, int stages, hipStream_t s);
// gemm_x3.hip: fp32 linear layers as exact three-way bf16 splits on the stream-K frame
void launch_linear_x3(const ConvGemmDev& e, hipStream_t s);
// gemm_ph8.hip: 256x256 eight-phase kernel for 16-bit linear layers with many row tiles
template <typename T, typename TO> void launch_linear_ph8(const ConvGemmDev& e, hipStream_t s);
void launch_linear_x3p(const ConvGemmDev& e, hipStream_t s);
void x3p_set_option(int which, long v);
// gemm_x3d.hip: exact-fit data-parallel form of linear_x3p (np = 2) — taken by launch_linear_x3p when x3d_plan finds a tiling
bool x3d_plan(const ConvGemmDev& e, int cus, int& tw, int& rgn, int& cgn, int& band);
void launch_linear_x3d(const ConvGemmDev& e, int tw, int rgn, int cgn, int band, hipStream_t s);
void x3d_set_option(int which, long v);
void ph8_set_split_max(long v);
void ph8_set_split_min_nk(long v);


}  // namespace mi
