// gconv_pairs.hip — fp32 grouped Conv1d with 64 channels per group and many taps (the DiT's ConvPositionEmbedding:
// Conv1d(1024, 1024, k = 31, groups = 16, padding = 15) -> Mish, twice; F5_TTS/modeling_modified/F5/modules.py:167-190),
// every fp32 product formed from fp16 {hi, lo * 2^11} pairs (x3_split.h) on the 16-bit matrix cores — with each operand
// split ONCE per workgroup instead of once per k-step.
//
// The LDS-DMA kernel this replaces for that shape (conv_gemm_dma_kernel<float, ..., 64, PAIRS>, gemm_conv.hip) fetches the
// fp32 A rows of every tap as a fresh (shifted) tile and splits both operands in registers in front of the MFMAs: an
// activation is split 31 times (once per tap that reads it), a weight once per row tile and wave — ~70 VALU operations per
// 16-deep k-step against 96 matrix-core cycles, and the kernel was VALU-bound (98 us per launch at one utterance).
// Here one workgroup owns BM output rows x the 64 output channels of one (batch item, group):
//   1. rows [m0 - pad, m0 + BM - pad + taps - 1) x 64 input channels are read once (256 contiguous bytes per row), split,
//      and parked in LDS as two fp16 planes — the aa_conv.hip arrangement: the A fragment of tap t is the same plane read
//      t rows further down, im2col never exists;
//   2. the weights [64 co][taps][64 ci] stream through LDS one tap (64 k) at a time, split on the way in (fp32 from L2,
//      16 values per thread per tap), as two fp16 planes;
//   3. the main loop is ds_read_b128 + v_mfma_f32_32x32x16_f16 only: per 16-deep k-step accA += a_hi b_hi ;
//      accB += a_lo b_hi + a_hi b_lo ; result accA + 2^-11 accB (lo lo ~ 2^-22 of the product is dropped) — the
//      arithmetic of gemm_x3p.hip NP = 2 and of the kernel it replaces;
//   4. accumulators -> LDS -> coalesced epilogue: + bias, Mish, + residual, 16-byte fp32 row stores.
#include "common.h"
#include "gemm_epilogue.h"
#include "mfma.h"
#include "x3_split.h"
#include <atomic>
#include <cstdlib>

namespace mi {

struct GConvPairsDev {
    const float* x; const float* w; const float* bias; float* out; const float* res;
    int T_in, M, taps, pad, act;
    long x_bstride, x_rstride, x_goff, out_bstride, out_rstride;
    int K;          // taps * 64
    int* sat;       // range watch of the activation split (x3_split.h sat_publish); may be null
};

constexpr int GCP_C = 64;                 // channels per group (in and out)
constexpr int GCP_S = GCP_C + 8;          // fp16 row stride of the LDS planes: S / 8 odd -> conflict-free ds_read_b128 rows

// eight fp32 -> the 16-byte slot of the hi plane and of the lo plane
__device__ __forceinline__ void gcp_split8(const float4 u, const float4 v, x3_u4& hi, x3_u4& lo, bool clamp) {
    const float f[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
    unsigned h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (clamp) x2_split_pair(f[2 * q], f[2 * q + 1], h[q], l[q]);
        else x2_split_pair_raw(f[2 * q], f[2 * q + 1], h[q], l[q]);        // weights: finite and in range
    }
    hi = x3_u4{h[0], h[1], h[2], h[3]};
    lo = x3_u4{l[0], l[1], l[2], l[3]};
}

// Mish(v) = v tanh(log(1 + e^v)) = v n / (n + 2), n = e^v (e^v + 2) — the same function without the log1pf / tanhf expansions
// (a few ulp of fp32; the libm form cost ~150 instructions per output and a fifth of this kernel's time).  v > 20: tanh = 1.
__device__ __forceinline__ float gcp_act(float v, int act) {
    if (act == ACT_MISH) {
        const float e = expf(fminf(v, 20.f));
        const float n = e * (e + 2.f);
        return v > 20.f ? v : v * (n / (n + 2.f));
    }
    return act_apply(v, act);
}

template <int BM>
__global__ __launch_bounds__(256, BM <= 128 ? 2 : 1) void gconv_pairs_kernel(const GConvPairsDev p) {
    using MH = Mfma<f16>;
    using FH = typename MH::Frag;
    constexpr int S = GCP_S;
    constexpr int WM = BM / 2, TM = WM / 32;                        // 2 (rows) x 2 (channels) waves, WM x 32 per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int rows_a = BM + p.taps - 1;
    f16* AH = reinterpret_cast<f16*>(smem_raw);                     // rows_a x S : hi plane of the activations
    f16* AL = AH + (size_t)rows_a * S;                              // ... lo plane
    f16* WH = AL + (size_t)rows_a * S;                              // 64 x S : hi plane of the current tap's weights [co][ci]
    f16* WL = WH + GCP_C * S;
    float* OUT = reinterpret_cast<float*>(smem_raw);                // BM x 64 fp32, aliases the planes after the main loop

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, g = blockIdx.y, b = blockIdx.z;
    const float* xb = p.x + (long)b * p.x_bstride + (long)g * p.x_goff;
    const float* wg = p.w + (long)g * GCP_C * p.K;

    // ---- weights of tap t: thread -> (co = v / 8, ci = 8 (v % 8) .. + 8), two of those per thread -----------------
    float4 wreg[2][2];
    auto wload = [&](int t) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int v = tid + q * 256, co = v >> 3, c8 = v & 7;
            const float* src = wg + (long)co * p.K + t * GCP_C + c8 * 8;
            wreg[q][0] = *reinterpret_cast<const float4*>(src);
            wreg[q][1] = *reinterpret_cast<const float4*>(src + 4);
        }
    };
    auto wstore = [&]() {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int v = tid + q * 256, co = v >> 3, c8 = v & 7;
            x3_u4 h, l;
            gcp_split8(wreg[q][0], wreg[q][1], h, l, false);
            *reinterpret_cast<x3_u4*>(WH + co * S + c8 * 8) = h;
            *reinterpret_cast<x3_u4*>(WL + co * S + c8 * 8) = l;
        }
    };
    wload(0);
    // ---- 1. activations: rows m0 - pad .. , split once ---------------------------------------------------------------
    {
        const int t_base = m0 - p.pad;
        const int nitem = rows_a * 8;
        unsigned sat = 0;
        for (int v = tid; v < nitem; v += 256) {
            const int row = v >> 3, c8 = v & 7;
            const int t = t_base + row;
            float4 u = float4{0.f, 0.f, 0.f, 0.f}, w2 = u;
            if (t >= 0 && t < p.T_in) {
                const float* src = xb + (long)t * p.x_rstride + c8 * 8;
                u = *reinterpret_cast<const float4*>(src);
                w2 = *reinterpret_cast<const float4*>(src + 4);
            }
            x3_u4 h, l;
            gcp_split8(u, w2, h, l, true);
            sat |= x2_sat_word(h.x) | x2_sat_word(h.y) | x2_sat_word(h.z) | x2_sat_word(h.w);
            *reinterpret_cast<x3_u4*>(AH + row * S + c8 * 8) = h;
            *reinterpret_cast<x3_u4*>(AL + row * S + c8 * 8) = l;
        }
        sat_publish(p.sat, sat);
    }
    wstore();
    __syncthreads();

    // ---- 2. main loop: one tap (64 k = four 16-deep steps) per iteration ---------------------------------------------
    f32x16 acc[TM], accb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; accb[i][r] = 0.f; }
    const f16* bh_row = WH + (wn * 32 + lr) * S + hi * 8;
    const f16* bl_row = WL + (wn * 32 + lr) * S + hi * 8;
    for (int t = 0; t < p.taps; ++t) {
        if (t + 1 < p.taps) wload(t + 1);
        const f16* ah_row = AH + (wm * WM + lr + t) * S + hi * 8;
        const f16* al_row = AL + (wm * WM + lr + t) * S + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const FH bh = *reinterpret_cast<const FH*>(bh_row + ks * 16);
            const FH bl = *reinterpret_cast<const FH*>(bl_row + ks * 16);
            FH ah[TM], al[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const FH*>(ah_row + i * 32 * S + ks * 16);
                al[i] = *reinterpret_cast<const FH*>(al_row + i * 32 * S + ks * 16);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                accb[i] = MH::mma(al[i], bh, accb[i]);
                accb[i] = MH::mma(ah[i], bl, accb[i]);
                acc[i] = MH::mma(ah[i], bh, acc[i]);
            }
        }
        if (t + 1 < p.taps) {
            __syncthreads();                  // every wave is done with this tap's weights
            wstore();
            __syncthreads();
        }
    }
    __syncthreads();                          // ... and with the activation planes, before OUT overwrites them
    // ---- 3. accumulators -> LDS -> coalesced epilogue ------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            OUT[(wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * GCP_C + wn * 32 + lr] = __builtin_fmaf(accb[i][r], 0x1p-11f, acc[i][r]);
    __syncthreads();
    {
        float* ob = p.out + (long)b * p.out_bstride + (long)g * GCP_C;
        const float* rb = p.res ? p.res + (long)b * p.out_bstride + (long)g * GCP_C : nullptr;
        const float* bias = p.bias ? p.bias + g * GCP_C : nullptr;
        for (int v = tid; v < BM * 16; v += 256) {
            const int row = v >> 4, c4 = v & 15;
            const int m = m0 + row;
            if (m >= p.M) continue;
            float4 o = *reinterpret_cast<const float4*>(OUT + row * GCP_C + c4 * 4);
            if (bias) { const float4 bv = *reinterpret_cast<const float4*>(bias + c4 * 4); o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w; }
            if (p.act != ACT_NONE) { o.x = gcp_act(o.x, p.act); o.y = gcp_act(o.y, p.act); o.z = gcp_act(o.z, p.act); o.w = gcp_act(o.w, p.act); }
            const long gi = (long)m * p.out_rstride + c4 * 4;
            if (rb) { const float4 rv = *reinterpret_cast<const float4*>(rb + gi); o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w; }
            *reinterpret_cast<float4*>(ob + gi) = o;
        }
    }
}

// -----------------------------------------------------------------------------------------------------------------------------
// Round 6: the same convolution with the weights split ONCE PER ENGINE (gconv_pairs_split_weights at load: fp16 {hi, lo} planes
// per (group, tap) in the LDS image, [plane][co][72] with the pad columns zero) and two taps in flight per workgroup.
// The kernel above re-splits every tap's weights in registers (16 values per thread per tap behind two barriers) and runs one wave
// per SIMD, so every LDS round trip and every barrier is an idle matrix pipe: 55 us per launch where the MFMAs alone want ~17.
// Here: eight waves = two tap-parity groups of 2 (rows) x 2 (channels) waves; group kg multiplies taps 2 j + kg, so a SIMD always
// holds two waves at different points of their k-steps.  The weight planes of taps 2 j + 2, 2 j + 3 arrive by LDS-DMA (a straight
// copy of 36 contiguous 1 KB pieces) while taps 2 j, 2 j + 1 are multiplied: one barrier per tap PAIR, no VALU on the weight side.
// The two groups' partial sums meet in the LDS staging of the epilogue (fixed order: even taps + odd taps).
// -----------------------------------------------------------------------------------------------------------------------------
constexpr int GCP_WTAP = 2 * GCP_C * GCP_S;        // fp16 elements of one (group, tap) weight image: hi plane, lo plane
size_t gconv_pairs_planes_bytes(int G, int taps) { return (size_t)G * ((taps + 1) / 2 * 2) * GCP_WTAP * sizeof(f16); }

__global__ __launch_bounds__(256) void gconv_split_weights_kernel(const float* __restrict__ w, f16* __restrict__ wp, int G, int taps, int tp) {
    // one thread per (g, tap, co, 8 input channels); w is [g][co][tap][ci]
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long n = (long)G * taps * GCP_C * 8;
    if (i >= n) return;
    const int c8 = (int)(i & 7), co = (int)((i >> 3) & 63);
    const long gt = i >> 9;
    const int t = (int)(gt % taps), g = (int)(gt / taps);
    const float* src = w + (((long)g * GCP_C + co) * taps + t) * GCP_C + c8 * 8;
    x3_u4 h, l;
    gcp_split8(*reinterpret_cast<const float4*>(src), *reinterpret_cast<const float4*>(src + 4), h, l, false);
    f16* dst = wp + ((long)g * tp + t) * GCP_WTAP + co * GCP_S + c8 * 8;
    *reinterpret_cast<x3_u4*>(dst) = h;
    *reinterpret_cast<x3_u4*>(dst + GCP_C * GCP_S) = l;
}

// wp: gconv_pairs_planes_bytes(G, taps) bytes; w: fp32 [G][64 co][taps][64 ci] (the layout launch_conv_gemm takes)
void gconv_pairs_split_weights(const float* w, void* wp, int G, int taps, hipStream_t s) {
    const int tp = (taps + 1) / 2 * 2;
    MI_HIP(hipMemsetAsync(wp, 0, gconv_pairs_planes_bytes(G, taps), s));          // pad columns and the pad tap of an odd count
    const long n = (long)G * taps * GCP_C * 8;
    hipLaunchKernelGGL(gconv_split_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, (f16*)wp, G, taps, tp);
    MI_HIP(hipGetLastError());
}

template <int BM>
__global__ __launch_bounds__(512, 1) void gconv_pairs2_kernel(const GConvPairsDev p, const f16* __restrict__ wplanes) {
    using MH = Mfma<f16>;
    using FH = typename MH::Frag;
    constexpr int S = GCP_S;
    constexpr int WM = BM / 2, TM = WM / 32;                        // per tap-parity group: 2 (rows) x 2 (channels) waves, WM x 32 per wave
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    const int rows_a = BM + p.taps - 1;
    const int a_el = ((rows_a * S + 511) / 512) * 512;              // plane size rounded to 1 KB: the weight slots stay 1 KB-aligned
    f16* AH = reinterpret_cast<f16*>(smem_raw);
    f16* AL = AH + a_el;
    f16* WS = AL + a_el;                                            // four tap slots: slot = 2 * (pair & 1) + parity
    float* OUT = reinterpret_cast<float*>(smem_raw);                // 2 x BM x 64 fp32 (one per tap-parity group), aliases everything after the main loop

    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, g = blockIdx.y, b = blockIdx.z;
    const float* xb = p.x + (long)b * p.x_bstride + (long)g * p.x_goff;
    const int tp = (p.taps + 1) & ~1, npair = tp >> 1;
    const f16* wg = wplanes + (long)g * tp * GCP_WTAP;

    // one tap pair = 2 * GCP_WTAP halfs = 36 pieces of 1 KB (64 lanes x 16 B), contiguous in HBM and in LDS
    constexpr int NPIECE = 2 * GCP_WTAP * (int)sizeof(f16) / 1024;
    static_assert(NPIECE * 1024 == 2 * GCP_WTAP * (int)sizeof(f16), "a tap pair is whole 1 KB pieces");
    auto wdma = [&](int pair) {
        const char* src = reinterpret_cast<const char*>(wg + (long)pair * 2 * GCP_WTAP) + lane * 16;
        char* dst = reinterpret_cast<char*>(WS + (pair & 1) * 2 * GCP_WTAP);
        for (int pc = wave; pc < NPIECE; pc += 8)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + pc * 1024), (lds_void*)(dst + pc * 1024), 16, 0, 0);
    };
    wdma(0);
    // ---- activations: rows m0 - pad .. , split once (as above) ------------------------------------------------------------------
    {
        const int t_base = m0 - p.pad;
        const int nitem = rows_a * 8;
        unsigned sat = 0;
        for (int v = tid; v < nitem; v += 512) {
            const int row = v >> 3, c8 = v & 7;
            const int t = t_base + row;
            float4 u = float4{0.f, 0.f, 0.f, 0.f}, w2 = u;
            if (t >= 0 && t < p.T_in) {
                const float* src = xb + (long)t * p.x_rstride + c8 * 8;
                u = *reinterpret_cast<const float4*>(src);
                w2 = *reinterpret_cast<const float4*>(src + 4);
            }
            x3_u4 h, l;
            gcp_split8(u, w2, h, l, true);
            sat |= x2_sat_word(h.x) | x2_sat_word(h.y) | x2_sat_word(h.z) | x2_sat_word(h.w);
            *reinterpret_cast<x3_u4*>(AH + row * S + c8 * 8) = h;
            *reinterpret_cast<x3_u4*>(AL + row * S + c8 * 8) = l;
        }
        sat_publish(p.sat, sat);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 acc[TM], accb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; accb[i][r] = 0.f; }
    // (fragment reads one k-step ahead in a second register set, with the pair boundary pipelined as in conv_gemm_dma3_kernel, measured
    // the same: the SIMD's other wave already covers the LDS round trips; the loop sits at the power-limited MFMA rate)
    for (int j = 0; j < npair; ++j) {
        if (j + 1 < npair) wdma(j + 1);                             // into the slots of pair j - 1: every wave passed the barrier behind it
        const int t = 2 * j + kg;
        if (t < p.taps) {                                           // (the last pair of an odd tap count has no odd tap)
            const f16* wt = WS + ((j & 1) * 2 + kg) * GCP_WTAP;
            const f16* bh_row = wt + (wn * 32 + lr) * S + hi * 8;
            const f16* bl_row = bh_row + GCP_C * S;
            const f16* ah_row = AH + (wm * WM + lr + t) * S + hi * 8;
            const f16* al_row = AL + (wm * WM + lr + t) * S + hi * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const FH bh = *reinterpret_cast<const FH*>(bh_row + ks * 16);
                const FH bl = *reinterpret_cast<const FH*>(bl_row + ks * 16);
                FH ah[TM], al[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[i] = *reinterpret_cast<const FH*>(ah_row + i * 32 * S + ks * 16);
                    al[i] = *reinterpret_cast<const FH*>(al_row + i * 32 * S + ks * 16);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    accb[i] = MH::mma(al[i], bh, accb[i]);
                    accb[i] = MH::mma(ah[i], bl, accb[i]);
                    acc[i] = MH::mma(ah[i], bh, acc[i]);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's pieces of pair j + 1 (a whole pair period to land)
        __syncthreads();                                            // ... and every wave is done with pair j's slots (and, at the end, with the planes)
    }
    // ---- accumulators -> LDS (one image per tap-parity group) -> summed in the coalesced epilogue ---------------------------------
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            OUT[(kg * BM + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * GCP_C + wn * 32 + lr] = __builtin_fmaf(accb[i][r], 0x1p-11f, acc[i][r]);
    __syncthreads();
    {
        float* ob = p.out + (long)b * p.out_bstride + (long)g * GCP_C;
        const float* rb = p.res ? p.res + (long)b * p.out_bstride + (long)g * GCP_C : nullptr;
        const float* bias = p.bias ? p.bias + g * GCP_C : nullptr;
        for (int v = tid; v < BM * 16; v += 512) {
            const int row = v >> 4, c4 = v & 15;
            const int m = m0 + row;
            if (m >= p.M) continue;
            float4 o = *reinterpret_cast<const float4*>(OUT + row * GCP_C + c4 * 4);
            const float4 o1 = *reinterpret_cast<const float4*>(OUT + (BM + row) * GCP_C + c4 * 4);
            o.x += o1.x; o.y += o1.y; o.z += o1.z; o.w += o1.w;                // even taps + odd taps
            if (bias) { const float4 bv = *reinterpret_cast<const float4*>(bias + c4 * 4); o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w; }
            if (p.act != ACT_NONE) { o.x = gcp_act(o.x, p.act); o.y = gcp_act(o.y, p.act); o.z = gcp_act(o.z, p.act); o.w = gcp_act(o.w, p.act); }
            const long gi = (long)m * p.out_rstride + c4 * 4;
            if (rb) { const float4 rv = *reinterpret_cast<const float4*>(rb + gi); o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w; }
            *reinterpret_cast<float4*>(ob + gi) = o;
        }
    }
}

// option gconv_two_taps (mi_set_option; MI355TTS_GCONV2=0 at start-up): 0 = the per-launch split kernel even when the planes exist (A/B, tests)
static std::atomic<bool> g_two_taps{[] { const char* e = std::getenv("MI355TTS_GCONV2"); return !(e && e[0] == '0'); }()};
void gconv_pairs_set_option(long v) { g_two_taps = v != 0; }

// true: launched.  false: not this kernel's shape (the caller goes on to its other kernels).
bool launch_gconv_pairs(const ConvGemm& p, hipStream_t s) {
    const int odt = p.out_dtype < 0 ? p.dtype : p.out_dtype;
    if (p.dtype != MI_F32 || odt != MI_F32 || p.N != GCP_C || p.Cin != GCP_C || p.taps < 8 || p.taps > 127 || p.dil != 1) return false;
    if (p.epi != EPI_PLAIN || p.gate || p.accumulate || p.alpha != 1.f || p.out_planes || p.xp) return false;
    if (p.x_rstride % 4 || p.x_bstride % 4 || p.x_goff % 4 || p.out_rstride % 4 || p.out_bstride % 4) return false;
    if (((uintptr_t)p.x | (uintptr_t)p.w | (uintptr_t)p.out | (uintptr_t)p.res | (uintptr_t)p.bias) % 16) return false;
    if (p.pad < 0 || p.pad >= p.taps || p.M <= 0) return false;
    GConvPairsDev d;
    d.x = (const float*)p.x; d.w = (const float*)p.w; d.bias = p.bias; d.out = (float*)p.out; d.res = (const float*)p.res;
    d.T_in = p.T_in; d.M = p.M; d.taps = p.taps; d.pad = p.pad; d.act = p.act;
    d.x_bstride = p.x_bstride; d.x_rstride = p.x_rstride; d.x_goff = p.x_goff; d.out_bstride = p.out_bstride; d.out_rstride = p.out_rstride;
    d.K = p.taps * GCP_C; d.sat = p.sat;
    // rows per workgroup: 128 (two workgroups per CU) or 192 (one), whichever puts fewer rows on the busiest CU — one
    // utterance (B = 2, M = 1126, 16 groups) is 288 workgroups of 128 rows (32 CUs get two: 256 rows) or 192 of 192 rows
    int cus = 256;
    {
        int dev = 0;
        MI_HIP(hipGetDevice(&dev));
        static int cu_count[16] = {0};
        if (!cu_count[dev & 15]) { hipDeviceProp_t pr; MI_HIP(hipGetDeviceProperties(&pr, dev)); cu_count[dev & 15] = pr.multiProcessorCount; }
        cus = cu_count[dev & 15];
    }
    auto load = [&](int bm) { const long n = (long)((p.M + bm - 1) / bm) * p.G * p.B; return ((n + cus - 1) / cus) * bm; };
    const int BM = load(192) < load(128) ? 192 : 128;
    const int rows_a = BM + p.taps - 1;
    size_t lds = ((size_t)2 * rows_a * GCP_S + (size_t)2 * GCP_C * GCP_S) * sizeof(f16);
    lds = std::max(lds, (size_t)BM * GCP_C * 4);
    lds = (lds + 15) / 16 * 16;
    if (lds > 160 * 1024) return false;
    const dim3 grid((p.M + BM - 1) / BM, p.G, p.B);
#define GCP_LAUNCH(BMv)                                                                                                       \
    do {                                                                                                                      \
        auto kfn = gconv_pairs_kernel<BMv>;                                                                                   \
        /* the opt-in is per device and cheap: set it whenever it is needed (a process-wide flag skipped device 1) */ \
        if (lds > 64 * 1024) MI_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        prof_set_kernel("gconv_pairs_kernel<" #BMv "> (fp32 grouped conv, fp16 pairs split once per workgroup)", "", "");      \
        hipLaunchKernelGGL(kfn, grid, dim3(256), lds, s, d);                                                                  \
    } while (0)
    if (p.gcp_w && g_two_taps) {
        // weights pre-split at load (gconv_pairs_split_weights): 192 rows, eight waves, two taps in flight
        constexpr int BM2 = 192;
        const int ra = BM2 + p.taps - 1;
        const size_t a_el = ((size_t)ra * GCP_S + 511) / 512 * 512;
        size_t lds2 = (2 * a_el + (size_t)4 * GCP_WTAP) * sizeof(f16);
        lds2 = std::max(lds2, (size_t)2 * BM2 * GCP_C * 4);
        if (lds2 <= 160 * 1024) {
            auto kfn = gconv_pairs2_kernel<BM2>;
            MI_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            prof_set_kernel("gconv_pairs2_kernel<192> (fp32 grouped conv, fp16 pairs, weights split at load, two taps in flight)", "", "");
            hipLaunchKernelGGL(kfn, dim3((p.M + BM2 - 1) / BM2, p.G, p.B), dim3(512), lds2, s, d, (const f16*)p.gcp_w);
            MI_HIP(hipGetLastError());
            return true;
        }
    }
    if (BM == 192) GCP_LAUNCH(192); else GCP_LAUNCH(128);
#undef GCP_LAUNCH
    MI_HIP(hipGetLastError());
    return true;
}

}  // namespace mi
