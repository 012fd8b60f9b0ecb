// aa_conv.hip — fused [anti-aliased SnakeBeta -> Conv1d(k, dilation) -> +bias (+residual) (*alpha) (+=)]
// for the HBM-bound low-channel BigVGAN stages (C = 96 / 48 / 24 at T = 32k..131k), gfx950.
//
// Reference: one half of an AMPBlock1 iteration, `xt = c(a(x))` (+ `x = xt + x` on the second half),
// BigVGAN/modeling_modified/bigvgan.py:132-140; Activation1d act.py:25-29.
//
// Unfused, an AMP iteration moves 9 tensor passes through HBM ([AA: R+W] [conv: R+W] [AA: R+W]
// [conv+res: 2R+W]); fused it is 5 ([R+W] [2R+W]).  One workgroup owns BM output time steps x all C channels:
//   1. one contiguous, 16-byte-coalesced HBM read of the x tile (+ conv halo + 5-sample AA halo) into LDS
//   2. AA in registers (polyphase FIR x2 -> snake -> FIR /2, fp32; see aa_act.hip) -> activated tile in LDS
//      with a bank-conflict-free row stride; rows outside [0,T) are the conv's zero padding
//   3. implicit-GEMM on MFMA: A fragments are read from the LDS tile at row offsets tap*dilation (im2col
//      never exists), B fragments (weights [co][tap][ci]) stream from L2
//   4. accumulators -> LDS (fp32) -> coalesced epilogue (bias, residual, alpha, accumulate) -> HBM
#include <atomic>
#include "common.h"
#include "mfma.h"
#include "aa_math.h"
#include <cstdlib>

namespace mi {

extern __constant__ float c_h_fused[12];
__constant__ float c_h_fused[12];

static std::atomic<long> g_aa_lds_min = -1;        // > 80 KB: one workgroup per CU (a diagnostic since round 3), see launch_t
bool aa_conv_set_option(const char*, long) { return false; }      // (round 4: "aa_conv_deterministic" removed — the default has been bit-reproducible since round 3; MI355TTS_AACONV_LDS_MIN stays as the diagnostic)

struct AAConvDev {
    const void* x; const void* w; const float* bias; const float* alpha_s; const float* inv_beta; void* out; const void* res;
    int T, C, S, k, dil, K, Kpad, halo, rows_act, rows_x;
    float alpha; int accumulate;
    int w_off;    // element offset of the weight ring inside the LDS block (0: aliases the dead raw-input region)
    int dbg;      // tuning: bit0 skip AA math, bit1 skip MFMA loop, bit2 skip epilogue global traffic
};

template <typename T, int BM, int TN>
__global__ __launch_bounds__(256) void aa_conv_kernel(const AAConvDev p) {
    using MF = Mfma<T>;
    constexpr int KP = MF::KP;
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int R = 16;
    constexpr int WM = BM / 4, TM = WM / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* XS = reinterpret_cast<T*>(smem_raw);                        // rows_x   x C   raw input
    T* AS = XS + (size_t)p.rows_x * p.C;                             // rows_act x S   activated
    float* OUT = reinterpret_cast<float*>(smem_raw);                 // BM x C fp32 (aliases XS/AS after the MFMA loop)

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int m0 = blockIdx.x * BM;
    const int C = p.C, S = p.S;
    const T* xb = (const T*)p.x + (long)b * p.T * C;

    // ---- 1. stage x rows [m0 - halo - 5, ...) ---------------------------------------------------------
    {
        const int cvn = C / VEC, nvec = p.rows_x * cvn;
        const int t_base = m0 - p.halo - 5;
        for (int v = tid; v < nvec; v += 256) {
            const int row = v / cvn, cv = v - row * cvn;
            const int t = t_base + row;
            uint4 raw = make_uint4(0, 0, 0, 0);
            if (t >= 0 && t < p.T) raw = *reinterpret_cast<const uint4*>(xb + (long)t * C + cv * VEC);
            *reinterpret_cast<uint4*>(XS + row * C + cv * VEC) = raw;
        }
    }
    __syncthreads();
    // ---- 2. AA (aa_math.h): runs of R outputs per (channel pair, run), packed-fp32 FIRs -------------------
    {
        constexpr bool FAST = sizeof(T) == 2;
        const AATaps tp = aa_make_taps(c_h_fused);
        const int runs = (p.rows_act + R - 1) / R;
        const int t_act0 = m0 - p.halo;                              // global time of activated row 0
        const int hi2 = 2 * p.T;
        const bool edge = (2 * (t_act0 - 3) - 1 < 0) || (2 * (t_act0 + runs * R + 3) >= hi2);
        // two adjacent channels per work item: every LDS access moves a channel PAIR (the 2-byte-per-lane version was
        // bound by LDS instruction issue, not by the FIR math)
        struct alignas(2 * sizeof(T)) Pair { T a, b; };
        const int CP = C >> 1;
        const int npairs = CP * runs;
        for (int it = tid; it < npairs; it += 256) {
            const int run = it / CP, cp = it - run * CP;
            const int c = 2 * cp;
            const int ml = run * R;
            const float s0 = FAST ? 0.15915494309189535f : 1.f;
            const aa_f2 al = aa_f2{p.alpha_s[c] * s0, p.alpha_s[c + 1] * s0};
            const aa_f2 ib = aa_f2{p.inv_beta[c], p.inv_beta[c + 1]};
            aa_f2 xv[R + 10], o[R];
#pragma unroll
            for (int j = 0; j < R + 10; ++j) {
                const Pair pr = *reinterpret_cast<const Pair*>(XS + (ml + j) * C + c);
                xv[j] = aa_f2{to_f32(pr.a), to_f32(pr.b)};
            }
            const int mp = t_act0 + ml;
            if (p.dbg & 1) {
#pragma unroll
                for (int r = 0; r < R; ++r) o[r] = xv[r + 5];
            } else if (edge) {
                aa_run<R, FAST, true>(xv, o, tp, al, ib, mp, 0, hi2);
            } else {
                aa_run<R, FAST, false>(xv, o, tp, al, ib, mp, 0, hi2);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = ml + r;
                if (row < p.rows_act) {
                    const int t = mp + r;
                    const bool in = t >= 0 && t < p.T;
                    Pair pr;
                    pr.a = from_f32<T>(in ? o[r].x : 0.f); pr.b = from_f32<T>(in ? o[r].y : 0.f);
                    *reinterpret_cast<Pair*>(AS + row * S + c) = pr;
                }
            }
        }
    }
    __syncthreads();
    // ---- 3. implicit GEMM: out[m][n] = sum_{tap,ci} AS[m + tap*dil][ci] * W[n][tap*C + ci] -----------------
    const int wave = tid >> 6, lane = tid & 63, lr = lane & 31, hi = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    {
        // Weights go through LDS once per workgroup (they used to be fetched from L2 by each of the four waves, with the
        // L2 latency exposed in front of every MFMA group): 64-deep K chunks [N_pad][64 (+pad)], double buffered in the
        // region that held the raw input tile (dead after the AA pass), next chunk's global loads issued before the MFMAs.
        constexpr int KCH = 64;                                   // K elements per chunk
        constexpr int NP = TN * 32;                               // padded N
        constexpr int LDW = KCH + (sizeof(T) == 4 ? 1 : 8);
        constexpr int WV = NP * KCH / VEC;                        // 16-byte vectors per chunk
        constexpr int WPT = (WV + 255) / 256;                     // per thread
        constexpr int STEPS = KCH / (2 * KP);                     // MFMA k-steps per chunk
        T* Wb = XS + p.w_off;                                     // 2 x NP x LDW ; w_off = 0 aliases the dead raw-input tile
        const T* wp = (const T*)p.w;
        const int nch = p.Kpad / KCH;
        uint4 wreg[WPT];
        auto wload = [&](int c) {
#pragma unroll
            for (int q = 0; q < WPT; ++q) {
                const int v = tid + q * 256;
                const int n = v / (KCH / VEC), kv = v - n * (KCH / VEC);
                const int k2 = c * KCH + kv * VEC;
                uint4 raw = make_uint4(0, 0, 0, 0);
                if (v < WV && n < C && k2 < p.K) raw = *reinterpret_cast<const uint4*>(wp + (long)n * p.K + k2);
                wreg[q] = raw;
            }
        };
        auto wstore = [&](int buf) {
#pragma unroll
            for (int q = 0; q < WPT; ++q) {
                const int v = tid + q * 256;
                if (v < WV) {
                    const int n = v / (KCH / VEC), kv = v - n * (KCH / VEC);
                    T* d = Wb + (buf * NP + n) * LDW + kv * VEC;
                    if constexpr (sizeof(T) == 4) {
                        float* f = reinterpret_cast<float*>(d);
                        f[0] = __uint_as_float(wreg[q].x); f[1] = __uint_as_float(wreg[q].y);
                        f[2] = __uint_as_float(wreg[q].z); f[3] = __uint_as_float(wreg[q].w);
                    } else {
                        *reinterpret_cast<uint4*>(d) = wreg[q];
                    }
                }
            }
        };
        int kk = hi * KP;                     // this lane's K offset inside the current MFMA k-step
        int tap = kk / C, ci = kk - tap * C;
        if (!(p.dbg & 2)) {
            wload(0);
            wstore(0);
            __syncthreads();
            for (int c = 0; c < nch; ++c) {
                if (c + 1 < nch) wload(c + 1);
                const T* Wc = Wb + (c & 1) * NP * LDW;
#pragma unroll
                for (int ks = 0; ks < STEPS; ++ks) {
                    const int tapc = tap < p.k ? tap : p.k - 1;        // padded K tail: weights are zero there
                    typename MF::Frag a[TM], bf[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        a[i] = *reinterpret_cast<const typename MF::Frag*>(AS + (wave * WM + i * 32 + lr + tapc * p.dil) * S + ci);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        bf[j] = *reinterpret_cast<const typename MF::Frag*>(Wc + (j * 32 + lr) * LDW + ks * 2 * KP + hi * KP);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma(a[i], bf[j], acc[i][j]);
                    ci += 2 * KP;
                    while (ci >= C) { ci -= C; ++tap; }
                }
                if (c + 1 < nch) {
                    wstore((c + 1) & 1);      // the other buffer: last read in iteration c-1, fenced by that barrier
                    __syncthreads();
                }
            }
        }
    }
    __syncthreads();                          // every wave is done reading AS before OUT overwrites it
    // ---- 4. accumulators -> LDS -> coalesced epilogue -------------------------------------------------------
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = j * 32 + lr;
        if (n < C) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    OUT[(wave * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * C + n] = acc[i][j][r];
        }
    }
    __syncthreads();
    {
        const int cvn = C / VEC, nvec = BM * cvn;
        T* ob = (T*)p.out + (long)b * p.T * C;
        const T* rb = p.res ? (const T*)p.res + (long)b * p.T * C : nullptr;
        for (int v = tid; v < nvec; v += 256) {
            const int row = v / cvn, cv = v - row * cvn;
            const int t = m0 + row;
            if (t >= p.T || (p.dbg & 4)) continue;
            const long gi = (long)t * C + cv * VEC;
            float o[VEC];
#pragma unroll
            for (int e4 = 0; e4 < VEC; e4 += 4) {
                const float4 ov4 = *reinterpret_cast<const float4*>(OUT + row * C + cv * VEC + e4);   // C % 8 == 0: 16-byte aligned
                o[e4] = ov4.x; o[e4 + 1] = ov4.y; o[e4 + 2] = ov4.z; o[e4 + 3] = ov4.w;
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] += p.bias[cv * VEC + e];
            if (rb) {
                const uint4 raw = *reinterpret_cast<const uint4*>(rb + gi);
                const T* rv = reinterpret_cast<const T*>(&raw);
#pragma unroll
                for (int e = 0; e < VEC; ++e) o[e] += to_f32(rv[e]);
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] *= p.alpha;
            if (p.accumulate) {
                const uint4 raw = *reinterpret_cast<const uint4*>(ob + gi);
                const T* pv = reinterpret_cast<const T*>(&raw);
#pragma unroll
                for (int e = 0; e < VEC; ++e) o[e] += to_f32(pv[e]);
            }
            T ov[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) ov[e] = from_f32<T>(o[e]);
            *reinterpret_cast<uint4*>(ob + gi) = *reinterpret_cast<const uint4*>(ov);
        }
    }
}

template <typename T>
static void launch_t(const AAConv& q, hipStream_t s) {
    constexpr int KP = Mfma<T>::KP;
    constexpr int VEC = 16 / (int)sizeof(T);
    MI_REQUIRE(q.C % VEC == 0 && q.C % 8 == 0 && q.C <= 96, "aa_conv: C must be a multiple of 8 and <= 96");
    AAConvDev d;
    d.x = q.x; d.w = q.w; d.bias = q.bias; d.alpha_s = q.snake_alpha; d.inv_beta = q.snake_inv_beta; d.out = q.out; d.res = q.res;
    d.T = q.T; d.C = q.C; d.k = q.k; d.dil = q.dil; d.K = q.k * q.C;
    d.Kpad = (d.K + 63) / 64 * 64;                         // whole 64-deep weight chunks (zero filled past K)
    d.halo = (q.k * q.dil - q.dil) / 2;
    d.S = ((q.C / 8) & 1) ? q.C : q.C + 8;             // S/8 odd => conflict-free ds_read_b128 fragment rows
    if (sizeof(T) == 4) d.S = (q.C % 2 == 0) ? q.C + 1 : q.C;   // fp32 fragments are ds_read_b32: odd dword stride
    d.alpha = q.alpha; d.accumulate = q.accumulate;
    { const char* e = std::getenv("MI355TTS_AACONV_DBG"); d.dbg = e ? std::atoi(e) : 0; }
    const int BM = q.C <= 48 ? 256 : 128;
    d.rows_act = BM + 2 * d.halo;
    d.rows_x = (d.rows_act + 15) / 16 * 16 + 10;
    const int TN = (q.C + 31) / 32;
    size_t lds = (size_t)d.rows_x * q.C * sizeof(T) + (size_t)d.rows_act * d.S * sizeof(T);
    lds = std::max(lds, (size_t)BM * q.C * 4);
    lds = (lds + 15) / 16 * 16;
    {
        const size_t ldw = 64 + (sizeof(T) == 4 ? 1 : 8);
        const size_t ring = (size_t)2 * TN * 32 * ldw;                       // elements
        if (ring <= (size_t)d.rows_x * q.C) d.w_off = 0;
        else {                                                               // tiny channel counts: ring gets its own region
            const size_t off = ((size_t)d.rows_x * q.C + (size_t)d.rows_act * d.S + 7) / 8 * 8;
            d.w_off = (int)off;
            lds = std::max(lds, ((off + ring) * sizeof(T) + 15) / 16 * 16);
        }
    }
    // Run-to-run identity.  The 16-bit kernels fit two workgroups per CU (<= 66 KB of LDS) and that is the default.  Rounds 1-2
    // were not bit-reproducible in that mode: one sample per channel was read through `v_pk_fma_f32 ... op_sel:[0,1,0]`, and
    // next to a co-resident workgroup's MFMAs that read came back one ulp off in a handful of the ~10^7 outputs of a launch
    // (profiles/r3/aa_conv_opsel_rootcause.txt / _ab.txt).  aa_math.h no longer produces that encoding (channel pairs, every VGPR
    // source a whole aligned pair): 0 of 119 runs differ with two workgroups per CU, and tests/test_gpu_bigvgan.py asserts
    // array_equal across batch items and runs in the default mode.  The one-workgroup-per-CU policy stays as a diagnostic:
    // MI355TTS_AACONV_LDS_MIN=83968 (+19 % forward time).
    {
        if (g_aa_lds_min < 0) { const char* e = std::getenv("MI355TTS_AACONV_LDS_MIN"); g_aa_lds_min = e ? std::atol(e) : 0; }
        if (sizeof(T) == 2) lds = std::max(lds, (size_t)g_aa_lds_min);
    }
    MI_REQUIRE(lds <= 160 * 1024, "aa_conv: tile does not fit LDS");
    dim3 grid((q.T + BM - 1) / BM, q.B);
    const double E = (double)q.B * q.T * q.C * sizeof(T);
    ProfScope ps(FAM_CONV_GEMM, s, E * (2.0 + (q.res ? 1.0 : 0.0) + (q.accumulate ? 1.0 : 0.0)) + (double)d.K * q.C * sizeof(T),
                 2.0 * q.B * (double)q.T * q.C * d.K + 60.0 * q.B * (double)q.T * q.C);
#define LAUNCH(BMv, TNv)                                                                                         \
    do {                                                                                                         \
        auto kfn = aa_conv_kernel<T, BMv, TNv>;                                                                  \
        /* the opt-in is per DEVICE and cheap: set whenever needed (a process-wide flag would skip device 1) */ \
        if (lds > 64 * 1024) MI_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        prof_set_kernel("aa_conv_kernel<T, " #BMv ", " #TNv ">", type_label<T>());                                \
        hipLaunchKernelGGL(kfn, grid, dim3(256), lds, s, d);                                                     \
    } while (0)
    if (BM == 256) { if (TN == 1) LAUNCH(256, 1); else LAUNCH(256, 2); }
    else { if (TN == 1) LAUNCH(128, 1); else if (TN == 2) LAUNCH(128, 2); else LAUNCH(128, 3); }
#undef LAUNCH
    MI_HIP(hipGetLastError());
}

void launch_aa_conv(const AAConv& q, hipStream_t s) {
    static bool uploaded[64] = {false};
    int dev = 0;
    MI_HIP(hipGetDevice(&dev));
    if (!uploaded[dev & 63]) {
        MI_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_h_fused), aa_filter_host(), sizeof(float) * 12));
        uploaded[dev & 63] = true;
    }
    if (q.dtype == MI_F32) launch_t<float>(q, s);
    else if (q.dtype == MI_F16) launch_t<f16>(q, s);
    else launch_t<bf16>(q, s);
}

}  // namespace mi
