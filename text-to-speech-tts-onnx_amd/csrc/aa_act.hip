// aa_act.hip — fused anti-aliased SnakeBeta activation (one reference Activation1d) for gfx950.
//
// Reference: Activation1d.forward (BigVGAN/modeling_modified/act.py:25-29) =
//   UpSample1d (resample.py:30-34: zero-pad 5, depthwise conv_transpose stride 2 with the 12-tap
//   kaiser-sinc filter, x2, crop 15/15) -> SnakeBeta -> DownSample1d (filter.py:94-98: zero-pad 5/6,
//   depthwise 12-tap conv stride 2).  The pad-15 "post" variant (bigvgan.py:370,381-382) returns
//   T+30 samples.
//
// Polyphase form used here (derivation in DESIGN.md §K-AA), centred index i' / output m':
//   U[2a]   = 2 * sum_e h[2e+1] * x[a+2-e]        U[2a+1] = 2 * sum_e h[2e] * x[a+3-e]   (e = 0..5)
//   S[i']   = U + inv_beta_c * sin^2(alpha_c * U)   for -ext <= i' < 2T+ext, else 0
//   y[m']   = sum_t h[t] * S[2m' - 5 + t]           for -shift <= m' < T+shift
// with (shift, ext) = (0, 0) in the residual blocks and (15, 20) for activation_post.
// x is zero outside [0, T).  Everything is computed in fp32 whatever the storage type.
//
// Memory plan: HBM-bound.  A workgroup stages a (TT+10) x CT tile of channels-last activations into
// LDS with 16-byte coalesced loads (the tile is one contiguous HBM span when CT == C), each thread
// then slides an R-long run down the time axis in registers (18 inputs -> 8 outputs), and the
// result goes back through LDS so the HBM store is again 16-byte coalesced.
#include "common.h"
#include "aa_math.h"
#include <algorithm>
#include <cstdlib>

namespace mi {

static float g_taps[12];
static bool g_taps_init = false;
__constant__ float c_h[12];

static double bessel_i0(double x) {
    double sum = 1.0, term = 1.0;
    for (int k = 1; k < 64; ++k) {
        term *= (x / (2.0 * k)) * (x / (2.0 * k));
        sum += term;
        if (term < 1e-18 * sum) break;
    }
    return sum;
}

// kaiser_sinc_filter1d(cutoff 0.25, half_width 0.3, kernel 12) — filter.py:30-62
const float* aa_filter_host() {
    if (!g_taps_init) {
        const int K = 12, half = 6;
        const double cutoff = 0.25, half_width = 0.3;
        const double delta_f = 4 * half_width;
        const double A = 2.285 * (half - 1) * M_PI * delta_f + 7.95;
        double beta;
        if (A > 50.0) beta = 0.1102 * (A - 8.7);
        else if (A >= 21.0) beta = 0.5842 * std::pow(A - 21.0, 0.4) + 0.07886 * (A - 21.0);
        else beta = 0.0;
        double f[12], sum = 0.0;
        for (int n = 0; n < K; ++n) {
            const double r = (n - (K - 1) / 2.0) / ((K - 1) / 2.0);
            const double win = bessel_i0(beta * std::sqrt(std::max(0.0, 1.0 - r * r))) / bessel_i0(beta);
            const double t = (n - half) + 0.5;
            const double a = 2 * cutoff * t;
            const double sinc = a == 0.0 ? 1.0 : std::sin(M_PI * a) / (M_PI * a);
            f[n] = 2 * cutoff * win * sinc;
            sum += f[n];
        }
        for (int n = 0; n < K; ++n) g_taps[n] = (float)(f[n] / sum);
        g_taps_init = true;
    }
    return g_taps;
}

template <typename T, int R>
__global__ __launch_bounds__(256) void aa_act_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                     const float* __restrict__ alpha,
                                                     const float* __restrict__ inv_beta, int Tlen, int C, int CT,
                                                     int TT, int shift, int ext) {
    constexpr int VEC = 16 / (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T* xs = reinterpret_cast<T*>(lds_raw);                // (TT+10) x CT  raw input (16-byte copies, no bank conflicts)
    T* ys = xs + (TT + 10) * CT;                          // TT x CT

    const int tid = threadIdx.x;
    const int o0 = blockIdx.x * TT;           // first output row of the tile
    const int c0 = blockIdx.y * CT;
    const int b = blockIdx.z;
    const int Tout = Tlen + 2 * shift;
    const int mp0 = o0 - shift;               // centred index m' of the first output row
    const T* xb = x + (long)b * Tlen * C;
    T* yb = y + (long)b * Tout * C;

    // ---- stage x rows [mp0-5, mp0+TT+5) ------------------------------------------------------
    const int cvn = CT / VEC;
    const int nvec = (TT + 10) * cvn;
    for (int v = tid; v < nvec; v += 256) {
        const int row = v / cvn, cv = v - row * cvn;
        const int t = mp0 - 5 + row;
        uint4 raw = make_uint4(0, 0, 0, 0);
        if (t >= 0 && t < Tlen) raw = *reinterpret_cast<const uint4*>(xb + (long)t * C + c0 + cv * VEC);
        *reinterpret_cast<uint4*>(xs + row * CT + cv * VEC) = raw;
    }
    __syncthreads();

    // ---- sliding-window compute (aa_math.h) ------------------------------------------------------
    constexpr bool FAST = sizeof(T) == 2;           // 16-bit storage: v_sin_f32; fp32 parity path: libm sinf
    const AATaps tp = aa_make_taps(c_h);
    const int lo = -ext, hi = 2 * Tlen + ext;
    const bool edge = (2 * (mp0 - 3) - 1 < lo) || (2 * (mp0 + TT + 3) >= hi);      // block-uniform
    // two adjacent channels per work item (CT is even): one LDS access moves a channel pair, and the packed FMAs of
    // aa_math.h carry the pair in their two halves
    struct alignas(2 * sizeof(T)) Pair { T a, b; };
    const int CP = CT >> 1;
    const int nitems = CP * (TT / R);
    for (int it = tid; it < nitems; it += 256) {
        const int run = it / CP, c = 2 * (it - run * CP);
        const int ml = run * R;               // local output row
        const float s0 = FAST ? 0.15915494309189535f : 1.f;
        const aa_f2 al = aa_f2{alpha[c0 + c] * s0, alpha[c0 + c + 1] * s0};
        const aa_f2 ib = aa_f2{inv_beta[c0 + c], inv_beta[c0 + c + 1]};
        aa_f2 xv[R + 10], acc[R];
#pragma unroll
        for (int j = 0; j < R + 10; ++j) {
            const Pair pr = *reinterpret_cast<const Pair*>(xs + (ml + j) * CT + c);
            xv[j] = aa_f2{to_f32(pr.a), to_f32(pr.b)};
        }
        if (edge) aa_run<R, FAST, true>(xv, acc, tp, al, ib, mp0 + ml, lo, hi);
        else aa_run<R, FAST, false>(xv, acc, tp, al, ib, mp0 + ml, lo, hi);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            Pair pr;
            pr.a = from_f32<T>(acc[r].x); pr.b = from_f32<T>(acc[r].y);
            *reinterpret_cast<Pair*>(ys + (ml + r) * CT + c) = pr;
        }
    }
    __syncthreads();

    // ---- coalesced store ----------------------------------------------------------------------
    const int nout = TT * cvn;
    for (int v = tid; v < nout; v += 256) {
        const int row = v / cvn, cv = v - row * cvn;
        const int o = o0 + row;
        if (o < Tout)
            *reinterpret_cast<uint4*>(yb + (long)o * C + c0 + cv * VEC) =
                *reinterpret_cast<const uint4*>(ys + row * CT + cv * VEC);
    }
}

// -----------------------------------------------------------------------------------------------------------------------------
// Round 6: the same tile, software-pipelined inside a persistent workgroup (16-bit storage).
// The kernel above is one tile per workgroup: load -> barrier -> compute -> barrier -> store.  A launch is only two or three rounds
// of workgroups, all started together, so the whole chip loads at the same time, computes at the same time and stores at the
// same time: its time is the SUM of the HBM phase and the VALU phase (C = 384, T = 8192, B = 8: 20 us of traffic + 28 us of VALU
// issue = the 48 us measured), not their maximum.  Here a workgroup walks tiles: the input of tile i + 1 arrives by LDS-DMA
// (buffer_load ... lds: no VGPR staging, rows outside [0, T) zero-filled by the descriptor's range check) while tile i is
// computed, and the stores of tile i leave while tile i + 1 is computed.  Same arithmetic per element: bit-identical outputs.
// -----------------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void aa_lds_void;

template <typename T, int R>
__global__ __launch_bounds__(256) void aa_act_pipe_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ alpha,
                                                          const float* __restrict__ inv_beta, int Tlen, int C, int CT, int TT,
                                                          int shift, int ext, int ntt, int nct, int ntiles, int xsp) {
    constexpr int VEC = 16 / (int)sizeof(T);
    static_assert(sizeof(T) == 2, "16-bit storage");
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
    T* const xs0 = reinterpret_cast<T*>(lds_raw);         // two input buffers of xsp elements ((TT + 10) x CT, rounded up to whole 1 KB DMA writes)
    T* const ys = xs0 + 2 * xsp;                          // TT x CT
    const int tid = threadIdx.x, lane = tid & 63;
    const int Tout = Tlen + 2 * shift;
    const int cvn = CT / VEC, nvec = (TT + 10) * cvn, nout = TT * cvn;
    const AATaps tp = aa_make_taps(c_h);
    const int lo = -ext, hi = 2 * Tlen + ext;
    struct alignas(2 * sizeof(T)) Pair { T a, b; };
    const int CP = CT >> 1;
    const int nitems = CP * (TT / R);

    auto decode = [&](int t, int& o0, int& c0, int& b) {   // time tiles fastest: neighbours share their halo rows in L2
        const int ti = t % ntt, r = t / ntt;
        const int ci = r % nct;
        b = r / nct; o0 = ti * TT; c0 = ci * CT;
    };
    // input rows [o0 - shift - 5, + TT + 10) x channels [c0, c0 + CT) of batch item b -> xs[buf], 64 lanes x 16 bytes per instruction
    auto dma = [&](int t, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)
        int o0, c0, b;
        decode(t, o0, c0, b);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long)b * Tlen * C), 0, (int)((long)Tlen * C * (long)sizeof(T)), 0x00020000);
        T* dst = xs0 + buf * xsp;
        for (int v = tid; v < xsp / VEC; v += 256) {
            const int row = v / cvn, cv = v - row * cvn;
            // rows before 0: a negative offset = a huge unsigned one, zero-filled like rows >= Tlen; lanes past the tile: into the pad
            const int off = v < nvec ? (int)(((long)(o0 - shift - 5 + row) * C + c0 + cv * VEC) * (long)sizeof(T)) : 0x7fffff00;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (aa_lds_void*)(dst + (v - lane) * VEC), 16, off, 0, 0, 0);
        }
#endif
    };
    auto store = [&](int t) {
        int o0, c0, b;
        decode(t, o0, c0, b);
        T* yb = y + (long)b * Tout * C;
        for (int v = tid; v < nout; v += 256) {
            const int row = v / cvn, cv = v - row * cvn;
            const int o = o0 + row;
            if (o < Tout)
                *reinterpret_cast<uint4*>(yb + (long)o * C + c0 + cv * VEC) = *reinterpret_cast<const uint4*>(ys + row * CT + cv * VEC);
        }
    };

    int t = blockIdx.x;
    if (t >= ntiles) return;
    dma(t, 0);
    int buf = 0, tprev = -1;
    for (; t < ntiles; t += gridDim.x) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this tile's rows have landed (this wave's share); the stores before them too
        __syncthreads();
        if (tprev >= 0) store(tprev);                         // the previous tile leaves while this one is computed
        __syncthreads();                                      // ys is free again
        if (t + (int)gridDim.x < ntiles) dma(t + gridDim.x, buf ^ 1);      // that buffer's tile was computed before the first barrier
        int o0, c0, b;
        decode(t, o0, c0, b);
        const int mp0 = o0 - shift;
        const bool edge = (2 * (mp0 - 3) - 1 < lo) || (2 * (mp0 + TT + 3) >= hi);      // workgroup-uniform
        const T* xs = xs0 + buf * xsp;
        for (int it = tid; it < nitems; it += 256) {
            const int run = it / CP, c = 2 * (it - run * CP);
            const int ml = run * R;
            const float s0 = 0.15915494309189535f;
            const aa_f2 al = aa_f2{alpha[c0 + c] * s0, alpha[c0 + c + 1] * s0};
            const aa_f2 ib = aa_f2{inv_beta[c0 + c], inv_beta[c0 + c + 1]};
            if constexpr (R > 8) {
                // long runs: inputs fetched and outputs stored as the window moves (aa_math.h aa_run_stream): six live inputs whatever R
                auto ld = [&](int j) __attribute__((always_inline)) {
                    const Pair pr = *reinterpret_cast<const Pair*>(xs + (ml + j) * CT + c);
                    return aa_f2{to_f32(pr.a), to_f32(pr.b)};
                };
                auto st = [&](int r, aa_f2 v) __attribute__((always_inline)) {
                    Pair pr;
                    pr.a = from_f32<T>(v.x); pr.b = from_f32<T>(v.y);
                    *reinterpret_cast<Pair*>(ys + (ml + r) * CT + c) = pr;
                };
                if (edge) aa_run_stream<R, true, true>(ld, st, tp, al, ib, mp0 + ml, lo, hi);
                else aa_run_stream<R, true, false>(ld, st, tp, al, ib, mp0 + ml, lo, hi);
                continue;
            }
            aa_f2 xv[R + 10], acc[R];
#pragma unroll
            for (int j = 0; j < R + 10; ++j) {
                const Pair pr = *reinterpret_cast<const Pair*>(xs + (ml + j) * CT + c);
                xv[j] = aa_f2{to_f32(pr.a), to_f32(pr.b)};
            }
            if (edge) aa_run<R, true, true>(xv, acc, tp, al, ib, mp0 + ml, lo, hi);
            else aa_run<R, true, false>(xv, acc, tp, al, ib, mp0 + ml, lo, hi);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                Pair pr;
                pr.a = from_f32<T>(acc[r].x); pr.b = from_f32<T>(acc[r].y);
                *reinterpret_cast<Pair*>(ys + (ml + r) * CT + c) = pr;
            }
        }
        tprev = t; buf ^= 1;
    }
    __syncthreads();
    store(tprev);
}

template <typename T>
static void launch_t(const AAAct& p, hipStream_t s) {
    // run length per work item.  16-bit storage: 8 (152 VGPRs = three waves per SIMD; with 16 the kernel holds 204 = two, and a
    // single wave issues at most one VALU instruction per 8 cycles, tools/ubench/mfma_valu_coexec.hip: 44.5 -> 42.6 us per launch
    // although a run of 8 recomputes more of the up-sampled halo); fp32: 16
    constexpr int R = sizeof(T) == 2 ? 8 : 16;
    constexpr int VEC = 16 / (int)sizeof(T);
    MI_REQUIRE(p.C % VEC == 0, "aa_act: C must be a multiple of the 16-byte vector");
    int CT = p.C;
    if (p.C > 96) {
        CT = 0;
        for (int cand : {64, 48, 32, 24, 16, 8})
            if (p.C % cand == 0 && cand % VEC == 0) { CT = cand; break; }
        MI_REQUIRE(CT > 0, "aa_act: unsupported channel count");
    }
    static int tt_elems = 0;                                // tile size in elements (rows x channels); tuning: MI355TTS_AA_TILE
    if (!tt_elems) { const char* e = std::getenv("MI355TTS_AA_TILE"); tt_elems = e ? std::atoi(e) : 8192; }
    int TT = (tt_elems / CT) / R * R;
    if (TT < R) TT = R;
    if (TT > 512) TT = 512;
    const int shift = p.post ? 15 : 0, ext = p.post ? 20 : 0;
    const int Tout = p.T + 2 * shift;
    dim3 grid((Tout + TT - 1) / TT, p.C / CT, p.B);
    const size_t lds = ((size_t)(TT + 10) * CT + (size_t)TT * CT) * sizeof(T);
    const double bytes = (double)p.B * p.C * ((double)p.T + Tout) * sizeof(T);
    const double flops = (double)p.B * p.C * Tout * 60.0;
    ProfScope ps(FAM_AA, s, bytes, flops);
    if constexpr (sizeof(T) == 2) {
        static const bool pipe = [] { const char* e = std::getenv("MI355TTS_AA_PIPE"); return !(e && e[0] == '0'); }();
        const long in_bytes = (long)p.T * p.C * (long)sizeof(T);
        // outputs per work item of the pipelined kernel: 16 in the streaming form (aa_run_stream: 135 VGPRs; 21 / 16 instead of 13 / 8
        // up-sampler + snake evaluations per output: 44.2 -> 41.4 us per launch, bit-identical; MI355TTS_AA_R=8: the A/B switch)
        static const int run_len = [] { const char* e = std::getenv("MI355TTS_AA_R"); return e ? std::atoi(e) : 16; }();
        const int RP = run_len == 8 ? 8 : 16;
        if (pipe && in_bytes < 0x7fff0000L) {
            // persistent workgroups walking tiles, input by LDS-DMA one tile ahead (aa_act_pipe_kernel)
            TT = (tt_elems / CT) / RP * RP;
            if (TT < RP) TT = RP;
            if (TT > 512) TT = 512;
            const int ntt = (Tout + TT - 1) / TT, nct = p.C / CT;
            const long ntiles = (long)ntt * nct * p.B;
            const int nvec = (TT + 10) * CT / VEC;
            const int xsp = (nvec + 63) / 64 * 64 * VEC;
            const size_t lds2 = ((size_t)2 * xsp + (size_t)TT * CT) * sizeof(T);
            int dev = 0, cus = 256;
            MI_HIP(hipGetDevice(&dev));
            {
                static int cu_count[16] = {0};
                if (!cu_count[dev & 15]) { hipDeviceProp_t pr; MI_HIP(hipGetDeviceProperties(&pr, dev)); cu_count[dev & 15] = pr.multiProcessorCount; }
                cus = cu_count[dev & 15];
            }
            const int per_cu = (int)std::min<size_t>(3, (size_t)(160 * 1024) / lds2);
            if (per_cu >= 1 && lds2 <= 64 * 1024 && ntiles < 0x7fffffffL) {
                const int grid_p = (int)std::min<long>(ntiles, (long)cus * per_cu);
                prof_set_kernel("aa_act_pipe_kernel<T>", type_label<T>());
                if (RP == 16)
                    hipLaunchKernelGGL((aa_act_pipe_kernel<T, 16>), dim3(grid_p), dim3(256), lds2, s, (const T*)p.x, (T*)p.y, p.alpha, p.inv_beta,
                                       p.T, p.C, CT, TT, shift, ext, ntt, nct, (int)ntiles, xsp);
                else
                    hipLaunchKernelGGL((aa_act_pipe_kernel<T, 8>), dim3(grid_p), dim3(256), lds2, s, (const T*)p.x, (T*)p.y, p.alpha, p.inv_beta,
                                       p.T, p.C, CT, TT, shift, ext, ntt, nct, (int)ntiles, xsp);
                MI_HIP(hipGetLastError());
                return;
            }
        }
    }
    prof_set_kernel("aa_act_kernel<T>", type_label<T>());
    hipLaunchKernelGGL((aa_act_kernel<T, R>), grid, dim3(256), lds, s, (const T*)p.x, (T*)p.y, p.alpha, p.inv_beta,
                       p.T, p.C, CT, TT, shift, ext);
    MI_HIP(hipGetLastError());
}

void launch_aa_act(const AAAct& p, hipStream_t s) {
    static bool uploaded[64] = {false};
    int dev = 0;
    MI_HIP(hipGetDevice(&dev));
    if (!uploaded[dev & 63]) {
        MI_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_h), aa_filter_host(), sizeof(float) * 12));
        uploaded[dev & 63] = true;
    }
    if (p.dtype == MI_F32) launch_t<float>(p, s);
    else if (p.dtype == MI_F16) launch_t<f16>(p, s);
    else launch_t<bf16>(p, s);
}

}  // namespace mi
