// gconv16.hip — the DiT's grouped position convolution for the 16-bit engines: Conv1d(1024, 1024, k = 31, groups = 16, padding = 15)
// -> Mish, twice (F5_TTS/modeling_modified/F5/modules.py:167-190), f16 / bf16 operands, fp32 accumulation.
//
// The shape (64 channels per group in and out, many taps) used to run on the register-staged 128 x 64 tile of conv_gemm_kernel
// (gemm_conv.hip): every tap re-fetches its (shifted) input rows from L2 and its weights per row tile — 180 us per launch at eight
// utterances (406 TFLOP/s).  This is the 16-bit sibling of gconv_pairs2_kernel (gconv_pairs.hip), one plane instead of two:
//   * a workgroup owns BM = 192 output rows x the 64 output channels of one (batch item, group); rows [m0 - pad, m0 + BM - pad +
//     taps - 1) x 64 input channels are staged ONCE in LDS (row stride 72: conflict-free 16-byte fragment reads) and tap t's A
//     fragments are the same plane read t rows further down;
//   * the weights come from an image built once per engine ([group][tap][co][72], pad columns zero: gconv16_build_weights) and
//     stream by LDS-DMA, a tap PAIR (18 contiguous 1 KB pieces) per step into one of two slot pairs;
//   * eight waves = two tap-parity groups (even taps | odd taps) of 2 (rows) x 2 (channels) waves; one barrier per tap pair;
//   * the odd group parks its partial sums in LDS, the even group adds its own to them in place (fixed order: even + odd), and all
//     512 threads run the coalesced epilogue (+ bias, Mish, + residual) from that one fp32 image.  69 KB of LDS: two workgroups per CU.
#include "common.h"
#include "gemm_epilogue.h"
#include "mfma.h"
#include <algorithm>
#include <atomic>
#include <cstdlib>

namespace mi {

constexpr int GC16_C = 64, GC16_S = GC16_C + 8, GC16_WTAP = GC16_C * GC16_S;     // elements of one (group, tap) weight image

size_t gconv16_image_bytes(int G, int taps) { return (size_t)G * ((taps + 1) / 2 * 2) * GC16_WTAP * 2; }

template <typename T>
__global__ __launch_bounds__(256) void gconv16_image_kernel(const T* __restrict__ w, T* __restrict__ img, int G, int taps, int tp) {
    // one thread per (g, tap, co, 8 input channels); w is [g][co][tap][ci]
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long n = (long)G * taps * GC16_C * 8;
    if (i >= n) return;
    const int c8 = (int)(i & 7), co = (int)((i >> 3) & 63);
    const long gt = i >> 9;
    const int t = (int)(gt % taps), g = (int)(gt / taps);
    *reinterpret_cast<uint4*>(img + ((long)g * tp + t) * GC16_WTAP + co * GC16_S + c8 * 8) =
        *reinterpret_cast<const uint4*>(w + (((long)g * GC16_C + co) * taps + t) * GC16_C + c8 * 8);
}

// img: gconv16_image_bytes(G, taps) bytes; w: [G][64 co][taps][64 ci] in the engine's 16-bit type (the layout launch_conv_gemm takes)
void gconv16_build_weights(const void* w, int dtype, void* img, int G, int taps, hipStream_t s) {
    MI_REQUIRE(dtype == MI_F16 || dtype == MI_BF16, "gconv16: 16-bit weights");
    const int tp = (taps + 1) / 2 * 2;
    MI_HIP(hipMemsetAsync(img, 0, gconv16_image_bytes(G, taps), s));          // pad columns and the pad tap of an odd count
    const long n = (long)G * taps * GC16_C * 8;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == MI_F16) hipLaunchKernelGGL(gconv16_image_kernel<f16>, grid, dim3(256), 0, s, (const f16*)w, (f16*)img, G, taps, tp);
    else hipLaunchKernelGGL(gconv16_image_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)w, (bf16*)img, G, taps, tp);
    MI_HIP(hipGetLastError());
}

struct GConv16Dev {
    const void* x; const void* img; const float* bias; void* out; const void* res;
    int T_in, M, taps, pad, act;
    long x_bstride, x_rstride, x_goff, out_bstride, out_rstride;
};

// Mish(v) = v tanh(log(1 + e^v)) = v n / (n + 2), n = e^v (e^v + 2) (as gconv_pairs.hip: no log1pf / tanhf expansions)
__device__ __forceinline__ float gc16_act(float v, int act) {
    if (act == ACT_MISH) {
        const float e = expf(fminf(v, 20.f));
        const float n = e * (e + 2.f);
        return v > 20.f ? v : v * (n / (n + 2.f));
    }
    return act_apply(v, act);
}

template <typename T, typename TO, int BM>
__global__ __launch_bounds__(512, 2) void gconv16_kernel(const GConv16Dev p) {
    using MF = Mfma<T>;
    using FR = typename MF::Frag;
    constexpr int S = GC16_S;
    constexpr int WM = BM / 2, TM = WM / 32;                        // per tap-parity group: 2 (rows) x 2 (channels) waves, WM x 32 per wave
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    const int rows_a = BM + p.taps - 1;
    const int a_el = ((rows_a * S + 511) / 512) * 512;              // plane size rounded to 1 KB: the weight slots stay 1 KB-aligned
    T* AP = reinterpret_cast<T*>(smem_raw);
    T* WS = AP + a_el;                                              // four tap slots: slot = 2 * (pair & 1) + parity
    float* OUT = reinterpret_cast<float*>(smem_raw);                // BM x 64 fp32, aliases everything after the main loop

    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, g = blockIdx.y, b = blockIdx.z;
    const T* xb = (const T*)p.x + (long)b * p.x_bstride + (long)g * p.x_goff;
    const int tp = (p.taps + 1) & ~1, npair = tp >> 1;
    const T* wg = (const T*)p.img + (long)g * tp * GC16_WTAP;

    constexpr int NPIECE = 2 * GC16_WTAP * 2 / 1024;                // a tap pair = 18 pieces of 1 KB, contiguous in HBM and in LDS
    static_assert(NPIECE * 1024 == 2 * GC16_WTAP * 2, "a tap pair is whole 1 KB pieces");
    auto wdma = [&](int pair) {
        const char* src = reinterpret_cast<const char*>(wg + (long)pair * 2 * GC16_WTAP) + lane * 16;
        char* dst = reinterpret_cast<char*>(WS + (pair & 1) * 2 * GC16_WTAP);
        for (int pc = wave; pc < NPIECE; pc += 8)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + pc * 1024), (lds_void*)(dst + pc * 1024), 16, 0, 0);
    };
    wdma(0);
    // ---- activations: rows m0 - pad .. (zeros outside [0, T_in)), once ---------------------------------------------------------
    {
        const int t_base = m0 - p.pad;
        for (int v = tid; v < rows_a * 8; v += 512) {
            const int row = v >> 3, c8 = v & 7;
            const int t = t_base + row;
            uint4 u = make_uint4(0, 0, 0, 0);
            if (t >= 0 && t < p.T_in) u = *reinterpret_cast<const uint4*>(xb + (long)t * p.x_rstride + c8 * 8);
            *reinterpret_cast<uint4*>(AP + row * S + c8 * 8) = u;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int j = 0; j < npair; ++j) {
        if (j + 1 < npair) wdma(j + 1);                             // into the slots of pair j - 1: every wave passed the barrier behind it
        const int t = 2 * j + kg;
        if (t < p.taps) {                                           // (the last pair of an odd tap count has no odd tap)
            const T* b_row = WS + ((j & 1) * 2 + kg) * GC16_WTAP + (wn * 32 + lr) * S + hi * 8;
            const T* a_row = AP + (wm * WM + lr + t) * S + hi * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const FR bf = *reinterpret_cast<const FR*>(b_row + ks * 16);
                FR af[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const FR*>(a_row + i * 32 * S + ks * 16);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = MF::mma(af[i], bf, acc[i]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's pieces of pair j + 1 (a whole pair period to land)
        __syncthreads();                                            // ... and every wave is done with pair j's slots (and, at the end, with the plane)
    }
    // ---- odd taps -> LDS, even taps added in place, coalesced epilogue --------------------------------------------------------
    auto at = [&](int i, int r) -> float& { return OUT[(wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * GC16_C + wn * 32 + lr]; };
    if (kg == 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) at(i, r) = acc[i][r];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) at(i, r) = acc[i][r] + at(i, r);      // even taps + odd taps
    }
    __syncthreads();
    {
        TO* ob = (TO*)p.out + (long)b * p.out_bstride + (long)g * GC16_C;
        const TO* rb = p.res ? (const TO*)p.res + (long)b * p.out_bstride + (long)g * GC16_C : nullptr;
        const float* bias = p.bias ? p.bias + g * GC16_C : nullptr;
        for (int v = tid; v < BM * 16; v += 512) {
            const int row = v >> 4, c4 = v & 15;
            const int m = m0 + row;
            if (m >= p.M) continue;
            float4 o = *reinterpret_cast<const float4*>(OUT + row * GC16_C + c4 * 4);
            if (bias) { const float4 bv = *reinterpret_cast<const float4*>(bias + c4 * 4); o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w; }
            if (p.act != ACT_NONE) { o.x = gc16_act(o.x, p.act); o.y = gc16_act(o.y, p.act); o.z = gc16_act(o.z, p.act); o.w = gc16_act(o.w, p.act); }
            const long gi = (long)m * p.out_rstride + c4 * 4;
            if constexpr (sizeof(TO) == 4) {
                if (rb) { const float4 rv = *reinterpret_cast<const float4*>(rb + gi); o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w; }
                *reinterpret_cast<float4*>(ob + gi) = o;
            } else {
                if (rb) { o.x += to_f32(rb[gi]); o.y += to_f32(rb[gi + 1]); o.z += to_f32(rb[gi + 2]); o.w += to_f32(rb[gi + 3]); }
                struct alignas(8) Q { TO a, b, c, d; };
                Q q; q.a = from_f32<TO>(o.x); q.b = from_f32<TO>(o.y); q.c = from_f32<TO>(o.z); q.d = from_f32<TO>(o.w);
                *reinterpret_cast<Q*>(ob + gi) = q;
            }
        }
    }
}

static std::atomic<bool> g_gconv16{[] { const char* e = std::getenv("MI355TTS_GCONV16"); return !(e && e[0] == '0'); }()};
void gconv16_set_option(long v) { g_gconv16 = v != 0; }

// true: launched.  false: not this kernel's shape (the caller goes on to its other kernels).
bool launch_gconv16(const ConvGemm& p, hipStream_t s) {
    const int odt = p.out_dtype < 0 ? p.dtype : p.out_dtype;
    if (!g_gconv16 || !p.gcp_w || (p.dtype != MI_F16 && p.dtype != MI_BF16) || (odt != p.dtype && odt != MI_F32)) return false;
    if (p.N != GC16_C || p.Cin != GC16_C || p.taps < 8 || p.taps > 127 || p.dil != 1) return false;
    if (p.epi != EPI_PLAIN || p.gate || p.accumulate || p.alpha != 1.f || p.out_planes || p.xp || p.ln_stats_in || p.ln_stats_out) return false;
    const int ev = odt == MI_F32 ? 4 : 4;          // output rows are written four values at a time (16 | 8 bytes)
    if (p.x_rstride % 8 || p.x_bstride % 8 || p.x_goff % 8 || p.out_rstride % ev || p.out_bstride % ev) return false;
    if (((uintptr_t)p.x | (uintptr_t)p.gcp_w | (uintptr_t)p.out | (uintptr_t)p.res | (uintptr_t)p.bias) % 16) return false;
    if (p.pad < 0 || p.pad >= p.taps || p.M <= 0) return false;
    GConv16Dev d;
    d.x = p.x; d.img = p.gcp_w; d.bias = p.bias; d.out = p.out; d.res = p.res;
    d.T_in = p.T_in; d.M = p.M; d.taps = p.taps; d.pad = p.pad; d.act = p.act;
    d.x_bstride = p.x_bstride; d.x_rstride = p.x_rstride; d.x_goff = p.x_goff; d.out_bstride = p.out_bstride; d.out_rstride = p.out_rstride;
    constexpr int BM = 192;
    const int rows_a = BM + p.taps - 1;
    const size_t a_el = ((size_t)rows_a * GC16_S + 511) / 512 * 512;
    size_t lds = (a_el + (size_t)4 * GC16_WTAP) * 2;
    lds = std::max(lds, (size_t)BM * GC16_C * 4);
    if (lds > 80 * 1024) return false;
    const dim3 grid((p.M + BM - 1) / BM, p.G, p.B);
#define GC16_LAUNCH(T_, TO_)                                                                                                  \
    do {                                                                                                                      \
        auto kfn = gconv16_kernel<T_, TO_, BM>;                                                                               \
        MI_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));                 \
        prof_set_kernel("gconv16_kernel<192> (16-bit grouped conv, weights as LDS images, two taps in flight)", "", "");      \
        hipLaunchKernelGGL(kfn, grid, dim3(512), lds, s, d);                                                                  \
    } while (0)
    if (p.dtype == MI_F16) { if (odt == MI_F32) GC16_LAUNCH(f16, float); else GC16_LAUNCH(f16, f16); }
    else { if (odt == MI_F32) GC16_LAUNCH(bf16, float); else GC16_LAUNCH(bf16, bf16); }
#undef GC16_LAUNCH
    MI_HIP(hipGetLastError());
    return true;
}

}  // namespace mi
