// gemm_conv.hip — implicit-GEMM convolution / linear kernel on CDNA4 MFMA (gfx950).
//
// One kernel family serves every dense contraction on the hot path:
//   * BigVGAN Conv1d k in {3,7,11}, dilation {1,3,5}         (bigvgan.py:56-88, 136-139)
//   * BigVGAN ConvTranspose1d stride u, k = 2u  (EPI_CONVT)   (bigvgan.py:300-315, 391)
//   * DiT / Vocos / text-embed Linear layers (taps = 1)       (modules.py:329-340, 459-468)
//   * grouped position-embedding conv k31 g16 (G > 1)         (modules.py:167-190)
//   * STFT / ISTFT / mel as framed GEMMs (x_rstride = hop)     (STFT_Process.py:153-166)
//
// Activations are channels-last (B, T, C) so that the GEMM K axis (tap, ci) is contiguous in
// 16-byte vectors for the MFMA A operand and the N axis (output channel) is the lane axis of the
// 32x32 accumulator tile => 64/128-byte coalesced row stores.
//
//   M = output time steps, N = output channels, K = taps * Cin
//   A[m][kk] = x[b, m - pad + tap*dil, ci]   (zero outside [0, T_in))      kk = tap*Cin + ci
//   B[kk][n] = w[n, kk]
//
// MFMA: f32 -> v_mfma_f32_32x32x2_f32 (exact fp32), f16/bf16 -> v_mfma_f32_32x32x16_{f16,bf16};
// fp32 accumulate always.  4 waves per workgroup, each owning a (WM x WN) sub-tile of 32x32 MFMA
// tiles.  K is streamed in KC-wide chunks: global -> registers (next chunk, issued before the
// MFMAs of the current one) -> LDS -> fragments.
#include "common.h"
#include "mfma.h"

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
    const float* rope_cos; const float* rope_sin; int heads, head_dim; void* out2; void* out3;
};

template <typename T, typename TO, int BM, int BN, int WGM, int WGN, int KC>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvGemmDev p) {
    using MF = Mfma<T>;
    constexpr int KP = MF::KP;
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int KV = KC / VEC;          // 16-byte vectors per tile row
    constexpr int RPP = 256 / KV;         // tile rows filled per pass of the 256 threads
    constexpr int AP = (BM + RPP - 1) / RPP, BP = (BN + RPP - 1) / RPP;
    constexpr int LDA = KC + (sizeof(T) == 4 ? 1 : 8);   // +pad: conflict-free fragment reads
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    static_assert(WGM * WGN == 4 && WM % 32 == 0 && WN % 32 == 0, "tile");
    static_assert(KC % (2 * KP) == 0, "KC");

    __shared__ __attribute__((aligned(16))) T smem[(BM + BN) * LDA];
    T* As = smem;
    T* Bs = smem + BM * LDA;

    const int tid = threadIdx.x;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int b = blockIdx.z / p.G, g = blockIdx.z % p.G;
    const T* xb = (const T*)p.x + (long)b * p.x_bstride + (long)g * p.x_goff;
    const T* wg = (const T*)p.w + (long)g * p.N * p.K;

    const int kv = tid % KV, r0 = tid / KV;
    uint4 areg[AP], breg[BP];

    auto load_regs = [&](int kk0) {
        const int kk = kk0 + kv * VEC;
        const bool kvalid = kk < p.K;
        const int tap = kk / p.Cin;
        const int ci = kk - tap * p.Cin;
        const int rowoff = tap * p.dil - p.pad;
#pragma unroll
        for (int q = 0; q < AP; ++q) {
            const int r = r0 + q * RPP;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < BM) {
                const int row = m0 + r + rowoff;
                if (kvalid && row >= 0 && row < p.T_in)
                    v = *reinterpret_cast<const uint4*>(xb + (long)row * p.x_rstride + ci);
            }
            areg[q] = v;
        }
#pragma unroll
        for (int q = 0; q < BP; ++q) {
            const int r = r0 + q * RPP;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < BN) {
                const int n = n0 + r;
                if (kvalid && n < p.N) v = *reinterpret_cast<const uint4*>(wg + (long)n * p.K + kk);
            }
            breg[q] = v;
        }
    };
    auto store_lds = [&]() {
#pragma unroll
        for (int q = 0; q < AP; ++q) {
            const int r = r0 + q * RPP;
            if (r < BM) {
                if constexpr (sizeof(T) == 4) {
                    float* d = reinterpret_cast<float*>(As) + r * LDA + kv * VEC;
                    d[0] = __uint_as_float(areg[q].x); d[1] = __uint_as_float(areg[q].y);
                    d[2] = __uint_as_float(areg[q].z); d[3] = __uint_as_float(areg[q].w);
                } else {
                    *reinterpret_cast<uint4*>(As + r * LDA + kv * VEC) = areg[q];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < BP; ++q) {
            const int r = r0 + q * RPP;
            if (r < BN) {
                if constexpr (sizeof(T) == 4) {
                    float* d = reinterpret_cast<float*>(Bs) + r * LDA + kv * VEC;
                    d[0] = __uint_as_float(breg[q].x); d[1] = __uint_as_float(breg[q].y);
                    d[2] = __uint_as_float(breg[q].z); d[3] = __uint_as_float(breg[q].w);
                } else {
                    *reinterpret_cast<uint4*>(Bs + r * LDA + kv * VEC) = breg[q];
                }
            }
        }
    };

    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WGN, wn = wave % WGN;
    const int lr = lane & 31, lk = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_regs(0);
    store_lds();
    __syncthreads();
    for (int kk0 = 0; kk0 < p.K; kk0 += KC) {
        const bool more = kk0 + KC < p.K;
        if (more) load_regs(kk0 + KC);
#pragma unroll
        for (int ks = 0; ks < KC / (2 * KP); ++ks) {
            typename MF::Frag a[TM], bb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const typename MF::Frag*>(As + (wm * WM + i * 32 + lr) * LDA + ks * 2 * KP + lk * KP);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bb[j] = *reinterpret_cast<const typename MF::Frag*>(Bs + (wn * WN + j * 32 + lr) * LDA + ks * 2 * KP + lk * KP);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma(a[i], bb[j], acc[i][j]);
        }
        __syncthreads();
        if (more) {
            store_lds();
            __syncthreads();
        }
    }

    // ---- epilogue -------------------------------------------------------------------------
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
            TO* dst = (TO*)(which == 0 ? p.out : which == 1 ? p.out2 : p.out3) + ((long)b * p.heads + hh) * p.M * p.head_dim + dd;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    float v = acc[i][j][r] + bv;
                    const float partner = __shfl_xor(v, 1);
                    if (m < p.M) {
                        if (which < 2) {
                            const float c = p.rope_cos[(long)m * p.head_dim + dd], sn = p.rope_sin[(long)m * p.head_dim + dd];
                            v = v * c + ((dd & 1) ? partner : -partner) * sn;
                        }
                        dst[(long)m * p.head_dim] = from_f32<TO>(v);
                    }
                }
            }
        }
        return;
    }
    TO* outp = (TO*)p.out + (long)b * p.out_bstride;
    const TO* resp = p.res ? (const TO*)p.res + (long)b * p.out_bstride : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + lr;
        if (n >= p.N) continue;
        int col, ph = 0;
        float bv;
        if (p.epi == EPI_CONVT) {
            ph = n / p.Cout;
            col = n - ph * p.Cout;
            bv = p.bias ? p.bias[col] : 0.f;
        } else {
            col = g * p.N + n;
            bv = p.bias ? p.bias[col] : 0.f;
        }
        const float gv = p.gate ? p.gate[(long)b * p.gate_bstride + col] : 1.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                long row;
                if (p.epi == EPI_CONVT) {
                    row = (long)m * p.u + ph - p.padT;
                    if (m >= p.M || row < 0 || row >= p.T_out) continue;
                } else {
                    row = m;
                    if (m >= p.M) continue;
                }
                const long idx = row * p.out_rstride + col;
                float v = act_apply(acc[i][j][r] + bv, p.act) * gv;
                if (resp) v += to_f32(resp[idx]);
                v *= p.alpha;
                if (p.accumulate) v += to_f32(outp[idx]);
                outp[idx] = from_f32<TO>(v);
            }
        }
    }
}

template <typename T, typename TO>
static void dispatch_tiles(const ConvGemmDev& d, int B, hipStream_t s) {
    constexpr int KC = sizeof(T) == 4 ? 16 : 64;
    dim3 blk(256);
    if (d.N <= 32) {
        dim3 grid((d.M + 255) / 256, (d.N + 31) / 32, B * d.G);
        hipLaunchKernelGGL((conv_gemm_kernel<T, TO, 256, 32, 4, 1, KC>), grid, blk, 0, s, d);
    } else if (d.N <= 64) {
        dim3 grid((d.M + 127) / 128, (d.N + 63) / 64, B * d.G);
        hipLaunchKernelGGL((conv_gemm_kernel<T, TO, 128, 64, 2, 2, KC>), grid, blk, 0, s, d);
    } else {
        dim3 grid((d.M + 127) / 128, (d.N + 127) / 128, B * d.G);
        hipLaunchKernelGGL((conv_gemm_kernel<T, TO, 128, 128, 2, 2, KC>), grid, blk, 0, s, d);
    }
    MI_HIP(hipGetLastError());
}

void launch_conv_gemm(const ConvGemm& p, hipStream_t s) {
    const int odt = p.out_dtype < 0 ? p.dtype : p.out_dtype;
    const int vec = 16 / (int)dtype_size(p.dtype);
    MI_REQUIRE(p.Cin % vec == 0, "conv_gemm: Cin must be a multiple of the 16-byte vector");
    MI_REQUIRE(p.x_rstride % vec == 0 && p.x_bstride % vec == 0 && p.x_goff % vec == 0, "conv_gemm: x strides");
    MI_REQUIRE(((uintptr_t)p.x % 16) == 0 && ((uintptr_t)p.w % 16) == 0, "conv_gemm: 16-byte alignment");
    MI_REQUIRE(odt == p.dtype || odt == MI_F32, "conv_gemm: out dtype");
    MI_REQUIRE(p.M > 0 && p.N > 0 && p.B > 0 && p.G > 0, "conv_gemm: empty problem");
    ConvGemmDev d;
    d.x = p.x; d.w = p.w; d.bias = p.bias; d.out = p.out; d.res = p.res; d.gate = p.gate;
    d.gate_bstride = p.gate_bstride;
    d.G = p.G; d.T_in = p.T_in; d.M = p.M; d.N = p.N; d.Cin = p.Cin; d.K = p.taps * p.Cin; d.dil = p.dil; d.pad = p.pad;
    d.x_bstride = p.x_bstride; d.x_rstride = p.x_rstride; d.out_bstride = p.out_bstride; d.out_rstride = p.out_rstride;
    d.x_goff = p.x_goff;
    d.act = p.act; d.alpha = p.alpha; d.accumulate = p.accumulate; d.epi = p.epi;
    d.u = p.u; d.Cout = p.Cout; d.padT = p.padT; d.T_out = p.T_out;
    d.rope_cos = p.rope_cos; d.rope_sin = p.rope_sin; d.heads = p.heads; d.head_dim = p.head_dim;
    d.out2 = p.out2; d.out3 = p.out3;
    if (p.epi == EPI_CONVT) MI_REQUIRE(p.Cout > 0 && p.N == p.u * p.Cout, "conv_gemm: convT shape");
    if (p.epi == EPI_QKV_ROPE)
        MI_REQUIRE(p.G == 1 && p.heads > 0 && p.head_dim % 2 == 0 && p.N == 3 * p.heads * p.head_dim && p.N % 32 == 0 &&
                       p.rope_cos && p.rope_sin && p.out2 && p.out3, "conv_gemm: qkv-rope epilogue arguments");

    const double esz = (double)dtype_size(p.dtype), osz = (double)dtype_size(odt);
    // algorithmic traffic: read x once, weights once, write out once (+ residual / accumulate reads)
    const double rows_out = p.epi == EPI_CONVT ? (double)p.T_out : (double)p.M;
    const double ncols = p.epi == EPI_CONVT ? (double)p.Cout : (double)p.N * p.G;
    double bytes = (double)p.B * p.T_in * (double)p.Cin * p.G * esz + (double)p.G * p.N * d.K * esz +
                   (double)p.B * rows_out * ncols * osz * (1.0 + (p.res ? 1.0 : 0.0) + (p.accumulate ? 1.0 : 0.0));
    double flops = 2.0 * p.B * p.G * (double)p.M * p.N * d.K;
    ProfScope ps(FAM_CONV_GEMM, s, bytes, flops);

    if (p.dtype == MI_F32) {
        dispatch_tiles<float, float>(d, p.B, s);
    } else if (p.dtype == MI_F16) {
        if (odt == MI_F32) dispatch_tiles<f16, float>(d, p.B, s);
        else dispatch_tiles<f16, f16>(d, p.B, s);
    } else {
        if (odt == MI_F32) dispatch_tiles<bf16, float>(d, p.B, s);
        else dispatch_tiles<bf16, bf16>(d, p.B, s);
    }
}

}  // namespace mi
