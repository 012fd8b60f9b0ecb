// f5_kernels.hip — the HBM-bound (non-GEMM) kernels of the F5-TTS path, gfx950.
//
//   row norms        AdaLayerNorm / LayerNorm / Vocos L2 "norm"   modules.py:301-305, 321-325, 609
//                                                                 vocos/models.py:80,83, vocos/modules.py:46
//   dwconv7          depthwise Conv1d k7 (ConvNeXt blocks)        modules.py:242-244, vocos/modules.py:28
//   GRN              global response norm over the SEQUENCE axis  modules.py:217-226
//   text gather      Embedding + sinus pos-emb + filler mask      dit.py:49-73
//   STFT helpers     int16->f32 reflect pad, |.|, log-mel         Export_F5.py:122-125, STFT_Process.py:144-157
//   Vocos head       exp/clip magnitude, phase -> re/im           vocos/heads.py:55-59, STFT_Process.py:160-163
//   ISTFT OLA        overlap-add + envelope + clamp + int16       STFT_Process.py:164-166, Export_F5.py:203
//   CFG/Euler        x += (p + (p - p1)*cfg) * dt[k]              Export_F5.py:179-180
#include <type_traits>
#include "wave_reduce.h"
#include "common.h"
#include "f5_kernels.h"
#include "x3_split.h"

namespace mi {


// -----------------------------------------------------------------------------------------------
// row norm: one 64-lane wave per row, row cached in registers (D <= 64*4*MAXV), two-pass statistics
// -----------------------------------------------------------------------------------------------
template <typename TO, int MAXV>
__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ x, TO* __restrict__ y,
                                                      const float* __restrict__ a, const float* __restrict__ b,
                                                      long rows, int D, int mode, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * D;
    float4 v[MAXV], av[MAXV], bv[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = c < D ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0, 0, 0, 0);
    }
    // the affine / modulation vectors are requested with the row, not after its statistics (one L2 round trip less per wave)
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4, cc = c < D ? c : 0;
        av[i] = *reinterpret_cast<const float4*>(a + cc);
        bv[i] = *reinterpret_cast<const float4*>(b + cc);
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    float mean = 0.f, inv;
    if (mode == NORM_L2) {
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) q += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
        inv = 1.0f / sqrtf(wave_sum(q));                  // no epsilon (vocos/models.py:80)
    } else {
        mean = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < D) {
                const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
                q += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
            }
        }
        inv = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);   // biased variance
    }
    TO* yr = y + row * D;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            float o[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
            const float aa[4] = {av[i].x, av[i].y, av[i].z, av[i].w}, bb[4] = {bv[i].x, bv[i].y, bv[i].z, bv[i].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float n = (o[k] - mean) * inv;
                o[k] = mode == NORM_LN_MOD ? n * (1.f + aa[k]) + bb[k] : n * aa[k] + bb[k];
            }
            if constexpr (sizeof(TO) == 4) {
                *reinterpret_cast<float4*>(yr + c) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
                TO h[4] = {from_f32<TO>(o[0]), from_f32<TO>(o[1]), from_f32<TO>(o[2]), from_f32<TO>(o[3])};
                *reinterpret_cast<uint2*>(yr + c) = *reinterpret_cast<const uint2*>(h);
            }
        }
    }
}

// LayerNorm + AdaLN modulation with the result written as gemm_x3p.hip panel planes (three-way bf16 split of the fp32
// value, x3_split.h) instead of fp32 rows: the A operand of the QKV / FF1 GEMMs needs no separate split pass.  One wave
// per row, a lane holds 8 consecutive columns per pass (one 16-byte k-slot of each plane).
template <int MAXP, int NP>
__global__ __launch_bounds__(256) void rownorm_x3p_kernel(const float* __restrict__ x, unsigned char* __restrict__ planes,
                                                          const float* __restrict__ a, const float* __restrict__ b,
                                                          long rows, int D, float eps, int* __restrict__ satp) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * D;
    unsigned sat = 0;
    float4 v[MAXP][2], av[MAXP][2], bv[MAXP][2];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int c = (i * 64 + lane) * 8;
        const bool in = c < D;
        v[i][0] = in ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0, 0, 0, 0);
        v[i][1] = in ? *reinterpret_cast<const float4*>(xr + c + 4) : make_float4(0, 0, 0, 0);
    }
    // scale / shift requested with the row, not after its statistics (one L2 round trip less per wave)
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int c = (i * 64 + lane) * 8, cc = c < D ? c : 0;
        av[i][0] = *reinterpret_cast<const float4*>(a + cc); av[i][1] = *reinterpret_cast<const float4*>(a + cc + 4);
        bv[i][0] = *reinterpret_cast<const float4*>(b + cc); bv[i][1] = *reinterpret_cast<const float4*>(b + cc + 4);
    }
#pragma unroll
    for (int i = 0; i < MAXP; ++i)
        s += ((v[i][0].x + v[i][0].y) + (v[i][0].z + v[i][0].w)) + ((v[i][1].x + v[i][1].y) + (v[i][1].z + v[i][1].w));
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < D) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float d0 = v[i][h].x - mean, d1 = v[i][h].y - mean, d2 = v[i][h].z - mean, d3 = v[i][h].w - mean;
                q += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
            }
        }
    }
    const float inv = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);   // biased variance
    const int nch = D >> 5;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int s8 = i * 64 + lane, c = s8 * 8;
        if (c < D) {
            float o[8] = {v[i][0].x, v[i][0].y, v[i][0].z, v[i][0].w, v[i][1].x, v[i][1].y, v[i][1].z, v[i][1].w};
            const float4 a0 = av[i][0], a1 = av[i][1], b0 = bv[i][0], b1 = bv[i][1];
            const float aa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (o[k] - mean) * inv * (1.f + aa[k]) + bb[k];
            x3_u4 pl[NP];
            xnp_split8_sat<NP>(o, pl, sat);
            unsigned char* dst = planes + x3p_slot_offset(row, s8, nch, NP);
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<x3_u4*>(dst + q * X3P_PLANE) = pl[q];
        }
    }
    if constexpr (NP == 2) sat_publish(satp, sat);
}

void launch_rownorm_x3p(const float* x, void* planes, const float* a, const float* b, long rows, int D, float eps, hipStream_t s, int np, int* sat) {
    MI_REQUIRE(D % 32 == 0 && D <= 2048, "rownorm_x3p: D must be whole 32-deep chunks and <= 2048");
    MI_REQUIRE(np == 2 || np == 3, "rownorm_x3p: 2 or 3 planes");
    dim3 grid((unsigned)((rows + 3) / 4));
    ProfScope ps(FAM_NORM, s, (double)rows * D * (4.0 + 2.0 * np), 8.0 * rows * D);
    prof_set_kernel(np == 3 ? "rownorm_x3p_kernel<3 planes>" : "rownorm_x3p_kernel<2 planes>", "", "");
    const int mp = (D + 511) / 512;
#define RNP(MP, NPL) hipLaunchKernelGGL((rownorm_x3p_kernel<MP, NPL>), grid, dim3(256), 0, s, x, (unsigned char*)planes, a, b, rows, D, eps, sat)
    if (np == 3) { if (mp == 1) RNP(1, 3); else if (mp == 2) RNP(2, 3); else RNP(4, 3); }
    else { if (mp == 1) RNP(1, 2); else if (mp == 2) RNP(2, 2); else RNP(4, 2); }
#undef RNP
    MI_HIP(hipGetLastError());
}

// AdaLN fold, first block of an evaluation (the row's producer is the position convolution, whose epilogue has no fold): what
// the O / FF2 epilogues do for every later norm (gemm_epilogue.h gemm_epilogue_resid_ln) as a pass of its own — the rows
// o (1 + scale) as the QKV GEMM's A operand (panel planes, or rows of TA) plus the per-row partial (sum, M2 about the block mean) over
// 32-column blocks.  A lane holds 8 consecutive columns and a quad of lanes one block: the same partials, bit for bit, as the
// GEMM epilogues write for the same x.
template <typename TA, int MAXP, int NP>
__global__ __launch_bounds__(256) void ln_prologue_kernel(const float* __restrict__ x, void* __restrict__ aout, float* __restrict__ stats,
                                                          const float* __restrict__ scale, long rows, int D, int* __restrict__ satp) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * D;
    const int nb = D / LN_BLK;
    unsigned sat = 0;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int s8 = i * 64 + lane, c = s8 * 8;
        if (c < D) {           // D % 32 == 0: a quad is in range as a whole
            const float4 v0 = *reinterpret_cast<const float4*>(xr + c), v1 = *reinterpret_cast<const float4*>(xr + c + 4);
            const float4 a0 = *reinterpret_cast<const float4*>(scale + c), a1 = *reinterpret_cast<const float4*>(scale + c + 4);
            float o[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            float s1, s2;
            ln_block_stats(o, s1, s2);
            if ((lane & 3) == 0) *reinterpret_cast<float2*>(stats + (row * nb + (c >> 5)) * 2) = make_float2(s1, s2);
            const float g[8] = {1.f + a0.x, 1.f + a0.y, 1.f + a0.z, 1.f + a0.w, 1.f + a1.x, 1.f + a1.y, 1.f + a1.z, 1.f + a1.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] *= g[k];
            if constexpr (sizeof(TA) == 4) {
                x3_u4 pl[NP];
                xnp_split8_sat<NP>(o, pl, sat);
                unsigned char* dst = (unsigned char*)aout + x3p_slot_offset(row, s8, D >> 5, NP);
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<x3_u4*>(dst + q * X3P_PLANE) = pl[q];
            } else {
                struct alignas(16) Pk { TA v[8]; } pk;
#pragma unroll
                for (int k = 0; k < 8; ++k) pk.v[k] = from_f32<TA>(o[k]);
                *reinterpret_cast<Pk*>((TA*)aout + row * D + c) = pk;
                if constexpr (std::is_same<TA, f16>::value) sat |= f16_range_word(o);      // (see gemm_epilogue_resid_ln)
            }
        }
    }
    if constexpr ((sizeof(TA) == 4 && NP == 2) || std::is_same<TA, f16>::value) sat_publish(satp, sat);
}

void launch_ln_prologue(const float* x, void* aout, int a_dtype, int np, float* stats, const float* scale, long rows, int D, int* sat,
                        hipStream_t s) {
    MI_REQUIRE(D % 128 == 0 && D <= 2048 && (a_dtype != MI_F32 || np == 2 || np == 3), "ln_prologue: D must be a multiple of 128 and <= 2048");
    dim3 grid((unsigned)((rows + 3) / 4));
    ProfScope ps(FAM_NORM, s, (double)rows * D * (4.0 + (a_dtype == MI_F32 ? 2.0 * np : 2.0)), 4.0 * rows * D);
    prof_set_kernel("ln_prologue_kernel (AdaLN fold, first block of an evaluation)", "", "");
    const int mp = (D + 511) / 512;
#define LNP(TA, MP, NPL) hipLaunchKernelGGL((ln_prologue_kernel<TA, MP, NPL>), grid, dim3(256), 0, s, x, aout, stats, scale, rows, D, sat)
#define LNP_T(TA, NPL) do { if (mp == 1) LNP(TA, 1, NPL); else if (mp == 2) LNP(TA, 2, NPL); else LNP(TA, 4, NPL); } while (0)
    if (a_dtype == MI_F32) { if (np == 3) LNP_T(float, 3); else LNP_T(float, 2); }
    else if (a_dtype == MI_F16) LNP_T(f16, 2);
    else LNP_T(bf16, 2);
#undef LNP_T
#undef LNP
    MI_HIP(hipGetLastError());
}

// AdaLN fold, 16-bit engines: the partial (sum, M2) pairs of every row merged ONCE into (rstd, mean * rstd) — the
// order of gemm_epilogue.h ln_rows32 (low half of the blocks in index order, then the high half, then low + high), so the result
// is the one a consumer epilogue would have formed itself.  One thread per row; 4.6 MB in, 144 KB out at 8 utterances.
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ part, float2* __restrict__ fin, long rows, int nb,
                                                          float eps) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const float* rowp = part + row * (long)(nb * 2);
    const float m0 = rowp[0] * (1.0f / 32.0f);
    const float4* sp = reinterpret_cast<const float4*>(rowp);
    LnMerge h[2];
    for (int hf = 0; hf < 2; ++hf)
        for (int i = 0; i < (nb >> 2); ++i) {
            const float4 t = sp[hf * (nb >> 2) + i];
            ln_merge_add(h[hf], t.x, t.y, m0);
            ln_merge_add(h[hf], t.z, t.w, m0);
        }
    float rstd, mrstd;
    ln_merge_finish(h[0], h[1], m0, nb, eps, rstd, mrstd);
    fin[row] = make_float2(rstd, mrstd);
}
void launch_ln_finalize(const float* partials, float* fin, long rows, int D, float eps, hipStream_t s) {
    MI_REQUIRE(D % 128 == 0, "ln_finalize: D must be a multiple of 128");
    ProfScope ps(FAM_NORM, s, (double)rows * (D / LN_BLK * 8.0 + 8.0), 2.0 * rows * (D / LN_BLK));
    prof_set_kernel("ln_finalize_kernel (AdaLN fold: partial row statistics -> rstd, mean * rstd)", "", "");
    hipLaunchKernelGGL(ln_finalize_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, partials, (float2*)fin, rows, D / LN_BLK, eps);
    MI_HIP(hipGetLastError());
}

// load-time helpers of the AdaLN fold (f5.hip): G[k][j] = 1 + mod[k][col_scale + j], S[k][j] = mod[k][col_shift + j]
__global__ __launch_bounds__(256) void ln_gather_kernel(const float* __restrict__ mod, long mod_ld, long col_scale, long col_shift,
                                                        float* __restrict__ G, float* __restrict__ S, int steps, int D) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)steps * D) return;
    const long k = i / D, j = i - k * D;
    G[i] = 1.f + mod[k * mod_ld + col_scale + j];
    S[i] = mod[k * mod_ld + col_shift + j];
}
void launch_ln_gather(const float* mod, long mod_ld, long col_scale, long col_shift, float* G, float* S, int steps, int D, hipStream_t s) {
    const long n = (long)steps * D;
    hipLaunchKernelGGL(ln_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, mod, mod_ld, col_scale, col_shift, G, S, steps, D);
    MI_HIP(hipGetLastError());
}
// dst (fp32) = the values of a 16-bit tensor (what the GEMM actually multiplies by)
template <typename T>
__global__ __launch_bounds__(256) void cast_to_f32_kernel(const T* __restrict__ src, float* __restrict__ dst, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = to_f32(src[i]);
}
void launch_cast_to_f32(const void* src, int dtype, float* dst, long n, hipStream_t s) {
    if (n <= 0) return;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == MI_F32) MI_HIP(hipMemcpyAsync(dst, src, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    else if (dtype == MI_F16) hipLaunchKernelGGL(cast_to_f32_kernel<f16>, grid, dim3(256), 0, s, (const f16*)src, dst, n);
    else hipLaunchKernelGGL(cast_to_f32_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)src, dst, n);
    MI_HIP(hipGetLastError());
}
// *out_bits = max(*out_bits, bits(max |x|)) over a fp32 tensor (non-negative floats order like their bit patterns; nan / inf sort on top)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ out_bits) {
    unsigned m = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const unsigned b = __float_as_uint(x[i]) & 0x7fffffffu;
        m = b > m ? b : m;
    }
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, m);
}
void launch_absmax(const float* x, long n, unsigned* out_bits, hipStream_t s) {
    if (n <= 0) return;
    const int blocks = (int)std::min<long>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, s, x, n, out_bits);
    MI_HIP(hipGetLastError());
}

void launch_rownorm(int mode, const float* x, void* y, int out_dtype, const float* a, const float* b, long rows, int D,
                    float eps, hipStream_t s) {
    MI_REQUIRE(D % 4 == 0 && D <= 2048, "rownorm: D must be a multiple of 4 and <= 2048");
    dim3 grid((unsigned)((rows + 3) / 4));
    ProfScope ps(FAM_NORM, s, (double)rows * D * (4.0 + (double)dtype_size(out_dtype)), 8.0 * rows * D);
#define RN(TO, MV) hipLaunchKernelGGL((rownorm_kernel<TO, MV>), grid, dim3(256), 0, s, x, (TO*)y, a, b, rows, D, mode, eps)
    const int mv = D <= 256 ? 1 : D <= 512 ? 2 : D <= 1024 ? 4 : 8;
    if (out_dtype == MI_F32) { if (mv == 1) RN(float, 1); else if (mv == 2) RN(float, 2); else if (mv == 4) RN(float, 4); else RN(float, 8); }
    else if (out_dtype == MI_F16) { if (mv == 1) RN(f16, 1); else if (mv == 2) RN(f16, 2); else if (mv == 4) RN(f16, 4); else RN(f16, 8); }
    else { if (mv == 1) RN(bf16, 1); else if (mv == 2) RN(bf16, 2); else if (mv == 4) RN(bf16, 4); else RN(bf16, 8); }
#undef RN
    MI_HIP(hipGetLastError());
}

// -----------------------------------------------------------------------------------------------
// depthwise conv k7 pad 3 over time, channels-last fp32: y[b,t,c] = sum_j w[c,j] x[b,t-3+j,c] + bias[c]
// -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dwconv7_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      const float* __restrict__ w, const float* __restrict__ bias,
                                                      int T, int C) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;       // over (b, t, c/4)
    const int c4n = C / 4;
    const long total = (long)gridDim.y * T * c4n;
    (void)total;
    const int b = blockIdx.y;
    if (i >= (long)T * c4n) return;
    const int t = (int)(i / c4n), c = (int)(i - (long)t * c4n) * 4;
    const float* xb = x + (long)b * T * C;
    float acc[4] = {bias[c], bias[c + 1], bias[c + 2], bias[c + 3]};
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int tt = t - 3 + j;
        if (tt < 0 || tt >= T) continue;
        const float4 xv = *reinterpret_cast<const float4*>(xb + (long)tt * C + c);
        acc[0] = fmaf(w[(c + 0) * 7 + j], xv.x, acc[0]);
        acc[1] = fmaf(w[(c + 1) * 7 + j], xv.y, acc[1]);
        acc[2] = fmaf(w[(c + 2) * 7 + j], xv.z, acc[2]);
        acc[3] = fmaf(w[(c + 3) * 7 + j], xv.w, acc[3]);
    }
    *reinterpret_cast<float4*>(y + ((long)b * T + t) * C + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

void launch_dwconv7(const float* x, float* y, const float* w, const float* bias, int B, int T, int C, hipStream_t s) {
    MI_REQUIRE(C % 4 == 0, "dwconv7: C % 4");
    const long n = (long)T * (C / 4);
    dim3 grid((unsigned)((n + 255) / 256), B);
    ProfScope ps(FAM_OTHER, s, 8.0 * B * T * C, 14.0 * B * T * C);
    hipLaunchKernelGGL(dwconv7_kernel, grid, dim3(256), 0, s, x, y, w, bias, T, C);
    MI_HIP(hipGetLastError());
}

// -----------------------------------------------------------------------------------------------
// GRN (modules.py:217-226): Gx[c] = ||y[:, c]||_2 over the sequence; Nx = Gx / (mean_c Gx + 1e-6);
// out = gamma * (y * Nx) + beta + y.   Kernel 1: per-(b,c) sum of squares (fp32, fixed order);
// kernel 2: each workgroup re-derives mean_c Gx (C <= 4096) and applies the elementwise map.
// -----------------------------------------------------------------------------------------------
// (sixteen time slices per workgroup with four independent loads in flight per lane: the first form walked T / 4 dependent strided
// loads per lane on 32 workgroups — 83 us for a 2 x 1126 x 1024 tensor)
__global__ __launch_bounds__(1024) void grn_sumsq_kernel(const float* __restrict__ y, float* __restrict__ ss, int T, int C) {
    constexpr int P = 16;
    __shared__ float red[P][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < C) {
        const float* yb = y + (long)b * T * C + c;
        int t = part;
        for (; t + 3 * P < T; t += 4 * P) {
            const float v0 = yb[(long)t * C], v1 = yb[(long)(t + P) * C], v2 = yb[(long)(t + 2 * P) * C], v3 = yb[(long)(t + 3 * P) * C];
            a0 = fmaf(v0, v0, a0); a1 = fmaf(v1, v1, a1); a2 = fmaf(v2, v2, a2); a3 = fmaf(v3, v3, a3);
        }
        for (; t < T; t += P) { const float v = yb[(long)t * C]; a0 = fmaf(v, v, a0); }
    }
    red[part][threadIdx.x & 63] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (part == 0 && c < C) {
        float r = 0.f;
#pragma unroll
        for (int q = 0; q < P; ++q) r += red[q][threadIdx.x];          // fixed order
        ss[(long)b * C + c] = r;
    }
}

__global__ __launch_bounds__(256) void grn_apply_kernel(float* __restrict__ y, const float* __restrict__ ss,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int T, int C) {
    __shared__ float red[4];
    __shared__ float s_mean;
    const int b = blockIdx.y;
    float part = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) part += sqrtf(ss[(long)b * C + c]);
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) s_mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)C;
    __syncthreads();
    const float denom = s_mean + 1e-6f;
    const long n = (long)T * C;
    float* yb = y + (long)b * n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const float v = yb[i];
        const float nx = sqrtf(ss[(long)b * C + c]) / denom;
        yb[i] = gamma[c] * (v * nx) + beta[c] + v;
    }
}

void launch_grn(float* y, float* ss_scratch, const float* gamma, const float* beta, int B, int T, int C, hipStream_t s) {
    ProfScope ps(FAM_OTHER, s, 12.0 * B * T * C, 6.0 * B * T * C);
    hipLaunchKernelGGL(grn_sumsq_kernel, dim3((C + 63) / 64, B), dim3(1024), 0, s, y, ss_scratch, T, C);
    const long n = (long)T * C;
    const int blocks = (int)std::min<long>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(grn_apply_kernel, dim3(blocks, B), dim3(256), 0, s, y, ss_scratch, gamma, beta, T, C);
    MI_HIP(hipGetLastError());
}

// -----------------------------------------------------------------------------------------------
// text embedding gather: out[v, n, :] = filler(n) ? 0 : Emb[v == 0 ? id : 0] + pos[n]
// -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void text_gather_kernel(const int* __restrict__ ids, const float* __restrict__ emb,
                                                          const float* __restrict__ pos, float* __restrict__ out,
                                                          int N, int C) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int v = blockIdx.y, u = blockIdx.z;
    if (i >= (long)N * C) return;
    const int n = (int)(i / C), c = (int)(i - (long)n * C);
    const int id = ids[(long)u * N + n];
    float r = 0.f;
    if (id != 0) r = emb[(long)(v == 0 ? id : 0) * C + c] + pos[(long)n * C + c];
    out[((long)(2 * u + v) * N + n) * C + c] = r;
}
void launch_text_gather(const int* ids, const float* emb, const float* pos, float* out, int U, int N, int C, hipStream_t s) {
    const long n = (long)N * C;
    hipLaunchKernelGGL(text_gather_kernel, dim3((unsigned)((n + 255) / 256), 2, U), dim3(256), 0, s, ids, emb, pos, out, N, C);
    MI_HIP(hipGetLastError());
}

// ids_out[u, n] = n < T ? ids_in[u, n] + 1 : 0 (text_ids + 1, zero = filler: Export_F5.py:136); ids outside the embedding
// table raise *err (checked by the host at its next synchronisation point) and are clamped to the filler
__global__ __launch_bounds__(256) void text_ids_kernel(const int32_t* __restrict__ in, int* __restrict__ out, int T, int N,
                                                       int vocab, int* __restrict__ err) {
    const int n = blockIdx.x * 256 + threadIdx.x, u = blockIdx.y;
    if (n >= N) return;
    int id = 0;
    if (n < T) {
        const long v = (long)in[(long)u * T + n] + 1;
        if (v < 0 || v > vocab) atomicOr(err, 1); else id = (int)v;
    }
    out[(long)u * N + n] = id;
}
void launch_text_ids(const int32_t* in, int* out, int U, int T, int N, int vocab, int* err, hipStream_t s) {
    hipLaunchKernelGGL(text_ids_kernel, dim3((unsigned)((N + 255) / 256), U), dim3(256), 0, s, in, out, T, N, vocab, err);
    MI_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void mask_rows_kernel(const int* __restrict__ ids, float* __restrict__ x, int N, int C) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * C) return;
    const int n = (int)(i / C);
    if (ids[(long)(blockIdx.y >> 1) * N + n] == 0) x[(long)blockIdx.y * N * C + i] = 0.f;       // slab 2u + v uses utterance u's ids
}
void launch_mask_rows(const int* ids, float* x, int V, int N, int C, hipStream_t s) {
    const long n = (long)N * C;
    hipLaunchKernelGGL(mask_rows_kernel, dim3((unsigned)((n + 255) / 256), V), dim3(256), 0, s, ids, x, N, C);
    MI_HIP(hipGetLastError());
}

// -----------------------------------------------------------------------------------------------
// generic strided copy/cast: dst[r*ldd + c] = (TO) src[r*lds + c]   (fp32 source)
// -----------------------------------------------------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void copy2d_kernel(const float* __restrict__ src, long lds_, TO* __restrict__ dst,
                                                     long ldd, long rows, int cols) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols) return;
    const long r = i / cols;
    const int c = (int)(i - r * cols);
    dst[r * ldd + c] = from_f32<TO>(src[r * lds_ + c]);
}
void launch_copy2d(const float* src, long lds_, void* dst, long ldd, long rows, int cols, int out_dtype, hipStream_t s) {
    const long n = rows * cols;
    if (n == 0) return;
    dim3 grid((unsigned)((n + 255) / 256));
    if (out_dtype == MI_F32) hipLaunchKernelGGL(copy2d_kernel<float>, grid, dim3(256), 0, s, src, lds_, (float*)dst, ldd, rows, cols);
    else if (out_dtype == MI_F16) hipLaunchKernelGGL(copy2d_kernel<f16>, grid, dim3(256), 0, s, src, lds_, (f16*)dst, ldd, rows, cols);
    else hipLaunchKernelGGL(copy2d_kernel<bf16>, grid, dim3(256), 0, s, src, lds_, (bf16*)dst, ldd, rows, cols);
    MI_HIP(hipGetLastError());
}

// -----------------------------------------------------------------------------------------------
// STFT helpers
// -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pad_reflect_kernel(const int16_t* __restrict__ a, float* __restrict__ out, long L, int half) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= L + 2 * half) return;
    a += (long)blockIdx.y * L; out += (long)blockIdx.y * (L + 2 * half);
    long j = i - half;
    if (j < 0) j = -j;
    if (j >= L) j = 2 * (L - 1) - j;
    out[i] = (float)a[j] * (1.0f / 32768.0f);
}
void launch_pad_reflect(const int16_t* a, float* out, int U, long L, int half, hipStream_t s) {
    hipLaunchKernelGGL(pad_reflect_kernel, dim3((unsigned)((L + 2 * half + 255) / 256), U), dim3(256), 0, s, a, out, L, half);
    MI_HIP(hipGetLastError());
}

// spec [F][2*nb] (re | im) -> mag [F][ldm] (zero padded columns)
__global__ __launch_bounds__(256) void spec_mag_kernel(const float* __restrict__ spec, float* __restrict__ mag, int F, int nb, int ldm, float eps) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)F * ldm) return;
    const int f = (int)(i / ldm), k = (int)(i - (long)f * ldm);
    float v = 0.f;
    if (k < nb) { const float re = spec[(long)f * 2 * nb + k], im = spec[(long)f * 2 * nb + nb + k]; v = sqrtf(re * re + im * im + eps); }
    mag[i] = v;
}
void launch_spec_mag(const float* spec, float* mag, int F, int nb, int ldm, float eps, hipStream_t s) {
    const long n = (long)F * ldm;
    hipLaunchKernelGGL(spec_mag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, spec, mag, F, nb, ldm, eps);
    MI_HIP(hipGetLastError());
}

// cat_mel_text[n, 0:M] = n < R ? log(max(melraw[n], 1e-5)) : 0 ;  cat_mel_text_drop[n, 0:M] = 0
__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ melraw, float* __restrict__ cmt,
                                                     float* __restrict__ cmtd, int N, int R, int M, int ld) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * M) return;
    const int n = (int)(i / M), m = (int)(i - (long)n * M);
    const long u = blockIdx.y;
    melraw += u * R * M; cmt += u * N * ld; cmtd += u * N * ld;
    cmt[(long)n * ld + m] = n < R ? logf(fmaxf(melraw[(long)n * M + m], 1e-5f)) : 0.f;
    cmtd[(long)n * ld + m] = 0.f;
}
void launch_logmel(const float* melraw, float* cmt, float* cmtd, int U, int N, int R, int M, int ld, hipStream_t s) {
    const long n = (long)N * M;
    hipLaunchKernelGGL(logmel_kernel, dim3((unsigned)((n + 255) / 256), U), dim3(256), 0, s, melraw, cmt, cmtd, N, R, M, ld);
    MI_HIP(hipGetLastError());
}

// -----------------------------------------------------------------------------------------------
// Vocos head: s [rows][2*nb] -> c [rows][ldc] = [min(exp(s_mag),100)*cos(ph) | ...*sin(ph) | 0 pad]
// -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vocos_head_kernel(const float* __restrict__ sp, float* __restrict__ c, long rows, int nb, int ldc) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * ldc) return;
    const long r = i / ldc;
    const int k = (int)(i - r * ldc);
    float v = 0.f;
    if (k < 2 * nb) {
        const int kk = k < nb ? k : k - nb;
        const float mag = fminf(expf(sp[r * 2 * nb + kk]), 100.0f);
        const float ph = sp[r * 2 * nb + nb + kk];
        v = k < nb ? mag * cosf(ph) : mag * sinf(ph);
    }
    c[i] = v;
}
void launch_vocos_head(const float* sp, float* c, long rows, int nb, int ldc, hipStream_t s) {
    const long n = rows * ldc;
    hipLaunchKernelGGL(vocos_head_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sp, c, rows, nb, ldc);
    MI_HIP(hipGetLastError());
}

// frames [U][F][n_fft] -> out[u][i], i in [0,(F-1)*hop): sum of the <= n_fft/hop overlapping frames at
// t = i + n_fft/2, times window_sum_inv[t]; then clamp +-1, *32767, truncate (Export_F5.py:203).
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ fr, const float* __restrict__ wsi,
                                                        int F, int nfft, int hop, float* __restrict__ out_f,
                                                        int16_t* __restrict__ out_i) {
    const long len = (long)(F - 1) * hop;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int u = blockIdx.y;
    if (i >= len) return;
    const long t = i + nfft / 2;
    long f1 = t / hop;
    if (f1 > F - 1) f1 = F - 1;
    long f0 = (t - nfft + hop) / hop;            // ceil((t - nfft + 1) / hop)
    if (t - nfft + 1 <= 0) f0 = 0;
    const float* fb = fr + (long)u * F * nfft;
    float acc = 0.f;
    for (long f = f0; f <= f1; ++f) acc += fb[f * nfft + (t - f * hop)];     // ascending frame order == conv_transpose
    float v = acc * wsi[t];
    if (out_f) out_f[(long)u * len + i] = v;
    if (out_i) {
        v = fminf(fmaxf(v, -1.0f), 1.0f) * 32767.0f;
        out_i[(long)u * len + i] = (int16_t)v;
    }
}
void launch_istft_ola(const float* frames, const float* wsi, int U, int F, int nfft, int hop, float* out_f,
                      int16_t* out_i, hipStream_t s) {
    const long len = (long)(F - 1) * hop;
    if (len <= 0) return;
    hipLaunchKernelGGL(istft_ola_kernel, dim3((unsigned)((len + 255) / 256), U), dim3(256), 0, s, frames, wsi, F, nfft, hop, out_f, out_i);
    MI_HIP(hipGetLastError());
}

// -----------------------------------------------------------------------------------------------
// sampler glue
// -----------------------------------------------------------------------------------------------
// cat[(2u+br), n, 0:M] = noise[u, n, :]   (both CFG branches see the same x)
template <typename T>
__global__ __launch_bounds__(256) void cat_noise_kernel(const float* __restrict__ noise, T* __restrict__ cat, long UN, int M, int ldc, int N) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= UN * M) return;
    const long un = i / M;
    const int m = (int)(i - un * M);
    const long u = un / N, n = un - u * N;
    const T v = from_f32<T>(noise[i]);
    cat[((2 * u) * N + n) * ldc + m] = v;
    cat[((2 * u + 1) * N + n) * ldc + m] = v;
}
void launch_cat_noise(const float* noise, void* cat, int U, int N, int M, int ldc, int dtype, hipStream_t s) {
    const long n = (long)U * N * M;
    dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == MI_F32) hipLaunchKernelGGL(cat_noise_kernel<float>, grid, dim3(256), 0, s, noise, (float*)cat, (long)U * N, M, ldc, N);
    else if (dtype == MI_F16) hipLaunchKernelGGL(cat_noise_kernel<f16>, grid, dim3(256), 0, s, noise, (f16*)cat, (long)U * N, M, ldc, N);
    else hipLaunchKernelGGL(cat_noise_kernel<bf16>, grid, dim3(256), 0, s, noise, (bf16*)cat, (long)U * N, M, ldc, N);
    MI_HIP(hipGetLastError());
}

// noise[u,n,m] += (p_c + (p_c - p_u) * cfg) * dt ; pred layout [(2u+br)][N][parts][M]: the K slices of proj_out side by side (F5::proj_parts),
// summed here in slice order
__global__ __launch_bounds__(256) void cfg_update_kernel(float* __restrict__ noise, const float* __restrict__ pred,
                                                         long NM, int M, int parts, long total, float cfg, const float* __restrict__ dt, int k) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long u = i / NM, r = i - u * NM;
    const long n = r / M; const int m = (int)(r - n * M);
    const float* c = pred + ((2 * u) * NM + n * M) * parts + m;
    const float* un = c + NM * parts;
    float pc = c[0], pu = un[0];
    for (int q = 1; q < parts; ++q) { pc += c[(long)q * M]; pu += un[(long)q * M]; }
    noise[i] += (pc + (pc - pu) * cfg) * dt[k];
}
void launch_cfg_update(float* noise, const float* pred, int U, int N, int M, float cfg, const float* dt, int k, hipStream_t s, int parts) {
    const long total = (long)U * N * M;
    hipLaunchKernelGGL(cfg_update_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, noise, pred, (long)N * M, M, parts, total, cfg, dt, k);
    MI_HIP(hipGetLastError());
}

// out[row][m] = sum over the slices of in[row][q][m], in slice order (the export of F5::pred_rows)
__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ in, float* __restrict__ out, long total, int M, int parts) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long row = i / M; const int m = (int)(i - row * M);
    const float* p = in + row * M * parts + m;
    float v = p[0];
    for (int q = 1; q < parts; ++q) v += p[(long)q * M];
    out[i] = v;
}
void launch_sum_parts(const float* in, float* out, long rows, int M, int parts, hipStream_t s) {
    const long total = rows * M;
    hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, total, M, parts);
    MI_HIP(hipGetLastError());
}

}  // namespace mi
