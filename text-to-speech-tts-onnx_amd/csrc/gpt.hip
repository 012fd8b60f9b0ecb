// gpt.hip — IndexTTS acoustic GPT-2 on gfx950: the autoregressive mel-code decoder behind graphs B, C and E.
//
//   text_embed   = IndexTTS_B.forward   IndexTTS/Export_IndexTTS.py:210-214
//   mel_embed    = IndexTTS_C.forward   :222-225
//   forward_rows = IndexTTS_E.forward   :270-289  (LayerNorm -> per-head q/k/v -> cache append -> softmax(qk + mask) v ->
//                                                   c_proj -> MLP(gelu_new) ; ln_f(last row) ; lm_head * penalty ; argmax)
//   decode_steps = the while-loop of IndexTTS/Inference_IndexTTS_ONNX.py:752-783, one hipGraph replay per token
//
// What is different from the reference graph, and why: the reference passes the whole KV cache in and out of every
// ORT call (2 * layers tensors that grow by one column per token, re-concatenated each step, :277-278).  Here the cache
// is a fixed [layer][head][max_seq][64] allocation in HBM that a step appends one row to; the history length, the last
// token, the penalty vector and the per-token outputs are device state, so a decode step has no host-dependent
// argument: it is captured once and replayed.  A decode step is a chain of weight-streaming GEMVs (HBM-bound:
// every weight byte is read once per token), so the GEMV kernel is a plain coalesced 16-byte-per-lane dot product,
// not an MFMA tile; the prompt pass (ids_len rows at once) goes through the MFMA implicit-GEMM kernel instead.
#include <atomic>
#include "gpt.h"
#include "mfma.h"
#include "wave_reduce.h"
#include <cstdlib>

namespace mi {

static std::atomic<long> g_gpt_mfma_min = 9;
bool gpt_set_option(const char* key, long v) {
    if (std::string(key) == "gpt_mfma_min") { g_gpt_mfma_min = v; return true; }
    return false;
}

GptCfg parse_gpt_cfg(const int32_t* ci, int ni) {
    MI_REQUIRE(ci && (ni == 9 || ni == 10), "gpt cfg: expected 9 or 10 ints");
    GptCfg c;
    int i = 0;
    c.hidden = ci[i++]; c.layers = ci[i++]; c.heads = ci[i++]; c.inner = ci[i++]; c.mel_codes = ci[i++];
    c.text_tokens = ci[i++]; c.max_mel_pos = ci[i++]; c.max_text_pos = ci[i++]; c.max_seq = ci[i++];
    MI_REQUIRE(c.heads > 0 && c.hidden == c.heads * 64, "gpt: the attention kernel is built for head_dim 64");
    MI_REQUIRE(c.hidden % 8 == 0 && c.hidden <= 2048 && c.inner % 8 == 0 && c.inner > 0, "gpt cfg: widths");
    MI_REQUIRE(c.layers > 0 && c.mel_codes > 1 && c.text_tokens > 1 && c.max_mel_pos > 1 && c.max_text_pos > 2, "gpt cfg: sizes");
    MI_REQUIRE(c.max_seq >= 8 && c.max_seq <= 8192, "gpt cfg: max_seq must be in [8, 8192]");
    c.max_batch = ni == 10 ? ci[9] : 1;
    MI_REQUIRE(c.max_batch >= 1 && c.max_batch <= 16, "gpt cfg: max_batch must be in [1, 16]");
    return c;
}

int64_t gpt_param_count(const GptCfg& c) {
    const int64_t h = c.hidden, n = c.inner;
    int64_t t = (int64_t)(c.text_tokens + c.max_text_pos + c.mel_codes + c.max_mel_pos) * h;
    t += (int64_t)c.layers * (2 * h + 3 * h * h + 3 * h + h * h + h + 2 * h + h * n + n + n * h + h);
    t += 4 * h + (int64_t)c.mel_codes * h + c.mel_codes;
    return t;
}

// ---------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------
template <typename T> struct Pack16 {
    static constexpr int N = 16 / sizeof(T);
    T v[N];
};
template <typename T> __device__ inline Pack16<T> ld16(const T* p) {
    Pack16<T> r;
    *reinterpret_cast<uint4*>(&r) = *reinterpret_cast<const uint4*>(p);
    return r;
}
// acc += dot(a[0..V), b[0..V)) with fp32 accumulation; 16-bit types use the packed dot instructions (v_dot2_f32_f16 /
// v_dot2_f32_bf16: two multiply-adds per lane per issue, no conversions)
__device__ inline float dot_pack(const Pack16<float>& a, const Pack16<float>& b, float acc) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = fmaf(a.v[e], b.v[e], acc);
    return acc;
}
__device__ inline float dot_pack(const Pack16<f16>& a, const Pack16<f16>& b, float acc) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2* pa = reinterpret_cast<const h2*>(&a);
    const h2* pb = reinterpret_cast<const h2*>(&b);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_fdot2(pa[e], pb[e], acc, false);
    return acc;
}
__device__ inline float dot_pack(const Pack16<bf16>& a, const Pack16<bf16>& b, float acc) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const b2* pa = reinterpret_cast<const b2*>(&a);
    const b2* pb = reinterpret_cast<const b2*>(&b);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_fdot2_f32_bf16(pa[e], pb[e], acc, false);
    return acc;
}

__device__ inline float gelu_new(float x) {
    return 0.5f * x * (1.f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}

// ---------------------------------------------------------------------------------------------------------------
// Decode-step linear layer of ONE sentence: out[n] = act(dot(w[n, :], x') + bias[n]) (+ res[n]), one wave per R
// output rows, lane l holds k = l*V + it*64*V .. +V of the row (V = elements per 16 bytes), fp32 FMAs, DPP reduction.
// A decode step is a chain of ~120 of these launches, each a dependent load -> reduce -> store chain of a few us, so
// the kernel is built around its latency, not its bandwidth (profiles/r3/gpt_decode_*_kernel_stats.csv: 7.3 -> 4.7 us
// at 9.8 MB of weights; a trivial one-block kernel costs 4.4 us in the same graph):
//   * every global read of the block is in flight before the first reduction, in the order x-side operands ->
//     weights (loads return in order: the statistics must not queue behind the weight rows), all branch-free —
//     a load under an exec-masked branch gets its own s_waitcnt vmcnt(0) from the compiler and the R x KI weight loads
//     serialise (measured: 9.1 us instead of 4.7); the weight loads are non-temporal (every byte is used once);
//   * the epilogue's bias / residual / cache position are fetched up front as well;
//   * the x row is prepared ONCE per block and shared through LDS: LN form (x = fp32 residual row, K <= 2048):
//     every wave derives the row statistics itself from registers (no block-wide reduction), then each wave turns one
//     (iteration, 4-element group) slice into LayerNorm(x) for all; plain form: the 16-bit activation row;
//   * `pre_w`: the row first goes through another LayerNorm with rownorm_kernel's lane map and arithmetic
//     (f5_kernels.hip) — ln_f in front of the lm_head's final_norm, Export_IndexTTS.py:286-288 — and block 0 stores
//     that intermediate row to `pre_out` (graph E's last_hidden_state output);
//   * QKV: rows [hidden, 3*hidden) go straight into the K / V cache row st[GS_HIST] of this layer.
#ifndef GPT_NT
#define GPT_NT 1
#endif
#if GPT_NT
#define GPT_WLOAD ldnt16
#else
#define GPT_WLOAD ld16
#endif
typedef unsigned int gd_u4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ inline Pack16<T> ldnt16(const T* p) {
    Pack16<T> r;
    *reinterpret_cast<gd_u4*>(&r) = __builtin_nontemporal_load(reinterpret_cast<const gd_u4*>(p));
    return r;
}

template <typename T, int R, int KI, int WAVES, bool LN, bool QKV>
__global__ __launch_bounds__(WAVES * 64) void gemv_d_kernel(const T* __restrict__ w, const void* __restrict__ xin,
                                                            const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                            const float* __restrict__ pre_w, const float* __restrict__ pre_b,
                                                            float* __restrict__ pre_out,
                                                            const float* __restrict__ bias, const float* res, void* out,
                                                            int out_f32, int act, int N, int K, T* __restrict__ kc,
                                                            T* __restrict__ vc, const int* __restrict__ st, int max_seq) {
    constexpr int V = Pack16<T>::N;
    constexpr int KP = KI * 64 * V;                       // padded K
    __shared__ __attribute__((aligned(16))) float xs[LN ? KP : 4];
    __shared__ __attribute__((aligned(16))) T xt[LN ? 8 : ((KI + WAVES - 1) / WAVES) * WAVES * 64 * V];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n0 = (blockIdx.x * WAVES + wave) * R;
    // epilogue operands of the row this lane will store (lane r < R), fetched up front: after the reduction they would
    // be one more dependent L2 round trip each
    const int nrow = min(n0 + (lane < R ? lane : 0), N - 1);
    const float bias_pre = bias ? bias[nrow] : 0.f;
    const float res_pre = res ? res[nrow] : 0.f;
    const int pos_pre = QKV ? st[GS_HIST] : 0;
    Pack16<T> p[KI][R];
    auto load_weights = [&]() {
#pragma unroll
    for (int it = 0; it < KI; ++it) {
        const int k = lane * V + it * 64 * V;
#pragma unroll
        for (int r = 0; r < R; ++r)         // branch-free: padded lanes re-read k = 0 of the row and meet x = 0 in LDS
            p[it][r] = GPT_WLOAD(w + (size_t)min(n0 + r, N - 1) * K + (k < K ? k : 0));
    }
    };
    if constexpr (LN) {
        const float* xf = (const float*)xin;
        if (pre_w) {      // rownorm_kernel<float, 8>, NORM_LN_AFFINE, one wave per row: here wave 0
            if (wave == 0) {
                constexpr int MAXV = 8;
                float4 v[MAXV];
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < MAXV; ++i) {
                    const int c = (i * 64 + lane) * 4;
                    v[i] = c < K ? *reinterpret_cast<const float4*>(xf + c) : make_float4(0, 0, 0, 0);
                    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
                }
                const float mean = wave_sum(s) / (float)K;
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < MAXV; ++i) {
                    const int c = (i * 64 + lane) * 4;
                    if (c < K) {
                        const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
                        q += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
                    }
                }
                const float inv = 1.0f / sqrtf(wave_sum(q) / (float)K + 1e-5f);
#pragma unroll
                for (int i = 0; i < MAXV; ++i) {
                    const int c = (i * 64 + lane) * 4;
                    if (c < K) {
                        const float4 av = *reinterpret_cast<const float4*>(pre_w + c);
                        const float4 bv = *reinterpret_cast<const float4*>(pre_b + c);
                        float o[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
                        const float aa[4] = {av.x, av.y, av.z, av.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) { const float n = (o[k] - mean) * inv; o[k] = n * aa[k] + bb[k]; }
                        const float4 ov = make_float4(o[0], o[1], o[2], o[3]);
                        *reinterpret_cast<float4*>(&xs[c]) = ov;
                        if (blockIdx.x == 0 && pre_out) *reinterpret_cast<float4*>(pre_out + c) = ov;
                    }
                }
            }
            __syncthreads();
        }
        float xe[KI][V];
#pragma unroll
        for (int it = 0; it < KI; ++it) {
            const int k = lane * V + it * 64 * V;
            const bool ok = k < K;
            const int kk = ok ? k : 0;
#pragma unroll
            for (int e = 0; e < V; e += 4) {
                float4 v;
                if (pre_w) v = *reinterpret_cast<const float4*>(&xs[kk + e]);      // uniform branch
                else v = *reinterpret_cast<const float4*>(xf + kk + e);
                xe[it][e] = ok ? v.x : 0.f; xe[it][e + 1] = ok ? v.y : 0.f; xe[it][e + 2] = ok ? v.z : 0.f; xe[it][e + 3] = ok ? v.w : 0.f;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // gamma / beta of the (iteration, 4-element group) items this wave turns into LayerNorm(x); then the weights:
        // everything the block reads is in flight before the first reduction, in this order (loads return in order: the
        // statistics must not wait behind the weight rows)
        constexpr int QV = V / 4, NITEM = KI * QV, MYI = (NITEM + WAVES - 1) / WAVES;
        float4 gq[MYI], bq[MYI];
#pragma unroll
        for (int m = 0; m < MYI; ++m) {
            const int item = wave + m * WAVES;
            const int k = lane * V + (item / QV) * 64 * V + (item % QV) * 4;
            const int kk = item < NITEM && k < K ? k : 0;
            gq[m] = *reinterpret_cast<const float4*>(ln_w + kk);
            bq[m] = *reinterpret_cast<const float4*>(ln_b + kk);
        }
        __builtin_amdgcn_sched_barrier(0);
        load_weights();
        __builtin_amdgcn_sched_barrier(0);
        float sum = 0.f;
#pragma unroll
        for (int it = 0; it < KI; ++it)
#pragma unroll
            for (int e = 0; e < V; ++e) sum += xe[it][e];
        const float mean = wave_sum(sum) / (float)K;
        float sq = 0.f;
#pragma unroll
        for (int it = 0; it < KI; ++it) {
            const bool ok = lane * V + it * 64 * V < K;
#pragma unroll
            for (int e = 0; e < V; ++e) { const float d = ok ? xe[it][e] - mean : 0.f; sq = fmaf(d, d, sq); }
        }
        const float rstd = rsqrtf(wave_sum(sq) / (float)K + 1e-5f);
        if (pre_w) __syncthreads();                       // every wave holds its share of the ln_f row: xs is reused
        // LayerNorm(x) -> LDS, one (iteration, 4-element group) item per wave and round
#pragma unroll
        for (int item = 0; item < NITEM; ++item) {
            if (item % WAVES == wave) {
                const int it = item / QV, q = item % QV, m = item / WAVES;
                const int k = lane * V + it * 64 * V + q * 4;
                float4 o = make_float4(0, 0, 0, 0);
                if (k < K) {
                    const float4 g = gq[m], b = bq[m];
                    o.x = (xe[it][q * 4 + 0] - mean) * rstd * g.x + b.x;
                    o.y = (xe[it][q * 4 + 1] - mean) * rstd * g.y + b.y;
                    o.z = (xe[it][q * 4 + 2] - mean) * rstd * g.z + b.z;
                    o.w = (xe[it][q * 4 + 3] - mean) * rstd * g.w + b.w;
                }
                *reinterpret_cast<float4*>(&xs[k]) = o;
            }
        }
        __syncthreads();
    } else {
        const T* xr = (const T*)xin;
        constexpr int XI = (KI + WAVES - 1) / WAVES;       // the x row first (L2), then the weights (HBM)
        gd_u4 xv[XI];                                      // bounds-checked buffer loads: 0 beyond K, no branch
        const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)xr, 0, K * (int)sizeof(T), 0x00020000);
#pragma unroll
        for (int i = 0; i < XI; ++i)
            xv[i] = __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)((threadIdx.x + i * WAVES * 64) * 16), 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_weights();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < XI; ++i) *reinterpret_cast<gd_u4*>(&xt[(threadIdx.x + i * WAVES * 64) * V]) = xv[i];
        __syncthreads();
    }
    float acc[R];                                         // rows beyond N: computed on clamped rows, never stored
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
#pragma unroll
    for (int it = 0; it < KI; ++it) {
        const int k = lane * V + it * 64 * V;
        {
            float xn[V];
            if constexpr (LN) {
#pragma unroll
                for (int e = 0; e < V; e += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(&xs[k + e]);
                    xn[e] = v.x; xn[e + 1] = v.y; xn[e + 2] = v.z; xn[e + 3] = v.w;
                }
            } else {
                const Pack16<T> xv = ld16(&xt[k]);
#pragma unroll
                for (int e = 0; e < V; ++e) xn[e] = (float)xv.v[e];
            }
#pragma unroll
            for (int e = 0; e < V; ++e)
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = fmaf((float)p[it][r].v[e], xn[e], acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
    if (lane < R && n0 + lane < N) {
        const int n = n0 + lane;
        float v = acc[0];
#pragma unroll
        for (int r = 1; r < R; ++r) v = lane == r ? acc[r] : v;
        v += bias_pre;
        if (act == ACT_GELU_TANH) v = gelu_new(v);
        v += res_pre;
        if (QKV) {
            const int hidden = N / 3;
            if (n >= hidden) {
                const int c = n - hidden, which = c / hidden, cc = c % hidden;
                if (pos_pre < max_seq) (which ? vc : kc)[((size_t)(cc >> 6) * max_seq + pos_pre) * 64 + (cc & 63)] = (T)v;
                return;
            }
        }
        if (out_f32) ((float*)out)[n] = v; else ((T*)out)[n] = (T)v;
    }
}

// NV per-lane partial sums -> full sums: after the call, lane l holds the total of value index
//   idx = sum_s ((l >> (5 - s)) & 1) << (log2(NV) - 1 - s)   (and every lane sharing those bits holds the same total).
// log2(NV) halving exchange steps (NV - 1 shuffles) + the remaining butterfly, instead of 6 * NV shuffles.
template <int NV> __device__ inline float reduce_multi(float (&v)[NV], int lane) {
    int off = 32;
#pragma unroll
    for (int n = NV; n > 1; n >>= 1) {
        const bool hi = (lane & off) != 0;
#pragma unroll
        for (int i = 0; i < n / 2; ++i) {
            const float send = hi ? v[i] : v[i + n / 2];
            const float keep = hi ? v[i + n / 2] : v[i];
            v[i] = keep + __shfl_xor(send, off, 64);
        }
        off >>= 1;
    }
    for (; off > 0; off >>= 1) v[0] += __shfl_xor(v[0], off, 64);
    return v[0];
}

// batched decode step: out[b][n] = act(dot(w[n, :], x[b, :]) + bias[n]) (+ res[b][n]) for BB slots at once — every
// weight row is streamed once for all sentences.  QKV: k / v rows of slot b go to that slot's cache row st[b].hist.
// 8 waves per block share the x rows through LDS in 1024-element K chunks (read straight from L2 by every wave they
// cost BB x the weight traffic); a chunk's weight loads are issued before the barrier that publishes the x chunk.
constexpr int GB_KC = 1024;
template <typename T, int R, int BB, bool QKV>
__global__ __launch_bounds__(512) void gemv_b_kernel(const T* __restrict__ w, const T* __restrict__ x,
                                                     const float* __restrict__ bias, const float* res, void* out,
                                                     int out_f32, int act, int N, int K, int nb, T* __restrict__ kc,
                                                     T* __restrict__ vc, const int* __restrict__ st, int max_seq,
                                                     size_t slot_stride) {
    constexpr int V = Pack16<T>::N;
    constexpr int ITS = GB_KC / (64 * V);          // k iterations of a wave inside one chunk
    __shared__ __attribute__((aligned(16))) T xs[BB * GB_KC];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n0 = (blockIdx.x * 8 + wave) * R;
    const T* wr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wr[r] = w + (size_t)min(n0 + r, N - 1) * K;
    float acc[R][BB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int b = 0; b < BB; ++b) acc[r][b] = 0.f;
    for (int k0 = 0; k0 < K; k0 += GB_KC) {
        const int kn = min(GB_KC, K - k0);
        Pack16<T> p[ITS][R];
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            const int kl = lane * V + it * 64 * V;
#pragma unroll
            for (int r = 0; r < R; ++r) p[it][r] = ld16(wr[r] + k0 + (kl < kn ? kl : 0));
        }
        __syncthreads();                            // the previous chunk has been consumed
        for (int i = threadIdx.x * V; i < BB * kn; i += 512 * V) {
            const int b = i / kn, kk = i % kn;      // kn is a multiple of V
            *reinterpret_cast<uint4*>(&xs[b * GB_KC + kk]) = *reinterpret_cast<const uint4*>(x + (size_t)b * K + k0 + kk);
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            const int kl = lane * V + it * 64 * V;
            if (kl < kn) {
#pragma unroll
                for (int b = 0; b < BB; ++b) {
                    const Pack16<T> xv = ld16(&xs[b * GB_KC + kl]);
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r][b] = dot_pack(p[it][r], xv, acc[r][b]);
                }
            }
        }
    }
    if (n0 >= N) return;
    // lane -> slot index it ends up holding (see reduce_multi)
    constexpr int LOG = BB == 16 ? 4 : BB == 8 ? 3 : BB == 4 ? 2 : BB == 2 ? 1 : 0;
    int b = 0;
#pragma unroll
    for (int s = 0; s < LOG; ++s) b |= ((lane >> (5 - s)) & 1) << (LOG - 1 - s);
    const bool writer = (lane & ((64 >> LOG) - 1)) == 0 && b < nb;
    float tot[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float tmp[BB];
#pragma unroll
        for (int q = 0; q < BB; ++q) tmp[q] = acc[r][q];
        tot[r] = reduce_multi<BB>(tmp, lane);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int n = n0 + r;
        if (writer && n < N) {
            float v = tot[r] + (bias ? bias[n] : 0.f);
            if (act == ACT_GELU_TANH) v = gelu_new(v);
            if (res) v += res[(size_t)b * N + n];
            bool to_cache = false;
            if (QKV) {
                const int hidden = N / 3;
                if (n >= hidden) {
                    const int c = n - hidden, which = c / hidden, cc = c % hidden;
                    const int pos = st[b * GS_WORDS + GS_HIST];
                    if (pos < max_seq)
                        (which ? vc : kc)[(size_t)b * slot_stride + ((size_t)(cc >> 6) * max_seq + pos) * 64 + (cc & 63)] = (T)v;
                    to_cache = true;
                }
            }
            if (!to_cache) {
                if (out_f32) ((float*)out)[(size_t)b * N + n] = v; else ((T*)out)[(size_t)b * N + n] = (T)v;
            }
        }
    }
}

// Batched decode step on the matrix cores (16-bit engines, 9..16 sentences; option "gpt_mfma_min" moves the threshold): out[b][n] = sum_k w[n][k] x[b][k] as
// v_mfma_f32_16x16x32 with the WEIGHT rows as the A operand (16 rows per wave) and the sentences as the 16 B columns,
// so one pass over a weight row serves every sentence at full rate (the v_dot2 GEMV above costs ~1 us per sentence per
// launch).  A wave owns 16 weight rows x K/KS of the K axis; within each 64-wide K block lane group g = lane>>4 takes
// k = 16g .. 16g+15 for BOTH operands (the contraction is order-free), so a lane's two 16-byte loads are adjacent and
// the four groups cover one full 128-byte line of the row.  The KS partial tiles meet in LDS.  x (<= 16 x K) is read
// from L2 by every wave: as many bytes as the weights, but they never leave the chip.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <typename T> struct Mfma16;
template <> struct Mfma16<f16> {
    using Frag = f16x8;
    static __device__ __forceinline__ f32x4_t mma(Frag a, Frag b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma16<bf16> {
    using Frag = bf16x8;
    static __device__ __forceinline__ f32x4_t mma(Frag a, Frag b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

template <typename T, bool QKV, int UNR>
__global__ __launch_bounds__(512) void gemm_skinny_kernel(const T* __restrict__ w, const T* __restrict__ x,
                                                          const float* __restrict__ bias, const float* res, void* out,
                                                          int out_f32, int act, int N, int K, int nb, int KS,
                                                          T* __restrict__ kc, T* __restrict__ vc,
                                                          const int* __restrict__ st, int max_seq, size_t slot_stride) {
    using MF = Mfma16<T>;
    using Frag = typename MF::Frag;
    __shared__ __attribute__((aligned(16))) float red[8][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int TR = 8 / KS, tile = wave / KS, ks = wave - tile * KS;
    const int n0 = (blockIdx.x * TR + tile) * 16;
    const int kw = K / KS;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    if (n0 < N) {
        const T* wr = w + (size_t)min(n0 + i, N - 1) * K + ks * kw + g * 16;
        const T* xr = x + (size_t)min(i, nb - 1) * K + ks * kw + g * 16;
        // UNR 64-wide K blocks per trip: all 4 * UNR loads are issued before the first MFMA
        for (int kb = 0; kb < kw; kb += 64 * UNR) {
            Frag a[2 * UNR], bq[2 * UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                a[2 * u] = *reinterpret_cast<const Frag*>(wr + kb + 64 * u);
                a[2 * u + 1] = *reinterpret_cast<const Frag*>(wr + kb + 64 * u + 8);
                bq[2 * u] = *reinterpret_cast<const Frag*>(xr + kb + 64 * u);
                bq[2 * u + 1] = *reinterpret_cast<const Frag*>(xr + kb + 64 * u + 8);
            }
#pragma unroll
            for (int u = 0; u < 2 * UNR; ++u) acc = MF::mma(a[u], bq[u], acc);
        }
    }
    *reinterpret_cast<f32x4_t*>(&red[wave][lane * 4]) = acc;
    __syncthreads();
    if (ks != 0 || n0 >= N) return;
    for (int q = 1; q < KS; ++q) {
        const f32x4_t o = *reinterpret_cast<const f32x4_t*>(&red[wave + q][lane * 4]);
        acc += o;
    }
    const int b = i;                                    // D layout: column = lane & 15 (sentence), rows 4*(lane>>4) + r
    if (b >= nb) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + 4 * g + r;
        if (n >= N) continue;
        float v = acc[r] + (bias ? bias[n] : 0.f);
        if (act == ACT_GELU_TANH) v = gelu_new(v);
        if (res) v += res[(size_t)b * N + n];
        bool to_cache = false;
        if (QKV) {
            const int hidden = N / 3;
            if (n >= hidden) {
                const int c = n - hidden, which = c / hidden, cc = c % hidden;
                const int pos = st[b * GS_WORDS + GS_HIST];
                if (pos < max_seq)
                    (which ? vc : kc)[(size_t)b * slot_stride + ((size_t)(cc >> 6) * max_seq + pos) * 64 + (cc & 63)] = (T)v;
                to_cache = true;
            }
        }
        if (!to_cache) {
            if (out_f32) ((float*)out)[(size_t)b * N + n] = v; else ((T*)out)[(size_t)b * N + n] = (T)v;
        }
    }
}

// rows of (q | k | v) -> K / V cache rows hist .. hist+rows-1 of one layer
template <typename T>
__global__ __launch_bounds__(256) void kv_append_kernel(const T* __restrict__ qkv, T* __restrict__ kc, T* __restrict__ vc,
                                                        const int* __restrict__ st, int rows, int hidden, int max_seq) {
    const int per = 2 * hidden / 8;                       // 16-byte groups for T = 2 bytes, 8 elements in general
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * per) return;
    const int r = (int)(i / per), c = (int)(i % per) * 8;
    const int which = c / hidden, cc = c % hidden, head = cc >> 6, d = cc & 63;
    const int pos = st[GS_HIST] + r;
    if (pos >= max_seq) return;
    const T* src = qkv + (size_t)r * 3 * hidden + hidden + c;
    T* dst = (which ? vc : kc) + ((size_t)head * max_seq + pos) * 64 + d;
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[e] = src[e];
}

// one block per (head, new row): softmax(q K^T + mask) V over the cache.  lane-per-key dot products (each lane streams
// one 64-wide key row with 16-byte loads), then 8 lanes per value row.
template <typename T>
__global__ __launch_bounds__(512) void gpt_attn_kernel(const T* __restrict__ qkv, const T* __restrict__ kc,
                                                       const T* __restrict__ vc, T* __restrict__ out,
                                                       const int* __restrict__ st, int rows, int flag, int hidden,
                                                       int max_seq, int batched, size_t slot_stride) {
    constexpr int V = Pack16<T>::N;            // elements per 16 bytes
    constexpr int CH = 64 / V;                 // 16-byte chunks per key row
    extern __shared__ float sm[];
    float* sc = sm;                            // [max_seq]
    float* qs = sm + max_seq;                  // [64]
    float* red = qs + 64;                      // [8 * 64]
    __shared__ float bc[2];
    // batched decode: blockIdx.y is the slot (one new row each, row index 0 inside its block); otherwise the new row
    const int head = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.y, i = batched ? 0 : row;
    if (batched) { st += row * GS_WORDS; kc += (size_t)row * slot_stride; vc += (size_t)row * slot_stride; }
    const int hist = st[GS_HIST];
    const int kv = min(hist + rows, max_seq);
    if (tid < 64) qs[tid] = (float)qkv[(size_t)row * 3 * hidden + head * 64 + tid];
    __syncthreads();
    const T* kb = kc + (size_t)head * max_seq * 64;
    const T* vb = vc + (size_t)head * max_seq * 64;
    float mx = -3.0e38f;
    for (int j = tid; j < kv; j += 512) {
        const T* kr = kb + (size_t)j * 64;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const Pack16<T> p = ld16(kr + c * V);
#pragma unroll
            for (int e = 0; e < V; ++e) s = fmaf(qs[c * V + e], (float)p.v[e], s);
        }
        if (flag && j > i) s += -128.f;
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    if (tid == 0) { float m = red[0]; for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w]); bc[0] = m; }
    __syncthreads();
    mx = bc[0];
    float sum = 0.f;
    for (int j = tid; j < kv; j += 512) { const float e = __expf(sc[j] - mx); sc[j] = e; sum += e; }
    sum = wave_sum(sum);
    __syncthreads();
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    if (tid == 0) { float t = 0.f; for (int w = 0; w < 8; ++w) t += red[w]; bc[1] = t; }
    __syncthreads();
    const float inv = 1.f / bc[1];
    // values: lane = (key-in-group g, chunk c) ; a wave covers 64/CH keys per pass
    constexpr int KPW = 64 / CH;               // keys per wave pass
    const int g = lane / CH, c = lane % CH;
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    for (int j = wave * KPW + g; j < kv; j += 8 * KPW) {
        const float p = sc[j];
        const Pack16<T> v = ld16(vb + (size_t)j * 64 + c * V);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = fmaf(p, (float)v.v[e], acc[e]);
    }
    // reduce over g (lanes that share c): xor over the g bits
#pragma unroll
    for (int o = CH; o < 64; o <<= 1)
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
    __syncthreads();
    if (g == 0)
#pragma unroll
        for (int e = 0; e < V; ++e) red[wave * 64 + c * V + e] = acc[e];
    __syncthreads();
    if (tid < 64) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += red[w * 64 + tid];
        out[(size_t)row * hidden + head * 64 + tid] = (T)(t * inv);
    }
}

// decode-step attention (one new row per sentence, no mask): one block of 4 waves per (head, slot).  Each wave owns
// every fourth 64-key pass and runs its own online softmax (lane = key for the scores, 16-byte K loads against the
// packed q; the pass's V rows are fetched before the softmax so their latency overlaps it), so the block needs ONE
// barrier: the four (max, sum, O) partials meet in LDS at the end.  The general kernel above (scores for all keys in
// LDS, three block-wide reductions) took 6.5 us per launch at 100-300 keys; this is the latency-critical launch of a
// decode step (24 per token).
template <typename T>
__global__ __launch_bounds__(256) void gpt_attn1_kernel(const T* __restrict__ qkv, const T* __restrict__ kc,
                                                        const T* __restrict__ vc, T* __restrict__ out,
                                                        const int* __restrict__ st, int hidden, int max_seq, int batched,
                                                        size_t slot_stride) {
    constexpr int V = Pack16<T>::N;            // elements per 16 bytes
    constexpr int CH = 64 / V;                 // 16-byte chunks per 64-wide row
    constexpr int KPI = 64 / CH;               // value rows per wave instruction
    constexpr int NIT = 64 / KPI;              // instructions per 64-key pass
    __shared__ float part[4][66];
    const int head = blockIdx.x, slot = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (batched) { st += slot * GS_WORDS; kc += (size_t)slot * slot_stride; vc += (size_t)slot * slot_stride; }
    const int kv = min(st[GS_HIST] + 1, max_seq);
    Pack16<T> q[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) q[c] = ld16(qkv + (size_t)slot * 3 * hidden + head * 64 + c * V);
    const T* kb = kc + (size_t)head * max_seq * 64;
    const T* vb = vc + (size_t)head * max_seq * 64;
    const int g = lane / CH, c = lane % CH;
    float m_run = -INFINITY, l_run = 0.f, acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    struct Pass { Pack16<T> kp[CH], vp[NIT]; };
    auto load_pass = [&](Pass& ps, int base) {
        const int j = base + lane;
        const T* kr = kb + (size_t)(j < kv ? j : base) * 64;
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) ps.kp[cc] = ld16(kr + cc * V);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int jj = base + it * KPI + g;
            ps.vp[it] = ld16(vb + (size_t)(jj < kv ? jj : base) * 64 + c * V);
        }
    };
    auto fold_pass = [&](const Pass& ps, int base) {
        const bool kok = base + lane < kv;
        float s = 0.f;
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) s = dot_pack(ps.kp[cc], q[cc], s);
        s = kok ? s : -INFINITY;
        const float mw = wave_max(s);
        const float m_new = fmaxf(m_run, mw);              // finite: the pass holds at least one key
        const float pj = kok ? __expf(s - m_new) : 0.f;
        const float corr = __expf(m_run - m_new);          // exp(-inf) = 0 on the first pass
        l_run = l_run * corr + wave_sum(pj);
        m_run = m_new;
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] *= corr;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const float pk = __shfl(pj, it * KPI + g, 64);  // 0 for keys beyond kv
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] = fmaf(pk, (float)ps.vp[it].v[e], acc[e]);
        }
    };
    // the wave's passes (every fourth 64-key block), the next one in flight while the current one is folded in
    int base = wave * 64;
    if (base < kv) {
        Pass pa, pb;
        load_pass(pa, base);
        for (;;) {
            bool more = base + 256 < kv;
            if (more) load_pass(pb, base + 256);
            fold_pass(pa, base);
            if (!more) break;
            base += 256;
            more = base + 256 < kv;
            if (more) load_pass(pa, base + 256);
            fold_pass(pb, base);
            if (!more) break;
            base += 256;
        }
    }
#pragma unroll
    for (int o = CH; o < 64; o <<= 1)
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
    if (lane == 0) { part[wave][0] = m_run; part[wave][1] = l_run; }
    if (g == 0)
#pragma unroll
        for (int e = 0; e < V; ++e) part[wave][2 + c * V + e] = acc[e];
    __syncthreads();
    if (wave == 0) {
        float M = part[0][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) M = fmaxf(M, part[w][0]);
        float L = 0.f, o = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = __expf(part[w][0] - M);        // waves without keys: exp(-inf) = 0
            L = fmaf(part[w][1], f, L);
            o = fmaf(part[w][2 + lane], f, o);
        }
        out[(size_t)slot * hidden + head * 64 + lane] = (T)(o / L);
    }
}

__global__ __launch_bounds__(256) void gpt_text_embed_kernel(const int* __restrict__ ids, const float* __restrict__ emb,
                                                             const float* __restrict__ pos, float* __restrict__ out,
                                                             int n, int hidden, int vocab) {
    const int r = blockIdx.x;                  // 0 .. n+1
    int id = r == 0 ? 0 : (r == n + 1 ? 1 : ids[r - 1]);
    id = min(max(id, 0), vocab - 1);
    for (int c = threadIdx.x; c < hidden; c += 256)
        out[(size_t)r * hidden + c] = emb[(size_t)id * hidden + c] + pos[(size_t)r * hidden + c];
}

// hidden state of the next decode step from the device state (graph C fed by the previous step's max_logit_id)
__global__ __launch_bounds__(256) void gpt_embed_state_kernel(const int* __restrict__ st, const float* __restrict__ emb,
                                                              const float* __restrict__ pos, float* __restrict__ x,
                                                              int hidden, int codes, int max_pos, int id_arg, int gen_arg) {
    st += blockIdx.x * GS_WORDS;               // one block per slot
    x += (size_t)blockIdx.x * hidden;
    const int id = min(max(id_arg >= 0 ? id_arg : st[GS_TOKEN], 0), codes - 1);
    const int g = min(max(gen_arg >= 0 ? gen_arg : st[GS_GEN_LEN], 0), max_pos - 1);
    for (int c = threadIdx.x; c < hidden; c += 256) x[c] = emb[(size_t)id * hidden + c] + pos[(size_t)g * hidden + c];
}

// logits * penalty -> argmax (first index on ties, like torch.argmax) -> the driver loop's bookkeeping
__global__ __launch_bounds__(1024) void gpt_pick_kernel(const float* __restrict__ logits, float* __restrict__ pen,
                                                        const float* __restrict__ last, int* __restrict__ st,
                                                        int* __restrict__ toks, float* __restrict__ hid, int codes,
                                                        int hidden, int rows, const float* __restrict__ rep_dev, int max_tok,
                                                        const float* __restrict__ emb, const float* __restrict__ pos,
                                                        int max_pos, float* xa, float* xb) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    __shared__ int slot;
    {   // one block per sentence slot
        const size_t sl = blockIdx.x;
        logits += sl * codes; pen += sl * codes; last += sl * hidden; st += sl * GS_WORDS; toks += sl * max_tok;
        hid += sl * (size_t)max_tok * hidden;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int next_id, next_gen;
    // everything that does not depend on the argmax is requested first (the kernel is a chain of dependent round trips: 8.4 us
    // for 33 KB of logits): the state words, the penalty scalar, the oldest penalised token, the last_hidden_state row
    int w[GS_WORDS];
    float repv = 0.f;
    int tok_r = 0;
    if (tid == 0) {
#pragma unroll
        for (int q = 0; q < GS_WORDS; q += 4) {
            const int4 v = *reinterpret_cast<const int4*>(st + q);
            w[q] = v.x; w[q + 1] = v.y; w[q + 2] = v.z; w[q + 3] = v.w;
        }
        repv = rep_dev[0];
        const int r = w[GS_RESET];
        tok_r = (r >= 0 && r < max_tok) ? toks[r] : 0;
    }
    float lastv[2];                                        // hidden <= 2048
#pragma unroll
    for (int q = 0; q < 2; ++q) lastv[q] = tid + q * 1024 < hidden ? last[tid + q * 1024] : 0.f;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int c = tid; c < codes; c += 1024) {
        const float v = logits[c] * pen[c];
        if (v > best || (v == best && c < idx)) { best = v; idx = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if (lane == 0) { bv[wave] = best; bi[wave] = idx; }
    __syncthreads();
    if (tid == 0) {
        for (int q = 1; q < 16; ++q)
            if (bv[q] > best || (bv[q] == best && bi[q] < idx)) { best = bv[q]; idx = bi[q]; }
        if (idx == 0x7fffffff) idx = 0;
        slot = -1;
        if (!w[GS_DONE]) {
            const int t = idx, n = w[GS_NDEC];
            w[GS_TOKEN] = t;
            if (n < max_tok) { toks[n] = t; slot = n; }
            w[GS_NDEC] = n + 1;
            bool stop = false;
#pragma unroll
            for (int q = 0; q < GS_WORDS - GS_STOP0; ++q) stop |= (q < w[GS_NSTOP] && w[GS_STOP0 + q] == t);
            if (stop) w[GS_DONE] = 1;
            else if (w[GS_UPDATE_PEN]) {                      // Inference_IndexTTS_ONNX.py:768-772
                pen[t] = repv;            // device scalar: the captured decode graphs must see a changed REPEAT_PENALITY
                const int r = w[GS_RESET];
                // toks[r] was fetched before toks[n] = t above: the same element only if r == n
                const int tr = (r == n && n < max_tok) ? t : tok_r;
                if (n + 1 > w[GS_RANGE] && r < max_tok && tr != t) { pen[tr] = 1.f; w[GS_RESET] = r + 1; }
            }
            w[GS_HIST] += rows;
            w[GS_GEN_LEN] += 1;
            if (w[GS_LIMIT] > 0 && n + 1 >= w[GS_LIMIT]) w[GS_DONE] = 1;      // `while num_decode < generate_limit`
#pragma unroll
            for (int q = 0; q < GS_STOP0; q += 4) *reinterpret_cast<int4*>(st + q) = make_int4(w[q], w[q + 1], w[q + 2], w[q + 3]);
            if (GS_STOP0 % 4) { for (int q = GS_STOP0 / 4 * 4; q < GS_STOP0; ++q) st[q] = w[q]; }
        }
        next_id = w[GS_TOKEN]; next_gen = w[GS_GEN_LEN];
    }
    __syncthreads();
    if (slot >= 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (tid + q * 1024 < hidden) hid[(size_t)slot * hidden + tid + q * 1024] = lastv[q];
    }
    // graph C for the next decode step (IndexTTS_C.forward, Export_IndexTTS.py:222-225) from the state just written:
    // the step's input row, so that a decode step does not start with a launch of its own for it
    const int id = min(max(next_id, 0), codes - 1), g = min(max(next_gen, 0), max_pos - 1);
    for (int c = tid; c < hidden; c += 1024) {
        const float v = emb[(size_t)id * hidden + c] + pos[(size_t)g * hidden + c];
        if (xa) xa[c] = v;
        if (xb) xb[(size_t)blockIdx.x * hidden + c] = v;
    }
}

// cache <-> the reference's tensor layouts: keys (H, D, hist), values (H, hist, D), fp32
template <typename T>
__global__ __launch_bounds__(256) void kv_export_kernel(const T* __restrict__ kc, const T* __restrict__ vc,
                                                        float* __restrict__ keys, float* __restrict__ values, int H,
                                                        int hist, int max_seq) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)H * hist * 64) return;
    const int d = (int)(i & 63), j = (int)((i >> 6) % hist), h = (int)((i >> 6) / hist);
    const size_t src = ((size_t)h * max_seq + j) * 64 + d;
    if (keys) keys[((size_t)h * 64 + d) * hist + j] = (float)kc[src];
    if (values) values[((size_t)h * hist + j) * 64 + d] = (float)vc[src];
}
template <typename T>
__global__ __launch_bounds__(256) void kv_import_kernel(T* __restrict__ kc, T* __restrict__ vc, const float* __restrict__ keys,
                                                        const float* __restrict__ values, int H, int hist, int max_seq) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)H * hist * 64) return;
    const int d = (int)(i & 63), j = (int)((i >> 6) % hist), h = (int)((i >> 6) / hist);
    const size_t dst = ((size_t)h * max_seq + j) * 64 + d;
    kc[dst] = (T)keys[((size_t)h * 64 + d) * hist + j];
    vc[dst] = (T)values[((size_t)h * hist + j) * 64 + d];
}

// ---------------------------------------------------------------------------------------------------------------
// engine
// ---------------------------------------------------------------------------------------------------------------
static void up_glin(Gpt::GLin& l, const float*& p, int n, int k, int dt, hipStream_t s) {
    l.n = n; l.k = k;
    upload_as(l.w, p, (size_t)n * k, dt, s); p += (size_t)n * k;
    upload_f32(l.b, p, n, s); p += n;
}
static void up_vec(DevBuf& b, const float*& p, size_t n, hipStream_t s) { upload_f32(b, p, n, s); p += n; }

Gpt::Gpt(const GptCfg& c, const float* w, int64_t nw, int dt, int dev) : cfg(c), dtype(dt), device(dev) {
    MI_REQUIRE(dt == MI_F32 || dt == MI_F16 || dt == MI_BF16, "gpt: dtype");
    MI_REQUIRE(nw == gpt_param_count(c), "gpt: weight blob size does not match the config");
    MI_HIP(hipSetDevice(dev));
    MI_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    hipStream_t s = stream;
    const int h = c.hidden, n = c.inner;
    const float* p = w;
    up_vec(text_emb, p, (size_t)c.text_tokens * h, s);
    up_vec(text_pos, p, (size_t)c.max_text_pos * h, s);
    up_vec(mel_emb, p, (size_t)c.mel_codes * h, s);
    up_vec(mel_pos, p, (size_t)c.max_mel_pos * h, s);
    L.resize(c.layers);
    for (auto& l : L) {
        up_vec(l.ln1_w, p, h, s); up_vec(l.ln1_b, p, h, s);
        up_glin(l.qkv, p, 3 * h, h, dt, s);
        up_glin(l.proj, p, h, h, dt, s);
        up_vec(l.ln2_w, p, h, s); up_vec(l.ln2_b, p, h, s);
        up_glin(l.fc, p, n, h, dt, s);
        up_glin(l.fc2, p, h, n, dt, s);
    }
    up_vec(lnf_w, p, h, s); up_vec(lnf_b, p, h, s);
    up_vec(fn_w, p, h, s); up_vec(fn_b, p, h, s);
    up_glin(head, p, c.mel_codes, h, dt, s);
    MI_REQUIRE(p - w == nw, "gpt: blob walk mismatch");

    const size_t es = dtype_size(dt), S = c.max_seq, MB = c.max_batch;
    MBp = c.max_batch <= 1 ? 1 : c.max_batch <= 2 ? 2 : c.max_batch <= 4 ? 4 : c.max_batch <= 8 ? 8 : 16;
    kc.ensure(MB * slot_cache_elems() * es); vc.ensure(MB * slot_cache_elems() * es);
    X.ensure(S * h * 4); xn.ensure(S * h * es); qkv.ensure(S * 3 * h * es); att.ensure(S * h * es); ff.ensure(S * n * es);
    const size_t P = MBp;                       // padded slot rows of the batched decode scratch
    logits.ensure(P * c.mel_codes * 4); last.ensure(P * h * 4); pen.ensure(P * c.mel_codes * 4);
    toks.ensure(MB * S * 4); hid.ensure(MB * S * h * 4); state.ensure(P * GS_WORDS * 4);
    Xd.ensure(P * h * 4); xnd.ensure(P * h * es); qkvd.ensure(P * 3 * h * es); attd.ensure(P * h * es);
    ffd.ensure(P * n * es); zd.ensure(P * h * es);
    for (DevBuf* bfr : {&kc, &vc, &state, &Xd, &xnd, &qkvd, &attd, &ffd, &zd, &last, &logits})
        MI_HIP(hipMemsetAsync(bfr->p, 0, bfr->bytes, s));
    std::vector<float> ones(P * c.mel_codes, 1.f);
    MI_HIP(hipMemcpyAsync(pen.p, ones.data(), ones.size() * 4, hipMemcpyHostToDevice, s));
    rep_dev.ensure(4);
    MI_HIP(hipStreamSynchronize(s));
    set_rep_value(0.7f);
    const char* ng = std::getenv("MI355TTS_NO_GRAPH");
    use_graph = !(ng && ng[0] == '1');
    // the attention kernel's score buffer is dynamic LDS: max_seq + 64 + 512 floats
    const int lds = (c.max_seq + 64 + 512) * 4;
    MI_HIP(hipFuncSetAttribute((const void*)gpt_attn_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    MI_HIP(hipFuncSetAttribute((const void*)gpt_attn_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    MI_HIP(hipFuncSetAttribute((const void*)gpt_attn_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
}

Gpt::~Gpt() {
    if (step_graph) (void)hipGraphExecDestroy(step_graph);
    for (auto& kv : batch_graphs) if (kv.second) (void)hipGraphExecDestroy(kv.second);
    if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
}

void Gpt::set_state(const std::vector<int32_t>& words, int slot) {
    MI_REQUIRE(words.size() == GS_WORDS && slot >= 0 && slot < cfg.max_batch, "gpt: state slot");
    MI_HIP(hipMemcpyAsync(state.as<int32_t>() + (size_t)slot * GS_WORDS, words.data(), GS_WORDS * 4, hipMemcpyHostToDevice, stream));
    MI_HIP(hipStreamSynchronize(stream));      // `words` may be a temporary
    if (slot == 0) history = words[GS_HIST];
}
std::vector<int32_t> Gpt::get_state(int slot) {
    MI_REQUIRE(slot >= 0 && slot < cfg.max_batch, "gpt: state slot");
    std::vector<int32_t> w(GS_WORDS);
    MI_HIP(hipMemcpyAsync(w.data(), state.as<int32_t>() + (size_t)slot * GS_WORDS, GS_WORDS * 4, hipMemcpyDeviceToHost, stream));
    MI_HIP(hipStreamSynchronize(stream));
    if (slot == 0) history = w[GS_HIST];
    return w;
}
void Gpt::reset() {
    std::vector<int32_t> w(GS_WORDS, 0);
    set_state(w, 0);
}

void Gpt::linear(const GLin& l, const void* x, int rows, void* out, int odt, int act, const float* res) {
    MI_REQUIRE(rows > 1, "gpt: linear() is the multi-row path");
    ConvGemm g;
    g.dtype = dtype; g.out_dtype = odt; g.x = x; g.w = l.w.p; g.bias = l.b.as<float>(); g.out = out; g.res = res;
    g.B = 1; g.T_in = rows; g.M = rows; g.N = l.n; g.Cin = l.k; g.taps = 1;
    g.x_bstride = (long)rows * l.k; g.x_rstride = l.k; g.out_bstride = (long)rows * l.n; g.out_rstride = l.n; g.act = act;
    launch_conv_gemm(g, stream);
}

// single-row linear layer: weight-streaming GEMV (gemv_d_kernel).  ln_w != nullptr: x is the fp32 residual row,
// LayerNorm fused in (pre_w: a first LayerNorm in front of it, its row also stored to pre_out).  kcl != nullptr: QKV
// epilogue (k and v rows go to the cache).
template <typename T, int R, int KI, bool LN, bool QKV>
static void gemv_d_launch(const Gpt::GLin& l, const void* x, const float* ln_w, const float* ln_b, const float* pre_w,
                          const float* pre_b, float* pre_out, const float* res, void* out, int of, int act, void* kcl,
                          void* vcl, const int* st, int max_seq, hipStream_t s) {
    constexpr int WAVES = LN ? 8 : 4;
    const dim3 grid((unsigned)((l.n + WAVES * R - 1) / (WAVES * R)));
    hipLaunchKernelGGL((gemv_d_kernel<T, R, KI, WAVES, LN, QKV>), grid, dim3(WAVES * 64), 0, s, (const T*)l.w.p, x, ln_w,
                       ln_b, pre_w, pre_b, pre_out, l.b.as<float>(), res, out, of, act, l.n, l.k, (T*)kcl, (T*)vcl, st, max_seq);
}
template <typename T>
static void gemv_d_dispatch(const Gpt::GLin& l, const void* x, const float* ln_w, const float* ln_b, const float* pre_w,
                            const float* pre_b, float* pre_out, const float* res, void* out, int of, int act, void* kcl,
                            void* vcl, const int* st, int max_seq, hipStream_t s) {
    constexpr int V = 16 / (int)sizeof(T);
    const int ki = (l.k + 64 * V - 1) / (64 * V);
#define GD(RR, KK, LNN, QK) gemv_d_launch<T, RR, KK, LNN, QK>(l, x, ln_w, ln_b, pre_w, pre_b, pre_out, res, out, of, act, kcl, vcl, st, max_seq, s)
    if (ln_w) {
        MI_REQUIRE(ki <= 8, "gemv: the fused LayerNorm supports K <= 2048");
        // rows per wave: the fewest (most waves in flight) for which the launch is one round of 512-thread blocks, one per CU
        // (5120 rows at R = 2 are 320 blocks on 256 CUs: the 64 CUs that get two set the time — 6.1 us against 4.7 for the
        // 3840-row layer; R = 3: 214 blocks, 5.4 us)
        static int cus = 0;
        if (!cus) { int dev = 0; hipDeviceProp_t pr; MI_HIP(hipGetDevice(&dev)); MI_HIP(hipGetDeviceProperties(&pr, dev)); cus = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
        int r = 1;
        while (r < (kcl ? 2 : 5) && (l.n + 8 * r - 1) / (8 * r) > cus) ++r;
#define GD_K(RR, QK)                                                                              \
    do {                                                                                          \
        if (ki <= 1) GD(RR, 1, true, QK); else if (ki == 2) GD(RR, 2, true, QK);                   \
        else if (ki == 3) GD(RR, 3, true, QK); else if (ki == 4) GD(RR, 4, true, QK);              \
        else if (ki == 5) GD(RR, 5, true, QK); else if (ki == 6) GD(RR, 6, true, QK);              \
        else GD(RR, 8, true, QK);                                                                 \
    } while (0)
        if (kcl) { if (r == 2) GD_K(2, true); else GD_K(1, true); }
        else { if (r == 1) GD_K(1, false); else if (r == 2) GD_K(2, false); else if (r == 3) GD_K(3, false); else if (r == 4) GD_K(4, false); else GD_K(5, false); }
#undef GD_K
    } else {
        MI_REQUIRE(ki <= 32, "gemv: K <= 8192");
        if (ki <= 1) GD(1, 1, false, false); else if (ki == 2) GD(1, 2, false, false);
        else if (ki == 3) GD(1, 3, false, false); else if (ki == 4) GD(1, 4, false, false);
        else if (ki == 5) GD(1, 5, false, false); else if (ki == 6) GD(1, 6, false, false);
        else if (ki <= 8) GD(1, 8, false, false); else if (ki <= 10) GD(1, 10, false, false);
        else if (ki <= 12) GD(1, 12, false, false); else if (ki <= 16) GD(1, 16, false, false);
        else if (ki <= 20) GD(1, 20, false, false); else GD(1, 32, false, false);
    }
#undef GD
}

void Gpt::gemv(const GLin& l, const void* x, const float* ln_w, const float* ln_b, void* out, int odt, int act,
               const float* res, void* kcl, void* vcl, int slot, const float* pre_w, const float* pre_b, float* pre_out) {
    MI_REQUIRE(l.k % 8 == 0, "gemv: K must be a multiple of 8");
    const int of = odt == MI_F32;
    MI_REQUIRE(of || odt == dtype, "gemv: output dtype");
    MI_REQUIRE(!kcl || ln_w, "gemv: the QKV epilogue comes with the fused LayerNorm");
    MI_REQUIRE(!ln_w || !res, "gemv: the fused-LayerNorm variant has no residual input");
    MI_REQUIRE(!pre_w || ln_w, "gemv: the leading LayerNorm comes with the fused one");
    ProfScope ps(FAM_CONV_GEMM, stream, (double)l.n * l.k * dtype_size(dtype), 2.0 * l.n * l.k);
    const int* st = state.as<int>() + (size_t)slot * GS_WORDS;
    if (dtype == MI_F32) gemv_d_dispatch<float>(l, x, ln_w, ln_b, pre_w, pre_b, pre_out, res, out, of, act, kcl, vcl, st, cfg.max_seq, stream);
    else if (dtype == MI_F16) gemv_d_dispatch<f16>(l, x, ln_w, ln_b, pre_w, pre_b, pre_out, res, out, of, act, kcl, vcl, st, cfg.max_seq, stream);
    else gemv_d_dispatch<bf16>(l, x, ln_w, ln_b, pre_w, pre_b, pre_out, res, out, of, act, kcl, vcl, st, cfg.max_seq, stream);
    MI_HIP(hipGetLastError());
}

#define GPT_DISPATCH(KERNEL, ...)                                                     \
    do {                                                                              \
        if (dtype == MI_F32) { KERNEL(float, __VA_ARGS__); }                          \
        else if (dtype == MI_F16) { KERNEL(f16, __VA_ARGS__); }                       \
        else { KERNEL(bf16, __VA_ARGS__); }                                           \
        MI_HIP(hipGetLastError());                                                    \
    } while (0)

void Gpt::forward_rows(int rows, int flag, int slot) {
    const GptCfg& c = cfg;
    const int h = c.hidden, S = c.max_seq;
    hipStream_t s = stream;
    const size_t es = dtype_size(dtype);
    MI_REQUIRE(slot >= 0 && slot < c.max_batch, "gpt: slot");
    const int* st = state.as<int>() + (size_t)slot * GS_WORDS;
    float* x = X.as<float>();
    const int lds = (S + 64 + 512) * 4;
    float* last_s = last.as<float>() + (size_t)slot * h;
    float* logits_s = logits.as<float>() + (size_t)slot * c.mel_codes;
    for (int li = 0; li < c.layers; ++li) {
        Layer& l = L[li];
        char* kcl = (char*)kc.p + ((size_t)slot * slot_cache_elems() + (size_t)li * h * S) * es;
        char* vcl = (char*)vc.p + ((size_t)slot * slot_cache_elems() + (size_t)li * h * S) * es;
        if (rows == 1) {
            gemv(l.qkv, x, l.ln1_w.as<float>(), l.ln1_b.as<float>(), qkv.p, dtype, ACT_NONE, nullptr, kcl, vcl, slot);
        } else {
            launch_rownorm(NORM_LN_AFFINE, x, xn.p, dtype, l.ln1_w.as<float>(), l.ln1_b.as<float>(), rows, h, 1e-5f, s);
            linear(l.qkv, xn.p, rows, qkv.p, dtype, ACT_NONE, nullptr);
            const long items = (long)rows * (2 * h / 8);
            const dim3 grid((unsigned)((items + 255) / 256));
#define KVA(T, ...) hipLaunchKernelGGL(kv_append_kernel<T>, grid, dim3(256), 0, s, (const T*)qkv.p, (T*)kcl, (T*)vcl, st, rows, h, S)
            GPT_DISPATCH(KVA, 0);
#undef KVA
        }
        {
            ProfScope ps(FAM_ATTN, s, 2.0 * (double)h * (history + rows) * es, 4.0 * (double)h * rows * (history + rows));
#define ATT(T, ...) hipLaunchKernelGGL(gpt_attn_kernel<T>, dim3(c.heads, rows), dim3(512), lds, s, (const T*)qkv.p, (const T*)kcl, (const T*)vcl, (T*)att.p, st, rows, flag, h, S, 0, (size_t)0)
#define ATT1(T, ...) hipLaunchKernelGGL(gpt_attn1_kernel<T>, dim3(c.heads, 1), dim3(256), 0, s, (const T*)qkv.p, (const T*)kcl, (const T*)vcl, (T*)att.p, st, h, S, 0, (size_t)0)
            if (rows == 1 && !flag) GPT_DISPATCH(ATT1, 0); else GPT_DISPATCH(ATT, 0);
#undef ATT1
#undef ATT
        }
        if (rows == 1) {
            gemv(l.proj, att.p, nullptr, nullptr, x, MI_F32, ACT_NONE, x, nullptr, nullptr);
            gemv(l.fc, x, l.ln2_w.as<float>(), l.ln2_b.as<float>(), ff.p, dtype, ACT_GELU_TANH, nullptr, nullptr, nullptr);
            gemv(l.fc2, ff.p, nullptr, nullptr, x, MI_F32, ACT_NONE, x, nullptr, nullptr);
        } else {
            linear(l.proj, att.p, rows, x, MI_F32, ACT_NONE, x);
            launch_rownorm(NORM_LN_AFFINE, x, xn.p, dtype, l.ln2_w.as<float>(), l.ln2_b.as<float>(), rows, h, 1e-5f, s);
            linear(l.fc, xn.p, rows, ff.p, dtype, ACT_GELU_TANH, nullptr);
            linear(l.fc2, ff.p, rows, x, MI_F32, ACT_NONE, x);
        }
    }
    const float* xl = x + (size_t)(rows - 1) * h;
    // ln_f (-> last_hidden_state) and final_norm in front of the lm_head, one launch
    gemv(head, xl, fn_w.as<float>(), fn_b.as<float>(), logits_s, MI_F32, ACT_NONE, nullptr, nullptr, nullptr, 0,
         lnf_w.as<float>(), lnf_b.as<float>(), last_s);
    hipLaunchKernelGGL(gpt_pick_kernel, dim3(1), dim3(1024), 0, s, logits_s, pen.as<float>() + (size_t)slot * c.mel_codes,
                       last_s, state.as<int>() + (size_t)slot * GS_WORDS, toks.as<int>() + (size_t)slot * S,
                       hid.as<float>() + (size_t)slot * S * h, c.mel_codes, h, rows, rep_dev.as<float>(), S,
                       mel_emb.as<float>(), mel_pos.as<float>(), c.max_mel_pos, slot == 0 ? X.as<float>() : nullptr,
                       Xd.as<float>() + (size_t)slot * h);
    MI_HIP(hipGetLastError());
}

void Gpt::set_rep_value(float v) {
    // synchronous on purpose: `v` lives on the caller's stack
    MI_HIP(hipMemcpyAsync(rep_dev.p, &v, 4, hipMemcpyHostToDevice, stream));
    MI_HIP(hipStreamSynchronize(stream));
}

// X row 0 holds graph C's output for the state: written by the pick kernel of the step (or prompt pass) before
void Gpt::decode_step_eager() { forward_rows(1, 0); }

void Gpt::check_graph_epoch() {
    if (graph_epoch == option_epoch()) return;
    if (step_graph) { (void)hipGraphExecDestroy(step_graph); step_graph = nullptr; }
    for (auto& kv : batch_graphs) if (kv.second) (void)hipGraphExecDestroy(kv.second);
    batch_graphs.clear();
    graph_epoch = option_epoch();
}

void Gpt::decode_steps(int n) {
    if (n <= 0) return;
    if (!use_graph || prof_mask() != 0) { for (int i = 0; i < n; ++i) decode_step_eager(); return; }
    check_graph_epoch();
    if (!step_graph) {
        decode_step_eager();                   // first step eager (one-time lazy initialisation stays out of the capture)
        --n;
        hipGraph_t graph = nullptr;
        MI_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        try {
            decode_step_eager();
        } catch (...) {
            (void)hipStreamEndCapture(stream, &graph);
            if (graph) (void)hipGraphDestroy(graph);
            throw;
        }
        MI_HIP(hipStreamEndCapture(stream, &graph));
        hipError_t err = hipGraphInstantiate(&step_graph, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (err != hipSuccess) { step_graph = nullptr; use_graph = false; for (int i = 0; i < n; ++i) decode_step_eager(); return; }
    }
    for (int i = 0; i < n; ++i) MI_HIP(hipGraphLaunch(step_graph, stream));
}

// batched GEMV dispatch: x (nb rows of K, engine dtype) -> out (nb rows of N)
void Gpt::gemv_b(const GLin& l, const void* x, int nb, void* out, int odt, int act, const float* res, void* kcl, void* vcl) {
    MI_REQUIRE(l.k % 8 == 0 && nb >= 1 && nb <= MBp, "gemv_b: shape");
    const int of = odt == MI_F32;
    MI_REQUIRE(of || odt == dtype, "gemv_b: output dtype");
    const int BB = nb <= 2 ? 2 : nb <= 4 ? 4 : nb <= 8 ? 8 : 16;
    const int R = l.n >= 4096 ? 2 : 1;
    ProfScope ps(FAM_CONV_GEMM, stream, (double)l.n * l.k * dtype_size(dtype), 2.0 * l.n * l.k * nb);
    const size_t sstride = slot_cache_elems();
    {   // matrix-core path: 16-bit engines, nine or more sentences, K splits into 64-wide blocks per wave
        static int no_mfma = -1;
        if (no_mfma < 0) { const char* e = std::getenv("MI355TTS_GPT_NO_MFMA"); no_mfma = (e && e[0] == '1') ? 1 : 0; }
        const int KS = l.k > 2048 ? 8 : 4;
        // measured (1280-wide model): the MFMA kernel costs ~9.5 us per launch whatever the batch, the v_dot2 GEMV
        // 10 / 12.7 / 20 us at 4 / 8 / 16 sentences
        if (!no_mfma && dtype != MI_F32 && nb >= g_gpt_mfma_min && nb <= 16 && l.k % (KS * 64) == 0) {
            const int TR = 8 / KS;
            const dim3 gs((unsigned)((l.n + 16 * TR - 1) / (16 * TR)));
            const bool u5 = (l.k / KS) % 320 == 0;
#define GS1(T, QK, U) hipLaunchKernelGGL((gemm_skinny_kernel<T, QK, U>), gs, dim3(512), 0, stream, (const T*)l.w.p, (const T*)x, l.b.as<float>(), res, out, of, act, l.n, l.k, nb, KS, (T*)kcl, (T*)vcl, state.as<int>(), cfg.max_seq, sstride)
#define GS(T, QK) do { if (u5) GS1(T, QK, 5); else GS1(T, QK, 1); } while (0)
            if (dtype == MI_F16) { if (kcl) GS(f16, true); else GS(f16, false); }
            else { if (kcl) GS(bf16, true); else GS(bf16, false); }
#undef GS
#undef GS1
            MI_HIP(hipGetLastError());
            return;
        }
    }
    const dim3 grid((unsigned)((l.n + 8 * R - 1) / (8 * R)));
#define GB(T, RR, B_, QK) hipLaunchKernelGGL((gemv_b_kernel<T, RR, B_, QK>), grid, dim3(512), 0, stream, (const T*)l.w.p, (const T*)x, l.b.as<float>(), res, out, of, act, l.n, l.k, nb, (T*)kcl, (T*)vcl, state.as<int>(), cfg.max_seq, sstride)
#define GB_B(T, RR, QK) do { if (BB == 2) GB(T, RR, 2, QK); else if (BB == 4) GB(T, RR, 4, QK); else if (BB == 8) GB(T, RR, 8, QK); else GB(T, RR, 16, QK); } while (0)
#define GB_T(T)                                                                 \
    do {                                                                        \
        if (kcl) { if (R == 2) GB_B(T, 2, true); else GB_B(T, 1, true); }        \
        else { if (R == 2) GB_B(T, 2, false); else GB_B(T, 1, false); }          \
    } while (0)
    if (dtype == MI_F32) GB_T(float); else if (dtype == MI_F16) GB_T(f16); else GB_T(bf16);
#undef GB_T
#undef GB_B
#undef GB
    MI_HIP(hipGetLastError());
}

// one decode step for slots 0..nb-1 together: graph C from each slot's state, graph E with every weight row streamed
// once for all slots, per-slot argmax / penalty / stop bookkeeping.  Finished slots keep running as no-ops.
void Gpt::decode_batch_eager(int nb) {
    const GptCfg& c = cfg;
    MI_REQUIRE(nb >= 1 && nb <= c.max_batch, "gpt: batch exceeds max_batch");
    const int h = c.hidden, S = c.max_seq;
    hipStream_t s = stream;
    const size_t es = dtype_size(dtype);
    float* x = Xd.as<float>();                 // row b = graph C of slot b's state, written by the pick kernel before
    for (int li = 0; li < c.layers; ++li) {
        Layer& l = L[li];
        char* kcl = (char*)kc.p + (size_t)li * h * S * es;          // slot 0's layer; slots are slot_cache_elems apart
        char* vcl = (char*)vc.p + (size_t)li * h * S * es;
        launch_rownorm(NORM_LN_AFFINE, x, xnd.p, dtype, l.ln1_w.as<float>(), l.ln1_b.as<float>(), nb, h, 1e-5f, s);
        gemv_b(l.qkv, xnd.p, nb, qkvd.p, dtype, ACT_NONE, nullptr, kcl, vcl);
        {
            ProfScope ps(FAM_ATTN, s, 0.0, 0.0);
#define ATT(T, ...) hipLaunchKernelGGL(gpt_attn1_kernel<T>, dim3(c.heads, nb), dim3(256), 0, s, (const T*)qkvd.p, (const T*)kcl, (const T*)vcl, (T*)attd.p, state.as<int>(), h, S, 1, slot_cache_elems())
            GPT_DISPATCH(ATT, 0);
#undef ATT
        }
        gemv_b(l.proj, attd.p, nb, x, MI_F32, ACT_NONE, x, nullptr, nullptr);
        launch_rownorm(NORM_LN_AFFINE, x, xnd.p, dtype, l.ln2_w.as<float>(), l.ln2_b.as<float>(), nb, h, 1e-5f, s);
        gemv_b(l.fc, xnd.p, nb, ffd.p, dtype, ACT_GELU_TANH, nullptr, nullptr, nullptr);
        gemv_b(l.fc2, ffd.p, nb, x, MI_F32, ACT_NONE, x, nullptr, nullptr);
    }
    launch_rownorm(NORM_LN_AFFINE, x, last.p, MI_F32, lnf_w.as<float>(), lnf_b.as<float>(), nb, h, 1e-5f, s);
    launch_rownorm(NORM_LN_AFFINE, last.as<float>(), zd.p, dtype, fn_w.as<float>(), fn_b.as<float>(), nb, h, 1e-5f, s);
    gemv_b(head, zd.p, nb, logits.p, MI_F32, ACT_NONE, nullptr, nullptr, nullptr);
    hipLaunchKernelGGL(gpt_pick_kernel, dim3(nb), dim3(1024), 0, s, logits.as<float>(), pen.as<float>(), last.as<float>(),
                       state.as<int>(), toks.as<int>(), hid.as<float>(), c.mel_codes, h, 1, rep_dev.as<float>(), S,
                       mel_emb.as<float>(), mel_pos.as<float>(), c.max_mel_pos, (float*)nullptr, Xd.as<float>());
    MI_HIP(hipGetLastError());
}

void Gpt::decode_batch_steps(int nb, int n) {
    if (n <= 0) return;
    if (!use_graph || prof_mask() != 0) { for (int i = 0; i < n; ++i) decode_batch_eager(nb); return; }
    check_graph_epoch();
    hipGraphExec_t& exec = batch_graphs[nb];
    if (!exec) {
        decode_batch_eager(nb);
        --n;
        hipGraph_t graph = nullptr;
        MI_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        try {
            decode_batch_eager(nb);
        } catch (...) {
            (void)hipStreamEndCapture(stream, &graph);
            if (graph) (void)hipGraphDestroy(graph);
            throw;
        }
        MI_HIP(hipStreamEndCapture(stream, &graph));
        hipError_t err = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (err != hipSuccess) { exec = nullptr; use_graph = false; for (int i = 0; i < n; ++i) decode_batch_eager(nb); return; }
    }
    for (int i = 0; i < n; ++i) MI_HIP(hipGraphLaunch(exec, stream));
}

void Gpt::text_embed(const int32_t* ids_dev, int n, float* out_dev) {
    MI_REQUIRE(n >= 0 && n + 2 <= cfg.max_text_pos, "gpt: text is longer than the text position table");
    hipLaunchKernelGGL(gpt_text_embed_kernel, dim3(n + 2), dim3(256), 0, stream, ids_dev, text_emb.as<float>(),
                       text_pos.as<float>(), out_dev, n, cfg.hidden, cfg.text_tokens);
    MI_HIP(hipGetLastError());
}

void Gpt::mel_embed(int32_t id, long gen_len, float* out_dev) {
    MI_REQUIRE(id >= 0 && id < cfg.mel_codes, "gpt: mel code out of range");
    MI_REQUIRE(gen_len >= 0 && gen_len < cfg.max_mel_pos, "gpt: gen_len exceeds the mel position table");
    hipLaunchKernelGGL(gpt_embed_state_kernel, dim3(1), dim3(256), 0, stream, state.as<int>(), mel_emb.as<float>(),
                       mel_pos.as<float>(), out_dev, cfg.hidden, cfg.mel_codes, cfg.max_mel_pos, (int)id, (int)gen_len);
    MI_HIP(hipGetLastError());
}

void Gpt::kv_read(int layer, float* keys_dev, float* values_dev) {
    MI_REQUIRE(layer >= 0 && layer < cfg.layers, "gpt: layer index");
    const int hist = history;
    if (hist == 0) return;
    const size_t es = dtype_size(dtype), off = (size_t)layer * cfg.hidden * cfg.max_seq * es;
    const long items = (long)cfg.heads * hist * 64;
    const dim3 grid((unsigned)((items + 255) / 256));
#define KVE(T, ...) hipLaunchKernelGGL(kv_export_kernel<T>, grid, dim3(256), 0, stream, (const T*)((char*)kc.p + off), (const T*)((char*)vc.p + off), keys_dev, values_dev, cfg.heads, hist, cfg.max_seq)
    GPT_DISPATCH(KVE, 0);
#undef KVE
}

void Gpt::kv_write(int layer, const float* keys_dev, const float* values_dev, int hist) {
    MI_REQUIRE(layer >= 0 && layer < cfg.layers, "gpt: layer index");
    MI_REQUIRE(hist >= 0 && hist < cfg.max_seq, "gpt: history does not fit the KV cache (max_seq)");
    if (hist > 0) {
        const size_t es = dtype_size(dtype), off = (size_t)layer * cfg.hidden * cfg.max_seq * es;
        const long items = (long)cfg.heads * hist * 64;
        const dim3 grid((unsigned)((items + 255) / 256));
#define KVI(T, ...) hipLaunchKernelGGL(kv_import_kernel<T>, grid, dim3(256), 0, stream, (T*)((char*)kc.p + off), (T*)((char*)vc.p + off), keys_dev, values_dev, cfg.heads, hist, cfg.max_seq)
        GPT_DISPATCH(KVI, 0);
#undef KVI
    }
    std::vector<int32_t> w = get_state();
    w[GS_HIST] = hist;
    set_state(w);
}

}  // namespace mi
