// cond.hip — IndexTTS graph A on the GPU (fp32): prompt audio -> mel -> Conformer conditioning encoder -> Perceiver resampler
// (conds_latent) and ECAPA-TDNN speaker encoder -> BigVGAN conditioning vectors.
//
// Reference: IndexTTS_A (IndexTTS/Export_IndexTTS.py:74-200; forward :131-200; the weight folds of __init__ :88-129 are applied by
// the packer, mi355tts.weights.fold_cond).  Runs once per prompt (the reference driver calls ort_session_A once,
// Inference_IndexTTS_ONNX.py:700-712), ~25 GFLOP at 6 s of audio: every matrix product with more than a few rows goes through
// the implicit-GEMM launcher of the other engines (launch_conv_gemm: MFMA, channels-last rows); the rest — the rel-pos attention
// with its rel_shift gather, GLU / GEGLU, depthwise conv, Res2Net adds, SE gates, attentive statistics pooling — are small
// one-purpose kernels: this graph is latency, not throughput.  Activations are channels-last [T][C] throughout.
#include "cond.h"
#include "f5_kernels.h"
#include <cmath>
#include <algorithm>

namespace mi {

CondCfg parse_cond_cfg(const int32_t* a, int n) {
    MI_REQUIRE(a && n >= 24, "cond cfg: too short");
    CondCfg c;
    int i = 0;
    c.n_fft = a[i++]; c.hop = a[i++]; c.mel = a[i++]; c.sr = a[i++]; c.audio_pad = a[i++]; c.max_len = a[i++];
    c.d = a[i++]; c.heads = a[i++]; c.lin = a[i++]; c.blocks = a[i++]; c.kern = a[i++];
    c.D = a[i++]; c.latents = a[i++]; c.pdepth = a[i++]; c.pheads = a[i++]; c.pdh = a[i++]; c.pmult = a[i++];
    c.att = a[i++]; c.r2scale = a[i++]; c.se = a[i++]; c.emb = a[i++]; c.voc0 = a[i++];
    const int ns = a[i++];
    MI_REQUIRE(ns >= 3 && ns <= 8 && i + 3 * ns + 1 <= n, "cond cfg: speaker encoder lists");
    for (int k = 0; k < ns; ++k) c.sch.push_back(a[i++]);
    for (int k = 0; k < ns; ++k) c.sk.push_back(a[i++]);
    for (int k = 0; k < ns; ++k) c.sd.push_back(a[i++]);
    const int nv = a[i++];
    MI_REQUIRE(nv >= 1 && nv <= 8 && i + nv <= n, "cond cfg: vocoder channel list");
    for (int k = 0; k < nv; ++k) c.vch.push_back(a[i++]);
    MI_REQUIRE(c.d % c.heads == 0 && c.dk() <= 64 && c.pdh <= 64 && c.d % 4 == 0 && c.D % 4 == 0 && c.mel % 4 == 0 && c.n_fft % 8 == 0 &&
               c.kern % 2 == 1 && c.emb % 4 == 0 && c.att % 4 == 0 && c.se % 4 == 0, "cond cfg: unsupported sizes");
    for (int k = 0; k + 1 < ns; ++k) MI_REQUIRE(c.sch[k] % (4 * c.r2scale) == 0 && c.sk[k] % 2 == 1, "cond cfg: ECAPA channels");
    MI_REQUIRE(c.sch[ns - 1] == (ns - 2) * c.sch[1], "cond cfg: mfa channels = concatenated SE-Res2Net outputs");
    for (int k = 1; k + 1 < ns; ++k) MI_REQUIRE(c.sch[k] == c.sch[0], "cond cfg: SE-Res2Net blocks keep the channel count (no shortcut conv)");
    return c;
}

static int64_t tdnn_params(int cin, int cout, int k) { return (int64_t)cout * cin * k + 5 * (int64_t)cout; }

int64_t cond_param_count(const CondCfg& c) {
    int64_t n = c.audio_pad + (int64_t)c.d * 9 + c.d + (int64_t)c.d * c.d * c.f2() + c.d;
    n += (int64_t)c.blocks * (10LL * c.d + 4LL * ((int64_t)c.d * c.d + c.d) + (int64_t)c.d * c.d + 2LL * c.d + 2LL * c.d * c.d + 2 * c.d +
                              (int64_t)c.d * c.kern + c.d + (int64_t)c.d * c.d + c.d + (int64_t)c.lin * c.d + c.lin + (int64_t)c.d * c.lin + c.d);
    n += 2 * c.d;
    n += (int64_t)c.D * c.d + c.D + (int64_t)c.latents * c.D;
    n += (int64_t)c.pdepth * (4LL * c.inner() * c.D + 2LL * c.ffi() * c.D + 2 * c.ffi() + (int64_t)c.D * c.ffi() + c.D);
    n += c.D;
    const int ns = (int)c.sch.size();
    n += tdnn_params(c.mel, c.sch[0], c.sk[0]);
    for (int i = 1; i + 1 < ns; ++i) {
        const int ch = c.sch[i], cs = ch / c.r2scale;
        n += tdnn_params(c.sch[i - 1], ch, 1) + (int64_t)(c.r2scale - 1) * tdnn_params(cs, cs, c.sk[i]) + tdnn_params(ch, ch, 1);
        n += (int64_t)c.se * ch + c.se + (int64_t)ch * c.se + ch;
    }
    const int cm = c.sch[ns - 1];
    n += tdnn_params(cm, cm, c.sk[ns - 1]) + tdnn_params(3 * cm, c.att, 1) + (int64_t)cm * c.att + cm + 4LL * 2 * cm + (int64_t)c.emb * 2 * cm + c.emb;
    n += (int64_t)c.voc0 * c.emb + c.voc0;
    for (int v : c.vch) n += (int64_t)v * c.emb + v;
    return n;
}

// ---------------------------------------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// [zeros half | audio_pad | audio / 32768 | zeros half]   (:132-134, STFT_Process 'constant' padding)
__global__ void k_pad_audio(const int16_t* a, const float* pad, float* out, long L, int apad, int half) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x, n = L + apad + 2L * half;
    if (i >= n) return;
    float v = 0.f;
    const long j = i - half;
    if (j >= 0 && j < apad) v = pad[j];
    else if (j >= apad && j < apad + L) v = (float)a[j - apad] * (1.0f / 32768.0f);
    out[i] = v;
}
__global__ void k_logclamp(float* x, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = logf(fmaxf(x[i], 1e-5f));
}
// Conv2d(1, d, 3, stride 2) + ReLU on the (T, mel) image, output already in the (T2, d * F2) order of the following Linear (:136-138)
__global__ void k_conv2d_sub(const float* mel, const float* w, const float* b, float* out, int T2, int M, int d, int F2) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x, n = (long)T2 * d * F2;
    if (i >= n) return;
    const int f = (int)(i % F2), c = (int)((i / F2) % d), t = (int)(i / ((long)F2 * d));
    float s = b[c];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int e = 0; e < 3; ++e) s += w[c * 9 + a * 3 + e] * mel[(long)(2 * t + a) * M + 2 * f + e];
    out[i] = fmaxf(s, 0.f);
}
// out[m][n] = act(sum_k x[m][k] w[n][k] + b[n]) (+ res[m][n]) for a handful of rows: one wave per output
__global__ void k_lin_small(const float* x, long ldx, const float* w, const float* b, float* out, long ldo, const float* res, int M, int N, int K, int act) {
    const int lane = threadIdx.x & 63;
    const long o = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= (long)M * N) return;
    const int m = (int)(o / N), n = (int)(o % N);
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += x[m * ldx + k] * w[(long)n * K + k];
    s = wsum(s);
    if (lane == 0) {
        s += b ? b[n] : 0.f;
        if (act == 1) s = fmaxf(s, 0.f);
        else if (act == 2) s = 1.f / (1.f + expf(-s));
        if (res) s += res[m * ldo + n];
        out[m * ldo + n] = s;
    }
}
// bd[h][i][j] = (q[i][h] + vb[h]) . p[j][h]      (:149 before rel_shift)
__global__ void k_scores(const float* q, long ldq, const float* qb, const float* p, long ldp, float* bd, int H, int dk, int T) {
    __shared__ float qs[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long idx = (long)blockIdx.x * 4 + w;
    const bool live = idx < (long)H * T;
    const int h = live ? (int)(idx / T) : 0, i = live ? (int)(idx % T) : 0;
    if (lane < dk) qs[w][lane] = q[i * ldq + h * dk + lane] + qb[h * dk + lane];
    __syncthreads();
    if (!live) return;
    for (int j = lane; j < T; j += 64) {
        float s = 0.f;
        for (int c = 0; c < dk; ++c) s += qs[w][c] * p[j * ldp + h * dk + c];
        bd[((long)h * T + i) * T + j] = s;
    }
}
// one wave per (head, query): softmax_j((q_i + qb) . k_j + shift(bd)[i][j]) . v   (:147-152 ; perceiver :171-172 with qb = bd = null)
// rel_shift (:67-71): shifted[i][j] = flat[(i + 1) * T + j] of the (T, T + 1) block whose column 0 is zero and column b + 1 is bd[.][b]
__global__ void k_rows_attn(const float* q, long ldq, const float* qb, const float* k, long ldk, const float* v, long ldv, const float* bd,
                            float* out, long ldo, int H, int dk, int Tq, int Tk) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* qs = sm + w * 64;
    float* sc = sm + 4 * 64 + (long)w * Tk;
    const long idx = (long)blockIdx.x * 4 + w;
    const bool live = idx < (long)H * Tq;
    const int h = live ? (int)(idx / Tq) : 0, i = live ? (int)(idx % Tq) : 0;
    if (lane < dk) qs[lane] = q[i * ldq + h * dk + lane] + (qb ? qb[h * dk + lane] : 0.f);
    __syncthreads();
    if (!live) return;
    float mx = -INFINITY;
    for (int j = lane; j < Tk; j += 64) {
        float s = 0.f;
        for (int c = 0; c < dk; ++c) s += qs[c] * k[j * ldk + h * dk + c];
        if (bd) {
            const long f = (long)(i + 1) * Tk + j;
            const int a = (int)(f / (Tk + 1)), b = (int)(f % (Tk + 1));
            if (b) s += bd[((long)h * Tq + a) * Tk + b - 1];
        }
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wmax(mx);
    float sum = 0.f;
    for (int j = lane; j < Tk; j += 64) { const float e = expf(sc[j] - mx); sc[j] = e; sum += e; }
    sum = wsum(sum);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float inv = 1.f / sum;
    for (int c = lane; c < dk; c += 64) {
        float acc = 0.f;
        for (int j = 0; j < Tk; ++j) acc += sc[j] * v[j * ldv + h * dk + c];
        out[i * ldo + h * dk + c] = acc * inv;
    }
}
__global__ void k_glu(const float* x, float* y, long T, int d) {            // (T, 2d) -> (T, d): a * sigmoid(b)   (:158)
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= T * d) return;
    const long t = i / d; const int c = (int)(i % d);
    y[i] = x[t * 2 * d + c] * (1.f / (1.f + expf(-x[t * 2 * d + d + c])));
}
__global__ void k_dwconv(const float* x, const float* w, const float* b, float* y, int T, int d, int k) {     // zero-padded depthwise, (T, d)
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)T * d) return;
    const int t = (int)(i / d), c = (int)(i % d), half = (k - 1) / 2;
    float s = b[c];
    for (int j = 0; j < k; ++j) { const int tt = t + j - half; if (tt >= 0 && tt < T) s += w[c * k + j] * x[(long)tt * d + c]; }
    y[i] = s;
}
__global__ void k_silu(float* x, long n) { const long i = (long)blockIdx.x * 256 + threadIdx.x; if (i < n) x[i] = x[i] / (1.f + expf(-x[i])); }
__global__ void k_tanh(float* x, long n) { const long i = (long)blockIdx.x * 256 + threadIdx.x; if (i < n) x[i] = tanhf(x[i]); }
__global__ void k_geglu(const float* x, float* y, int L, int ffi, int ldy) {     // (L, 2 ffi) -> (L, ldy): gelu_erf(gate) * x, zero tail
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)L * ldy) return;
    const int r = (int)(i / ldy), c = (int)(i % ldy);
    float o = 0.f;
    if (c < ffi) { const float g = x[(long)r * 2 * ffi + ffi + c]; o = 0.5f * g * (1.f + erff(g * 0.7071067811865476f)) * x[(long)r * 2 * ffi + c]; }
    y[i] = o;
}
__global__ void k_rmsnorm(const float* x, const float* g, float* y, int L, int D, float scale) {     // F.normalize(x) * sqrt(D) * gamma
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= L) return;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) { const float v = x[(long)r * D + c]; s += v * v; }
    const float inv = scale / fmaxf(sqrtf(wsum(s)), 1e-12f);
    for (int c = lane; c < D; c += 64) y[(long)r * D + c] = x[(long)r * D + c] * inv * g[c];
}
// dst (T + 2p, cs) = reflect-padded (a [+ b]) channel slice
__global__ void k_reflect_pad(const float* a, long lda, const float* b, long ldb, float* dst, int T, int cs, int p) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)(T + 2 * p) * cs) return;
    const int t = (int)(i / cs), c = (int)(i % cs);
    int s = t - p;
    if (s < 0) s = -s; else if (s >= T) s = 2 * (T - 1) - s;
    dst[i] = a[(long)s * lda + c] + (b ? b[(long)s * ldb + c] : 0.f);
}
__global__ void k_relu_affine(float* x, long ld, const float* sc, const float* sh, int T, int C) {      // BatchNorm(ReLU(x)), BN folded
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)T * C) return;
    const int t = (int)(i / C), c = (int)(i % C);
    x[(long)t * ld + c] = fmaxf(x[(long)t * ld + c], 0.f) * sc[c] + sh[c];
}
// per channel over time: mean = sum_t w x, std = sqrt(clamp(sum_t w (x - mean)^2, 1e-6)); w = weights (T, C) or uniform 1 / T   (:60-63)
__global__ void k_colstats(const float* x, long ld, const float* wgt, float* mean, float* stdv, int T, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float u = 1.0f / (float)T;
    float m = 0.f;
    for (int t = 0; t < T; ++t) m += (wgt ? wgt[(long)t * C + c] : u) * x[(long)t * ld + c];
    float s = 0.f;
    for (int t = 0; t < T; ++t) { const float d = x[(long)t * ld + c] - m; s += (wgt ? wgt[(long)t * C + c] : u) * d * d; }
    mean[c] = m; stdv[c] = sqrtf(fmaxf(s, 1e-6f));
}
__global__ void k_colmean(const float* x, long ld, float* mean, int T, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float m = 0.f;
    for (int t = 0; t < T; ++t) m += x[(long)t * ld + c];
    mean[c] = m / (float)T;
}
__global__ void k_se_apply(const float* y, const float* gate, const float* res, long ldr, float* out, long ldo, int T, int C) {   // gate * y + res
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)T * C) return;
    const int t = (int)(i / C), c = (int)(i % C);
    out[(long)t * ldo + c] = gate[c] * y[i] + res[(long)t * ldr + c];
}
__global__ void k_asp_in(const float* x, const float* mean, const float* stdv, float* out, int T, int C) {     // [x | mean | std] rows (:187-189)
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)T * 3 * C) return;
    const int t = (int)(i / (3 * C)), c = (int)(i % (3 * C));
    out[i] = c < C ? x[(long)t * C + c] : c < 2 * C ? mean[c - C] : stdv[c - 2 * C];
}
__global__ void k_colsoftmax(float* x, int T, int C) {       // softmax over time per channel, in place (:191)
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float mx = -INFINITY;
    for (int t = 0; t < T; ++t) mx = fmaxf(mx, x[(long)t * C + c]);
    float s = 0.f;
    for (int t = 0; t < T; ++t) { const float e = expf(x[(long)t * C + c] - mx); x[(long)t * C + c] = e; s += e; }
    const float inv = 1.f / s;
    for (int t = 0; t < T; ++t) x[(long)t * C + c] *= inv;
}
__global__ void k_affine(float* x, const float* sc, const float* sh, int n) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < n) x[i] = x[i] * sc[i] + sh[i]; }
__global__ void k_copy_rows(const float* src, float* dst, long n) { const long i = (long)blockIdx.x * 256 + threadIdx.x; if (i < n) dst[i] = src[i]; }

static inline dim3 g1(long n) { return dim3((unsigned)((n + 255) / 256)); }

// ---------------------------------------------------------------------------------------------------------------------
// engine
// ---------------------------------------------------------------------------------------------------------------------
struct WB { DevBuf w, b; int n = 0, k = 0; };                  // linear / conv weight [n][k (tap-major)] + bias
struct BN { DevBuf sc, sh; };                                  // eval BatchNorm folded: y = x * sc + sh
struct TD { WB c; BN bn; int k = 1, d = 1; };                  // TDNNBlock

struct Cond {
    CondCfg cfg; int device; hipStream_t s = nullptr;
    DevBuf audio_pad, stft_w, fbank, pe, sub_w, sub_b;
    WB sub_out;
    struct Layer {
        DevBuf nm_w, nm_b, nc_w, nc_b, nf_w, nf_b, nl_w, nl_b, cn_w, cn_b, bu, bv, dw_w, dw_b;
        WB qkv, pos, out, pw1, pw2, w1, w2;
    };
    std::vector<Layer> layers;
    DevBuf an_w, an_b;
    WB proj; DevBuf lat0, gamma;
    struct PL { WB q, kv, out, f0, f2; };
    std::vector<PL> pl;
    TD b0; struct SR { TD t1, t2; std::vector<TD> r; WB se1, se2; };
    std::vector<SR> sr;
    TD mfa, asp_t; WB asp_c; BN asp_bn; WB fc, conds;
    // workspace
    DevBuf a_dev, padded, spec, mag, mel, sub, x, x1, qkv, pp, bd, ao, t2d, td, hid, ctxcat, lat, pq, pkv, pao, pf, pg, clat;
    DevBuf e0, e1, e2, e3, cat, chunkpad, ypad, gate1, gate2, mean, stdv, aspin, att, attw, embv, condv;
    long cap_T = 0;
};
int cond_device(const Cond* e) { return e->device; }

static void up(DevBuf& d, const float* p, size_t n, hipStream_t s) { upload_f32(d, p, n, s); }

// torch Conv1d weight (co, ci, k) -> [co][tap][ci]
static std::vector<float> relay(const float* w, int co, int ci, int k) {
    std::vector<float> o((size_t)co * ci * k);
    for (int a = 0; a < co; ++a) for (int b = 0; b < ci; ++b) for (int t = 0; t < k; ++t) o[((size_t)a * k + t) * ci + b] = w[((size_t)a * ci + b) * k + t];
    return o;
}

void cond_destroy(Cond* e);

Cond* cond_create(const CondCfg& c, const float* w, int64_t nw, int device) {
    MI_REQUIRE(w && nw == cond_param_count(c), "mi_indextts_cond_create: weight count does not match the config");
    MI_HIP(hipSetDevice(device));
    // owned until creation has succeeded: an upload or a size check that throws half way must not leak the object and its stream
    struct Guard { Cond* e; ~Guard() { if (e) cond_destroy(e); } } guard{new Cond};
    Cond* e = guard.e; e->cfg = c; e->device = device;
    MI_HIP(hipStreamCreateWithFlags(&e->s, hipStreamNonBlocking));
    hipStream_t s = e->s;
    const float* p = w;
    auto take = [&](size_t n) { const float* r = p; p += n; return r; };
    auto lin = [&](WB& L, int n, int k, bool bias) { L.n = n; L.k = k; up(L.w, take((size_t)n * k), (size_t)n * k, s); if (bias) up(L.b, take(n), n, s); };
    auto vec = [&](DevBuf& d, int n) { up(d, take(n), n, s); };
    auto conv = [&](WB& L, int co, int ci, int k, bool bias = true) {
        const float* ww = take((size_t)co * ci * k);
        auto r = relay(ww, co, ci, k);
        L.n = co; L.k = ci * k; up(L.w, r.data(), r.size(), s);
        if (bias) up(L.b, take(co), co, s);
    };
    auto bnfold = [&](BN& b, int n) {
        const float* g = take(n); const float* be = take(n); const float* rm = take(n); const float* rv = take(n);
        std::vector<float> sc(n), sh(n);
        for (int i = 0; i < n; ++i) { sc[i] = g[i] / std::sqrt(rv[i] + c.bn_eps); sh[i] = be[i] - rm[i] * sc[i]; }
        up(b.sc, sc.data(), n, s); up(b.sh, sh.data(), n, s);
    };
    auto tdnn = [&](TD& t, int ci, int co, int k, int d) { t.k = k; t.d = d; conv(t.c, co, ci, k); bnfold(t.bn, co); };

    vec(e->audio_pad, c.audio_pad);
    const int d = c.d, F2 = c.f2();
    up(e->sub_w, take((size_t)d * 9), (size_t)d * 9, s); vec(e->sub_b, d);
    lin(e->sub_out, d, d * F2, true);
    e->layers.resize(c.blocks);
    for (auto& L : e->layers) {
        // spec order: five norms (mha, conv, ff, final, conv_module.norm), q k v out (w, b), pos, bias u, bias v, pw1, dw, pw2, w_1, w_2
        vec(L.nm_w, d); vec(L.nm_b, d); vec(L.nc_w, d); vec(L.nc_b, d); vec(L.nf_w, d); vec(L.nf_b, d); vec(L.nl_w, d); vec(L.nl_b, d);
        vec(L.cn_w, d); vec(L.cn_b, d);
        L.qkv.n = 3 * d; L.qkv.k = d; L.qkv.w.ensure((size_t)3 * d * d * 4); L.qkv.b.ensure((size_t)3 * d * 4);
        for (int t3 = 0; t3 < 3; ++t3) {
            MI_HIP(hipMemcpyAsync((float*)L.qkv.w.p + (size_t)t3 * d * d, take((size_t)d * d), (size_t)d * d * 4, hipMemcpyHostToDevice, s));
            MI_HIP(hipMemcpyAsync((float*)L.qkv.b.p + (size_t)t3 * d, take(d), (size_t)d * 4, hipMemcpyHostToDevice, s));
        }
        lin(L.out, d, d, true);
        lin(L.pos, d, d, false);
        vec(L.bu, d); vec(L.bv, d);
        conv(L.pw1, 2 * d, d, 1);
        up(L.dw_w, take((size_t)d * c.kern), (size_t)d * c.kern, s); vec(L.dw_b, d);
        conv(L.pw2, d, d, 1);
        lin(L.w1, c.lin, d, true); lin(L.w2, d, c.lin, true);
    }
    vec(e->an_w, d); vec(e->an_b, d);
    lin(e->proj, c.D, d, true);
    up(e->lat0, take((size_t)c.latents * c.D), (size_t)c.latents * c.D, s);
    e->pl.resize(c.pdepth);
    const int ffi = c.ffi(), ldf = (ffi + 3) / 4 * 4, inner = c.inner();
    for (auto& P : e->pl) {
        lin(P.q, inner, c.D, false); lin(P.kv, 2 * inner, c.D, false); lin(P.out, c.D, inner, false);
        lin(P.f0, 2 * ffi, c.D, true);
        {   // second FF linear: K = ffi padded to a multiple of 4 with zero columns (the GEGLU kernel writes zeros there)
            const float* ww = take((size_t)c.D * ffi);
            std::vector<float> r((size_t)c.D * ldf, 0.f);
            for (int a = 0; a < c.D; ++a) std::copy(ww + (size_t)a * ffi, ww + (size_t)(a + 1) * ffi, r.begin() + (size_t)a * ldf);
            P.f2.n = c.D; P.f2.k = ldf; up(P.f2.w, r.data(), r.size(), s);
            up(P.f2.b, take(c.D), c.D, s);
        }
    }
    vec(e->gamma, c.D);
    const int ns = (int)c.sch.size();
    tdnn(e->b0, c.mel, c.sch[0], c.sk[0], c.sd[0]);
    e->sr.resize(ns - 2);
    for (int i = 1; i + 1 < ns; ++i) {
        auto& R = e->sr[i - 1];
        const int ch = c.sch[i], cs = ch / c.r2scale;
        tdnn(R.t1, c.sch[i - 1], ch, 1, 1);
        R.r.resize(c.r2scale - 1);
        for (auto& t : R.r) tdnn(t, cs, cs, c.sk[i], c.sd[i]);
        tdnn(R.t2, ch, ch, 1, 1);
        conv(R.se1, c.se, ch, 1); conv(R.se2, ch, c.se, 1);
    }
    const int cm = c.sch[ns - 1];
    tdnn(e->mfa, cm, cm, c.sk[ns - 1], c.sd[ns - 1]);
    tdnn(e->asp_t, 3 * cm, c.att, 1, 1);
    conv(e->asp_c, cm, c.att, 1);
    bnfold(e->asp_bn, 2 * cm);
    conv(e->fc, c.emb, 2 * cm, 1);
    {   // cond_layer and conds[i] stacked into one [ncond][emb] matrix: one product gives cond_layer | conds_0 | ... (:196-199)
        const int nc = c.ncond();
        std::vector<float> W((size_t)nc * c.emb), B(nc);
        size_t ro = 0;
        auto one = [&](int co) {
            const float* ww = take((size_t)co * c.emb); const float* bb = take(co);
            std::copy(ww, ww + (size_t)co * c.emb, W.begin() + ro * c.emb); std::copy(bb, bb + co, B.begin() + ro); ro += co;
        };
        one(c.voc0);
        for (int v : c.vch) one(v);
        e->conds.n = nc; e->conds.k = c.emb; up(e->conds.w, W.data(), W.size(), s); up(e->conds.b, B.data(), nc, s);
    }
    MI_REQUIRE(p - w == nw, "cond: weight walk mismatch");
    // ---- tables: STFT kernels (STFT_Process.py:86-98, torch's fp32 evaluation order), HTK mel bank, rel-pos table through fp16 (:87, :139)
    {
        const int nf = c.n_fft, nb = nf / 2 + 1, ldm = (nb + 7) / 8 * 8;
        std::vector<float> win(nf), sw((size_t)2 * nb * nf);
        const float wstep = (float)(2.0 * M_PI / nf), two_pi = (float)(2.0 * M_PI);
        for (int n = 0; n < nf; ++n) win[n] = cosf((float)n * wstep) * -0.5f + 0.5f;
        for (int f = 0; f < nb; ++f)
            for (int t = 0; t < nf; ++t) {
                const float om = ((two_pi * (float)f) * (float)t) / (float)nf;
                sw[(size_t)f * nf + t] = cosf(om) * win[t];
                sw[(size_t)(nb + f) * nf + t] = -sinf(om) * win[t];
            }
        up(e->stft_w, sw.data(), sw.size(), s);
        std::vector<float> fb((size_t)c.mel * ldm, 0.f);
        const double m_max = 2595.0 * std::log10(1.0 + (c.sr / 2) / 700.0);
        std::vector<double> fp(c.mel + 2);
        for (int i = 0; i < c.mel + 2; ++i) fp[i] = 700.0 * (std::pow(10.0, (m_max * i / (c.mel + 1)) / 2595.0) - 1.0);
        for (int k = 0; k < nb; ++k) {
            const double fr = (double)(c.sr / 2) * k / (nb - 1);
            for (int m = 0; m < c.mel; ++m)
                fb[(size_t)m * ldm + k] = (float)std::max(0.0, std::min((fr - fp[m]) / (fp[m + 1] - fp[m]), (fp[m + 2] - fr) / (fp[m + 2] - fp[m + 1])));
        }
        up(e->fbank, fb.data(), fb.size(), s);
        std::vector<float> pe((size_t)c.max_len * d);
        for (int q = 0; q < c.max_len; ++q)
            for (int i = 0; i < d; i += 2) {
                const float div = expf((float)i * (float)(-(std::log(10000.0) / d)));
                pe[(size_t)q * d + i] = (float)(f16)sinf((float)q * div);
                pe[(size_t)q * d + i + 1] = (float)(f16)cosf((float)q * div);
            }
        up(e->pe, pe.data(), pe.size(), s);
    }
    MI_HIP(hipStreamSynchronize(s));
    guard.e = nullptr;
    return e;
}

void cond_destroy(Cond* e) {
    if (!e) return;
    if (e->s) (void)hipStreamDestroy(e->s);
    delete e;
}

// out (M, N) = act(x (M, K taps over rows) . w + b) (+ res): the MFMA launcher for real row counts, one wave per output for a few rows
static void gemm(Cond* e, const float* x, long ldx, int Cin, int taps, int dil, const WB& L, float* out, long ldo, int M, int act = ACT_NONE,
                 const float* res = nullptr) {
    MI_REQUIRE(L.k == Cin * taps, "cond: weight shape");
    if (M < 8 || ((uintptr_t)x % 16) || (ldx % 4)) {
        MI_REQUIRE(taps == 1, "cond: small path is one-tap");
        const int a = act == ACT_NONE ? 0 : -1;
        MI_REQUIRE(a == 0, "cond: small path has no activation here");
        hipLaunchKernelGGL(k_lin_small, dim3((unsigned)(((long)M * L.n + 3) / 4)), dim3(256), 0, e->s, x, ldx, L.w.as<float>(), L.b.p ? L.b.as<float>() : nullptr,
                           out, ldo, res, M, L.n, L.k, 0);
        return;
    }
    ConvGemm g;
    g.dtype = MI_F32; g.x = x; g.w = L.w.p; g.bias = L.b.p ? L.b.as<float>() : nullptr; g.out = out; g.res = res;
    g.B = 1; g.T_in = M + (taps - 1) * dil; g.M = M; g.N = L.n; g.Cin = Cin; g.taps = taps; g.dil = dil; g.pad = 0;
    g.x_bstride = 0; g.x_rstride = ldx; g.out_bstride = 0; g.out_rstride = ldo; g.act = act;
    launch_conv_gemm(g, e->s);
}
static void small(Cond* e, const float* x, const WB& L, float* out, int act) {       // one row (SE gates, embedding, cond vectors)
    hipLaunchKernelGGL(k_lin_small, dim3((unsigned)((L.n + 3) / 4)), dim3(256), 0, e->s, x, (long)L.k, L.w.as<float>(), L.b.p ? L.b.as<float>() : nullptr, out,
                       (long)L.n, (const float*)nullptr, 1, L.n, L.k, act);
}
static void ln(Cond* e, const float* x, float* y, const DevBuf& w, const DevBuf& b, long rows, int D) {
    launch_rownorm(NORM_LN_AFFINE, x, y, MI_F32, w.as<float>(), b.as<float>(), rows, D, e->cfg.ln_eps, e->s);
}
// TDNNBlock on a channel slice: reflect-pad (a [+ b]) -> conv(k, dil) -> ReLU -> BN, into out (T, ldo)
static void tdnn_run(Cond* e, const TD& t, const float* a, long lda, const float* b, long ldb, int cin, float* out, long ldo, int T) {
    const int p = t.d * (t.k - 1) / 2;
    const float* src = a; long lds_ = lda;
    if (p > 0 || b) {
        e->chunkpad.ensure((size_t)(T + 2 * p) * cin * 4);
        hipLaunchKernelGGL(k_reflect_pad, g1((long)(T + 2 * p) * cin), dim3(256), 0, e->s, a, lda, b, ldb, e->chunkpad.as<float>(), T, cin, p);
        src = e->chunkpad.as<float>(); lds_ = cin;
    }
    gemm(e, src, lds_, cin, t.k, t.d, t.c, out, ldo, T);
    hipLaunchKernelGGL(k_relu_affine, g1((long)T * t.c.n), dim3(256), 0, e->s, out, ldo, t.bn.sc.as<float>(), t.bn.sh.as<float>(), T, t.c.n);
}

void cond_run(Cond* e, const int16_t* audio, long L, float* conds_out, float* latent_out, float* mel_out, int mem) {
    const CondCfg& c = e->cfg;
    MI_HIP(hipSetDevice(e->device));
    hipStream_t s = e->s;
    const int nf = c.n_fft, nb = nf / 2 + 1, ldm = (nb + 7) / 8 * 8, half = nf / 2, d = c.d, H = c.heads, dk = c.dk();
    const long T = c.frames(L);
    const int T2 = (int)((T - 3) / 2 + 1), F2 = c.f2();
    int maxpad = 1;
    for (size_t i = 0; i < c.sk.size(); ++i) maxpad = std::max(maxpad, c.sd[i] * (c.sk[i] - 1) / 2);
    MI_REQUIRE(audio && L >= 1 && T > maxpad + 1 && T2 >= 2 && T2 <= c.max_len && T < (1 << 20), "mi_indextts_cond_run: prompt too short or too long");
    // ---- mel front end (:132-135) ---------------------------------------------------------------------------------------
    const int16_t* da = audio;
    if (mem == MI_HOST) { e->a_dev.ensure((size_t)L * 2); MI_HIP(hipMemcpyAsync(e->a_dev.p, audio, (size_t)L * 2, hipMemcpyHostToDevice, s)); da = e->a_dev.as<int16_t>(); }
    const long Lp = L + c.audio_pad + nf;
    e->padded.ensure((size_t)Lp * 4 + 64); e->spec.ensure((size_t)T * 2 * nb * 4); e->mag.ensure((size_t)T * ldm * 4); e->mel.ensure((size_t)T * c.mel * 4);
    hipLaunchKernelGGL(k_pad_audio, g1(Lp), dim3(256), 0, s, da, e->audio_pad.as<float>(), e->padded.as<float>(), L, c.audio_pad, half);
    {
        ConvGemm g;      // framed GEMM: row f = padded[f * hop : f * hop + n_fft]
        g.dtype = MI_F32; g.x = e->padded.p; g.w = e->stft_w.p; g.out = e->spec.p;
        g.B = 1; g.T_in = (int)T; g.M = (int)T; g.N = 2 * nb; g.Cin = nf; g.x_rstride = c.hop; g.x_bstride = 0; g.out_rstride = 2 * nb; g.out_bstride = 0;
        launch_conv_gemm(g, s);
        launch_spec_mag(e->spec.as<float>(), e->mag.as<float>(), (int)T, nb, ldm, 0.f, s);
        ConvGemm m;
        m.dtype = MI_F32; m.x = e->mag.p; m.w = e->fbank.p; m.out = e->mel.p;
        m.B = 1; m.T_in = (int)T; m.M = (int)T; m.N = c.mel; m.Cin = ldm; m.x_rstride = ldm; m.out_rstride = c.mel;
        launch_conv_gemm(m, s);
        hipLaunchKernelGGL(k_logclamp, g1(T * c.mel), dim3(256), 0, s, e->mel.as<float>(), T * c.mel);
    }
    const float* mel = e->mel.as<float>();        // (T, mel) channels-last
    // ---- Conformer conditioning encoder (:136-165) ------------------------------------------------------------------------
    e->sub.ensure((size_t)T2 * d * F2 * 4);
    for (DevBuf* b : {&e->x, &e->x1, &e->ao, &e->td, &e->pp}) b->ensure((size_t)T2 * d * 4);
    e->qkv.ensure((size_t)T2 * 3 * d * 4); e->bd.ensure((size_t)H * T2 * T2 * 4); e->t2d.ensure((size_t)T2 * 2 * d * 4); e->hid.ensure((size_t)T2 * c.lin * 4);
    float* x = e->x.as<float>(); float* x1 = e->x1.as<float>();
    hipLaunchKernelGGL(k_conv2d_sub, g1((long)T2 * d * F2), dim3(256), 0, s, mel, e->sub_w.as<float>(), e->sub_b.as<float>(), e->sub.as<float>(), T2, c.mel, d, F2);
    gemm(e, e->sub.as<float>(), (long)d * F2, d * F2, 1, 1, e->sub_out, x, d, T2);
    const size_t attn_lds = (size_t)(4 * 64 + 4 * (size_t)std::max(T2, c.latents + T2)) * 4;
    MI_REQUIRE(attn_lds <= 64 * 1024, "mi_indextts_cond_run: prompt too long for the attention row buffer");
    for (auto& Ly : e->layers) {
        ln(e, x, x1, Ly.nm_w, Ly.nm_b, T2, d);
        gemm(e, x1, d, d, 1, 1, Ly.qkv, e->qkv.as<float>(), 3 * d, T2);
        gemm(e, e->pe.as<float>(), d, d, 1, 1, Ly.pos, e->pp.as<float>(), d, T2);
        const float* q = e->qkv.as<float>();
        hipLaunchKernelGGL(k_scores, dim3((unsigned)(((long)H * T2 + 3) / 4)), dim3(256), 0, s, q, (long)3 * d, Ly.bv.as<float>(), e->pp.as<float>(), (long)d, e->bd.as<float>(), H, dk, T2);
        hipLaunchKernelGGL(k_rows_attn, dim3((unsigned)(((long)H * T2 + 3) / 4)), dim3(256), attn_lds, s, q, (long)3 * d, Ly.bu.as<float>(), q + d, (long)3 * d, q + 2 * d,
                           (long)3 * d, e->bd.as<float>(), e->ao.as<float>(), (long)d, H, dk, T2, T2);
        gemm(e, e->ao.as<float>(), d, d, 1, 1, Ly.out, x, d, T2, ACT_NONE, x);                                   // x += attn_out (:153)
        ln(e, x, x1, Ly.nc_w, Ly.nc_b, T2, d);
        gemm(e, x1, d, d, 1, 1, Ly.pw1, e->t2d.as<float>(), 2 * d, T2);
        hipLaunchKernelGGL(k_glu, g1((long)T2 * d), dim3(256), 0, s, e->t2d.as<float>(), x1, (long)T2, d);
        hipLaunchKernelGGL(k_dwconv, g1((long)T2 * d), dim3(256), 0, s, x1, Ly.dw_w.as<float>(), Ly.dw_b.as<float>(), e->td.as<float>(), T2, d, c.kern);
        ln(e, e->td.as<float>(), x1, Ly.cn_w, Ly.cn_b, T2, d);
        hipLaunchKernelGGL(k_silu, g1((long)T2 * d), dim3(256), 0, s, x1, (long)T2 * d);
        gemm(e, x1, d, d, 1, 1, Ly.pw2, x, d, T2, ACT_NONE, x);                                                   // x += conv module (:161-162)
        ln(e, x, x1, Ly.nf_w, Ly.nf_b, T2, d);
        gemm(e, x1, d, d, 1, 1, Ly.w1, e->hid.as<float>(), c.lin, T2, ACT_SILU);
        gemm(e, e->hid.as<float>(), c.lin, c.lin, 1, 1, Ly.w2, x, d, T2, ACT_NONE, x);                             // x += ff (:163)
        ln(e, x, x, Ly.nl_w, Ly.nl_b, T2, d);
    }
    ln(e, x, x1, e->an_w, e->an_b, T2, d);
    // ---- Perceiver resampler (:166-176) --------------------------------------------------------------------------------------
    const int Ln = c.latents, D = c.D, inner = c.inner(), ffi = c.ffi(), ldf = (ffi + 3) / 4 * 4, PH = c.pheads, pdh = c.pdh, Tk = Ln + T2;
    e->ctxcat.ensure((size_t)Tk * D * 4); e->pq.ensure((size_t)Ln * inner * 4); e->pkv.ensure((size_t)Tk * 2 * inner * 4); e->pao.ensure((size_t)Ln * inner * 4);
    e->pf.ensure((size_t)Ln * 2 * ffi * 4); e->pg.ensure((size_t)Ln * ldf * 4); e->clat.ensure((size_t)Ln * D * 4);
    float* cat = e->ctxcat.as<float>();           // rows [0, Ln): the latents ; rows [Ln, Ln + T2): the projected context (:169)
    gemm(e, x1, d, d, 1, 1, e->proj, cat + (size_t)Ln * D, D, T2);
    MI_HIP(hipMemcpyAsync(cat, e->lat0.p, (size_t)Ln * D * 4, hipMemcpyDeviceToDevice, s));
    for (auto& P : e->pl) {
        gemm(e, cat, D, D, 1, 1, P.q, e->pq.as<float>(), inner, Ln);
        gemm(e, cat, D, D, 1, 1, P.kv, e->pkv.as<float>(), 2 * inner, Tk);
        hipLaunchKernelGGL(k_rows_attn, dim3((unsigned)(((long)PH * Ln + 3) / 4)), dim3(256), attn_lds, s, e->pq.as<float>(), (long)inner, (const float*)nullptr,
                           e->pkv.as<float>(), (long)2 * inner, e->pkv.as<float>() + inner, (long)2 * inner, (const float*)nullptr, e->pao.as<float>(), (long)inner, PH, pdh, Ln, Tk);
        gemm(e, e->pao.as<float>(), inner, inner, 1, 1, P.out, cat, D, Ln, ACT_NONE, cat);                          // latents += attn (:174)
        gemm(e, cat, D, D, 1, 1, P.f0, e->pf.as<float>(), 2 * ffi, Ln);
        hipLaunchKernelGGL(k_geglu, g1((long)Ln * ldf), dim3(256), 0, s, e->pf.as<float>(), e->pg.as<float>(), Ln, ffi, ldf);
        gemm(e, e->pg.as<float>(), ldf, ldf, 1, 1, P.f2, cat, D, Ln, ACT_NONE, cat);                                // latents += ff (:175)
    }
    hipLaunchKernelGGL(k_rmsnorm, dim3((unsigned)((Ln + 3) / 4)), dim3(256), 0, s, cat, e->gamma.as<float>(), e->clat.as<float>(), Ln, D, sqrtf((float)D));
    // ---- ECAPA-TDNN speaker encoder + conditioning vectors (:178-199) ----------------------------------------------------------
    const int ns = (int)c.sch.size(), cm = c.sch[ns - 1], ch = c.sch[1], Ti = (int)T;
    e->e0.ensure((size_t)T * c.sch[0] * 4); e->e1.ensure((size_t)T * ch * 4); e->e2.ensure((size_t)T * ch * 4); e->e3.ensure((size_t)T * ch * 4);
    e->cat.ensure((size_t)T * cm * 4); e->ypad.ensure((size_t)T * cm * 4);
    e->gate1.ensure((size_t)std::max(c.se, 4) * 4); e->gate2.ensure((size_t)cm * 4); e->mean.ensure((size_t)2 * cm * 4); e->stdv.ensure((size_t)cm * 4);
    e->aspin.ensure((size_t)T * 3 * cm * 4); e->att.ensure((size_t)T * c.att * 4); e->attw.ensure((size_t)T * cm * 4); e->embv.ensure((size_t)c.emb * 4);
    e->condv.ensure((size_t)c.ncond() * 4);
    tdnn_run(e, e->b0, mel, c.mel, nullptr, 0, c.mel, e->e0.as<float>(), c.sch[0], Ti);
    const float* xin = e->e0.as<float>(); long ldin = c.sch[0];
    for (int i = 1; i + 1 < ns; ++i) {
        auto& R = e->sr[i - 1];
        const int cs = ch / c.r2scale;
        float* y1 = e->e1.as<float>(); float* y2 = e->e2.as<float>(); float* y3 = e->e3.as<float>();
        tdnn_run(e, R.t1, xin, ldin, nullptr, 0, c.sch[i - 1], y1, ch, Ti);
        // Res2Net: chunk 0 passes through, chunk 1 = block_0(x_1), chunk i = block_{i-1}(x_i + y_{i-1})
        // chunk 0: y2[:, 0:cs] = y1[:, 0:cs]
        MI_HIP(hipMemcpy2DAsync(y2, (size_t)ch * 4, y1, (size_t)ch * 4, (size_t)cs * 4, (size_t)Ti, hipMemcpyDeviceToDevice, s));
        for (int r = 1; r < c.r2scale; ++r)
            tdnn_run(e, R.r[r - 1], y1 + (size_t)r * cs, ch, r == 1 ? nullptr : y2 + (size_t)(r - 1) * cs, ch, cs, y2 + (size_t)r * cs, ch, Ti);
        tdnn_run(e, R.t2, y2, ch, nullptr, 0, ch, y3, ch, Ti);
        // SE gate: sigmoid(conv2(relu(conv1(mean_t y3)))) ; out = gate * y3 + block input, written into its third of the concatenation
        hipLaunchKernelGGL(k_colmean, dim3((unsigned)((ch + 255) / 256)), dim3(256), 0, s, y3, (long)ch, e->mean.as<float>(), Ti, ch);
        small(e, e->mean.as<float>(), R.se1, e->gate1.as<float>(), 1);
        small(e, e->gate1.as<float>(), R.se2, e->gate2.as<float>(), 2);
        float* dst = e->cat.as<float>() + (size_t)(i - 1) * ch;
        hipLaunchKernelGGL(k_se_apply, g1((long)Ti * ch), dim3(256), 0, s, y3, e->gate2.as<float>(), xin, ldin, dst, (long)cm, Ti, ch);
        xin = dst; ldin = cm;
    }
    float* xm = e->ypad.as<float>();              // mfa output (T, cm)
    tdnn_run(e, e->mfa, e->cat.as<float>(), cm, nullptr, 0, cm, xm, cm, Ti);
    hipLaunchKernelGGL(k_colstats, dim3((unsigned)((cm + 255) / 256)), dim3(256), 0, s, xm, (long)cm, (const float*)nullptr, e->mean.as<float>(), e->stdv.as<float>(), Ti, cm);
    hipLaunchKernelGGL(k_asp_in, g1((long)Ti * 3 * cm), dim3(256), 0, s, xm, e->mean.as<float>(), e->stdv.as<float>(), e->aspin.as<float>(), Ti, cm);
    tdnn_run(e, e->asp_t, e->aspin.as<float>(), 3 * cm, nullptr, 0, 3 * cm, e->att.as<float>(), c.att, Ti);
    hipLaunchKernelGGL(k_tanh, g1((long)Ti * c.att), dim3(256), 0, s, e->att.as<float>(), (long)Ti * c.att);
    gemm(e, e->att.as<float>(), c.att, c.att, 1, 1, e->asp_c, e->attw.as<float>(), cm, Ti);
    hipLaunchKernelGGL(k_colsoftmax, dim3((unsigned)((cm + 255) / 256)), dim3(256), 0, s, e->attw.as<float>(), Ti, cm);
    hipLaunchKernelGGL(k_colstats, dim3((unsigned)((cm + 255) / 256)), dim3(256), 0, s, xm, (long)cm, e->attw.as<float>(), e->mean.as<float>(), e->mean.as<float>() + cm, Ti, cm);
    hipLaunchKernelGGL(k_affine, dim3((unsigned)((2 * cm + 255) / 256)), dim3(256), 0, s, e->mean.as<float>(), e->asp_bn.sc.as<float>(), e->asp_bn.sh.as<float>(), 2 * cm);
    small(e, e->mean.as<float>(), e->fc, e->embv.as<float>(), 0);
    small(e, e->embv.as<float>(), e->conds, e->condv.as<float>(), 0);
    // ---- outputs ---------------------------------------------------------------------------------------------------------------
    const hipMemcpyKind kd = mem == MI_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (conds_out) MI_HIP(hipMemcpyAsync(conds_out, e->condv.p, (size_t)c.ncond() * 4, kd, s));
    if (latent_out) MI_HIP(hipMemcpyAsync(latent_out, e->clat.p, (size_t)Ln * D * 4, kd, s));
    if (mel_out) MI_HIP(hipMemcpyAsync(mel_out, e->mel.p, (size_t)T * c.mel * 4, kd, s));
    MI_HIP(hipStreamSynchronize(s));
    MI_HIP(hipGetLastError());
}

}  // namespace mi
