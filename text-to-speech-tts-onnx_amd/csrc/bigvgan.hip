// bigvgan.hip — BigVGAN-v2 generator on gfx950: weight re-layout + the forward launch sequence.
//
// Reference: BigVGAN.forward (BigVGAN/modeling_modified/bigvgan.py:384-410), AMPBlock1.forward
// (:132-140), BIGVGAN.forward int16 tail (BigVGAN/Export_BigVGAN.py:44-49).
//
// HBM plan (channels-last activations, dtype = engine dtype):
//   IN/XS ping-pong : stage input / stage output (mean of the 3 AMP blocks)
//   X               : ConvTranspose1d output = input of all three AMP blocks of the stage
//   T1, T2          : AA-activation / conv1 temporaries
//   P, Q            : running residual stream of the current AMP block
// all sized for the largest stage (B * 1536*F*2 elements) and reused; nothing is re-allocated
// between calls with the same (B, frames).
#include "common.h"
#include "bigvgan.h"
#include "f5_kernels.h"
#include <atomic>
#include <cstdlib>

namespace mi {

static int round_up(int a, int b) { return (a + b - 1) / b * b; }

BigVGANCfg parse_bigvgan_cfg(const int32_t* c, int n) {
    MI_REQUIRE(c && n >= 7, "bigvgan cfg too short");
    BigVGANCfg g;
    int i = 0;
    g.num_mels = c[i++]; g.c0 = c[i++]; g.n_up = c[i++]; g.n_kernels = c[i++];
    g.bias_final = c[i++]; g.tanh_final = c[i++]; g.logscale = c[i++];
    MI_REQUIRE(g.n_up > 0 && g.n_up <= 8 && g.n_kernels > 0 && g.n_kernels <= 4, "bigvgan cfg: bad counts");
    MI_REQUIRE(n >= 7 + 2 * g.n_up + g.n_kernels + 1, "bigvgan cfg too short");
    for (int k = 0; k < g.n_up; ++k) g.rates.push_back(c[i++]);
    for (int k = 0; k < g.n_up; ++k) g.up_k.push_back(c[i++]);
    for (int k = 0; k < g.n_kernels; ++k) g.res_k.push_back(c[i++]);
    g.n_dil = c[i++];
    MI_REQUIRE(g.n_dil > 0 && g.n_dil <= 4 && (n == i + g.n_kernels * g.n_dil || n == i + g.n_kernels * g.n_dil + 2),
               "bigvgan cfg: dilation table");
    for (int k = 0; k < g.n_kernels; ++k) {
        std::vector<int> d;
        for (int l = 0; l < g.n_dil; ++l) d.push_back(c[i++]);
        g.dil.push_back(d);
    }
    if (i + 2 <= n) { g.pre_ln = c[i++]; g.cond = c[i++]; }
    g.hop = 1;
    for (int k = 0; k < g.n_up; ++k) {
        MI_REQUIRE(g.rates[k] > 0 && g.up_k[k] % g.rates[k] == 0 && g.up_k[k] / g.rates[k] <= 2 &&
                       (g.up_k[k] - g.rates[k]) % 2 == 0,
                   "bigvgan: ConvTranspose1d needs kernel == stride or 2*stride, and (kernel - stride) even");
        g.hop *= g.rates[k];
    }
    MI_REQUIRE(g.c0 % (1 << g.n_up) == 0, "bigvgan: initial channel not divisible");
    return g;
}

int64_t bigvgan_param_count(const BigVGANCfg& g) {
    int64_t n = (int64_t)g.c0 * g.num_mels * 7 + g.c0 + (g.pre_ln ? 2 * g.num_mels : 0);
    for (int i = 0; i < g.n_up; ++i) {
        const int64_t cin = g.c0 >> i, cout = g.c0 >> (i + 1);
        n += cin * cout * g.up_k[i] + cout;
        for (int j = 0; j < g.n_kernels; ++j)
            n += (int64_t)g.n_dil * 2 * (cout * cout * g.res_k[j] + cout) + (int64_t)2 * g.n_dil * 2 * cout;
    }
    const int64_t cl = g.c0 >> g.n_up;
    n += 2 * cl + cl * 7 + (g.bias_final ? 1 : 0);
    return n;
}

// Conv1d weight (Co,Ci,k) fp32 -> [co][tap][ci_pad]
static void relayout_conv(const float* w, int Co, int Ci, int k, int Cip, std::vector<float>& out) {
    out.assign((size_t)Co * k * Cip, 0.f);
    for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < Ci; ++ci)
            for (int j = 0; j < k; ++j) out[((size_t)co * k + j) * Cip + ci] = w[((size_t)co * Ci + ci) * k + j];
}
// ConvTranspose1d weight (Ci,Co,k), k = taps*u  ->  [n = r*Co + co][tap][ci].
// out[tau] = sum_{s,j: tau + p = s*u + j} x[s] W[:,:,j]; with q = (tau+p)/u, r = (tau+p)%u the contributing taps are
// j = r + m*u (m = 0..taps-1) at s = q - m; as a conv over rows q - (taps-1) + t, tap t uses j = r + (taps-1-t)*u.
static void relayout_convt(const float* w, int Ci, int Co, int u, int k, std::vector<float>& out) {
    const int taps = k / u;
    out.assign((size_t)u * Co * taps * Ci, 0.f);
    for (int r = 0; r < u; ++r)
        for (int co = 0; co < Co; ++co)
            for (int t = 0; t < taps; ++t)
                for (int ci = 0; ci < Ci; ++ci) {
                    const size_t n = (size_t)r * Co + co;
                    out[(n * taps + t) * Ci + ci] = w[((size_t)ci * Co + co) * k + r + (taps - 1 - t) * u];
                }
}

static void make_snake(const float* a, const float* b, int C, bool logscale, DevBuf& d_alpha, DevBuf& d_ib, hipStream_t s) {
    std::vector<float> al(C), ib(C);
    for (int c = 0; c < C; ++c) {
        const float av = logscale ? expf(a[c]) : a[c];
        const float bv = logscale ? expf(b[c]) : b[c];
        al[c] = av;
        ib[c] = 1.0f / (bv + 1e-9f);
    }
    upload_f32(d_alpha, al.data(), C, s);
    upload_f32(d_ib, ib.data(), C, s);
}

BigVGAN::BigVGAN(const BigVGANCfg& g, const float* w, int64_t nw, int dt, int dev) : cfg(g), dtype(dt), device(dev) {
    MI_REQUIRE(dt == MI_F32 || dt == MI_F16 || dt == MI_BF16, "bigvgan: bad dtype");
    MI_REQUIRE(nw == bigvgan_param_count(g), "bigvgan: weight blob size does not match the config");
    MI_HIP(hipSetDevice(dev));
    MI_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    if (const char* e = std::getenv("MI355TTS_NO_FUSED_AA")) use_fused = !(e[0] == '1');
    if (const char* e = std::getenv("MI355TTS_FUSED_MAX_C")) fused_max_c = std::atoi(e);
    const int vec = 16 / (int)dtype_size(dt);
    MI_REQUIRE((g.c0 >> g.n_up) % vec == 0, "bigvgan: last-stage channels must be a multiple of the 16-byte vector");
    mel_pad = round_up(g.num_mels, vec);
    std::vector<float> tmp;
    const float* p = w;
    if (g.pre_ln) {
        MI_REQUIRE(g.num_mels % 4 == 0 && g.num_mels <= 2048 && g.num_mels % vec == 0, "bigvgan: LayerNorm width");
        upload_f32(ln_w, p, g.num_mels, stream); p += g.num_mels;
        upload_f32(ln_b, p, g.num_mels, stream); p += g.num_mels;
    }
    relayout_conv(p, g.c0, g.num_mels, 7, mel_pad, tmp);
    upload_as(pre.w, tmp.data(), tmp.size(), dt, stream);
    p += (size_t)g.c0 * g.num_mels * 7;
    upload_f32(pre.b, p, g.c0, stream);
    pre.hb.assign(p, p + g.c0);
    p += g.c0;
    stages.resize(g.n_up);
    for (int i = 0; i < g.n_up; ++i) {
        Stage& st = stages[i];
        st.cin = g.c0 >> i; st.cout = g.c0 >> (i + 1); st.u = g.rates[i];
        st.k = g.up_k[i];
        relayout_convt(p, st.cin, st.cout, st.u, st.k, tmp);
        upload_as(st.up.w, tmp.data(), tmp.size(), dt, stream);
        p += (size_t)st.cin * st.cout * g.up_k[i];
        upload_f32(st.up.b, p, st.cout, stream);
        st.up.hb.assign(p, p + st.cout);
        p += st.cout;
        st.blocks.resize(g.n_kernels);
        for (int j = 0; j < g.n_kernels; ++j) {
            AmpBlock& bk = st.blocks[j];
            bk.k = g.res_k[j];
            bk.c1.resize(g.n_dil); bk.c2.resize(g.n_dil); bk.acts.resize(2 * g.n_dil);
            const int C = st.cout;
            for (int l = 0; l < g.n_dil; ++l) {
                relayout_conv(p, C, C, bk.k, C, tmp);
                upload_as(bk.c1[l].w, tmp.data(), tmp.size(), dt, stream);
                p += (size_t)C * C * bk.k;
                upload_f32(bk.c1[l].b, p, C, stream); p += C;
                relayout_conv(p, C, C, bk.k, C, tmp);
                upload_as(bk.c2[l].w, tmp.data(), tmp.size(), dt, stream);
                p += (size_t)C * C * bk.k;
                upload_f32(bk.c2[l].b, p, C, stream); p += C;
            }
            for (int m = 0; m < 2 * g.n_dil; ++m) {
                make_snake(p, p + C, C, g.logscale, bk.acts[m].alpha, bk.acts[m].inv_beta, stream);
                p += 2 * C;
            }
        }
    }
    const int cl = g.c0 >> g.n_up;
    make_snake(p, p + cl, cl, g.logscale, post_act.alpha, post_act.inv_beta, stream);
    p += 2 * cl;
    // conv_post (1, C, 7) -> [7][C] fp32
    tmp.assign((size_t)7 * cl, 0.f);
    for (int c = 0; c < cl; ++c)
        for (int j = 0; j < 7; ++j) tmp[(size_t)j * cl + c] = p[(size_t)c * 7 + j];
    upload_f32(post_w, tmp.data(), tmp.size(), stream);
    p += (size_t)cl * 7;
    post_bias = g.bias_final ? *p++ : 0.f;
    MI_REQUIRE(p - w == nw, "bigvgan: weight walk mismatch");
}

BigVGAN::~BigVGAN() {
    for (hipStream_t& q : side) if (q) (void)hipStreamDestroy(q);
    if (ev_x) (void)hipEventDestroy(ev_x);
    for (hipEvent_t& e : ev_last) if (e) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
}

// Streams for the AMP blocks of a stage.  bigvgan_streams: 1 = everything on the engine's stream (rounds 1-3); 2 = blocks 1.. of
// the stages whose AMP halves are separate AA and conv launches (C > 96) on side streams; 3 = every stage.
// fp16, mel (8,100,512), same box (tools/r3/bigvgan_streams_ab.py, profiles/r3/s6_bigvgan_streams_ab.txt): 16.8 / 16.2 / 16.07 ms per
// forward for 1 / 2 / 3, waveforms identical bit for bit.
static std::atomic<long> g_bigvgan_streams = 3;
bool bigvgan_set_option(const char* key, long v) {
    if (std::string(key) != "bigvgan_streams") return false;
    g_bigvgan_streams = std::max(1L, std::min(3L, v));
    return true;
}

void BigVGAN::ensure_side(int n) {
    MI_REQUIRE(n <= MAX_SIDE, "bigvgan: more resblock kernels than side streams");
    if (!ev_x) MI_HIP(hipEventCreateWithFlags(&ev_x, hipEventDisableTiming));
    for (int j = 0; j <= n; ++j) if (!ev_last[j]) MI_HIP(hipEventCreateWithFlags(&ev_last[j], hipEventDisableTiming));
    for (int j = 0; j < n; ++j) {
        if (!side[j]) MI_HIP(hipStreamCreateWithFlags(&side[j], hipStreamNonBlocking));
        for (DevBuf* b : {&sT1[j], &sT2[j], &sP[j], &sQ[j]}) b->ensure(bT1.bytes);
    }
}

void BigVGAN::ensure_workspace(int B, int F) {
    if (B == ws_B && F == ws_F) return;
    const size_t es = dtype_size(dtype);
    size_t max_elems = (size_t)B * F * std::max(cfg.c0, mel_pad);
    long T = F;
    for (int i = 0; i < cfg.n_up; ++i) {
        T *= cfg.rates[i];
        max_elems = std::max(max_elems, (size_t)B * (size_t)(T + 30) * (size_t)(cfg.c0 >> (i + 1)));
    }
    for (DevBuf* b : {&bIN, &bX, &bT1, &bT2, &bP, &bQ}) b->ensure(max_elems * es);
    d_mel.ensure((size_t)B * cfg.num_mels * F * 4);
    d_out_f32.ensure((size_t)B * (T + 30) * 4);
    d_out_i16.ensure((size_t)B * (T + 30) * 2);
    ws_B = B; ws_F = F;
}

void BigVGAN::conv(const ConvW& cw, const void* x, void* out, int B, int T, int Cin, int Cout, int k, int dil,
                   const void* res, float alpha, int accumulate) {
    ConvGemm p;
    p.dtype = dtype; p.x = x; p.w = cw.w.p; p.bias = cw.b.as<float>(); p.out = out; p.res = res;
    p.B = B; p.T_in = T; p.M = T; p.N = Cout; p.Cin = Cin; p.taps = k; p.dil = dil; p.pad = (k * dil - dil) / 2;
    p.x_bstride = (long)T * Cin; p.x_rstride = Cin; p.out_bstride = (long)T * Cout; p.out_rstride = Cout;
    p.alpha = alpha; p.accumulate = accumulate;
    launch_conv_gemm(p, ls ? ls : stream);
}

void BigVGAN::aa_conv(const SnakeP& sp, const ConvW& cw, const void* x, void* out, int B, int T, int C, int k, int dil,
                      const void* res, float alpha, int accumulate) {
    AAConv a;
    a.dtype = dtype; a.x = x; a.w = cw.w.p; a.bias = cw.b.as<float>(); a.snake_alpha = sp.alpha.as<float>();
    a.snake_inv_beta = sp.inv_beta.as<float>(); a.out = out; a.res = res; a.B = B; a.T = T; a.C = C; a.k = k; a.dil = dil;
    a.alpha = alpha; a.accumulate = accumulate;
    launch_aa_conv(a, ls ? ls : stream);
}

void BigVGAN::aa(const SnakeP& sp, const void* x, void* y, int B, int T, int C, int post) {
    AAAct a;
    a.dtype = dtype; a.x = x; a.y = y; a.alpha = sp.alpha.as<float>(); a.inv_beta = sp.inv_beta.as<float>();
    a.B = B; a.T = T; a.C = C; a.post = post;
    launch_aa_act(a, ls ? ls : stream);
}

long BigVGAN::total_cond() const {
    long n = cfg.c0;
    for (int i = 0; i < cfg.n_up; ++i) n += cfg.c0 >> (i + 1);
    return n;
}

void BigVGAN::run(const float* mel, int B, int F, float* out_f32, int16_t* out_i16, int mem) {
    MI_REQUIRE(mel && B > 0 && F > 0, "bigvgan_forward: bad arguments");
    MI_REQUIRE(out_f32 || out_i16, "bigvgan_forward: no output buffer");
    MI_REQUIRE(!cfg.pre_ln && !cfg.cond, "bigvgan_forward: this handle is an IndexTTS graph-F vocoder, use mi_bigvgan_forward_latent");
    MI_REQUIRE((long)F * cfg.hop < (1L << 30), "bigvgan_forward: too many frames");
    MI_HIP(hipSetDevice(device));
    ensure_workspace(B, F);
    const float* dmel = mel;
    if (mem == MI_HOST) {
        MI_HIP(hipMemcpyAsync(d_mel.p, mel, (size_t)B * cfg.num_mels * F * 4, hipMemcpyHostToDevice, stream));
        dmel = d_mel.as<float>();
    }
    // mel (B,100,F) -> channels-last padded (B,F,mel_pad)
    launch_ncl_to_nlc(dmel, bT1.p, B, cfg.num_mels, F, mel_pad, dtype, stream);
    body(bT1.p, B, F, nullptr, out_f32, out_i16, mem);
}

// IndexTTS graph F (IndexTTS/Export_IndexTTS.py:300-314): gpt.final_norm(latent[:-2]) -> conv_pre + cond_pre -> stages with
// + cond_i after each upsampler -> post activation -> conv_post(+bias) -> tanh -> int16.
void BigVGAN::run_latent(const float* latent, int T_codes, const float* conds, long n_conds, float* out_f32,
                         int16_t* out_i16, int mem) {
    MI_REQUIRE(cfg.pre_ln && cfg.cond, "bigvgan_forward_latent: handle was not created with the IndexTTS graph-F flags");
    MI_REQUIRE(latent && conds && T_codes >= 3 && (out_f32 || out_i16), "bigvgan_forward_latent: bad arguments (needs >= 3 latent rows)");
    MI_REQUIRE(n_conds == total_cond(), "bigvgan_forward_latent: conditioning vector length");
    // the LayerNorm below writes rows of num_mels elements; conv_pre reads them with row stride mel_pad
    MI_REQUIRE(mel_pad == cfg.num_mels, "bigvgan_forward_latent: the latent width must be a multiple of the 16-byte vector");
    MI_HIP(hipSetDevice(device));
    const int F = T_codes - 2;                          // the reference drops the last two latent rows
    ensure_workspace(1, F);
    std::vector<float> hc((size_t)n_conds);
    if (mem == MI_HOST) std::memcpy(hc.data(), conds, (size_t)n_conds * 4);
    else MI_HIP(hipMemcpy(hc.data(), conds, (size_t)n_conds * 4, hipMemcpyDeviceToHost));
    std::vector<const float*> ptrs(cfg.n_up + 1);
    long off = 0;
    for (int i = 0; i < cfg.n_up; ++i) { ptrs[i] = hc.data() + off; off += cfg.c0 >> (i + 1); }
    ptrs[cfg.n_up] = hc.data() + off;                   // cond_pre (c0 values) is last, like the graph's input order
    const float* dl = latent;
    if (mem == MI_HOST) {
        d_latent.ensure((size_t)T_codes * cfg.num_mels * 4);
        MI_HIP(hipMemcpyAsync(d_latent.p, latent, (size_t)F * cfg.num_mels * 4, hipMemcpyHostToDevice, stream));
        dl = d_latent.as<float>();
    }
    launch_rownorm(NORM_LN_AFFINE, dl, bT1.p, dtype, ln_w.as<float>(), ln_b.as<float>(), F, cfg.num_mels, 1e-5f, stream);
    body(bT1.p, 1, F, ptrs.data(), out_f32, out_i16, mem);
}

static void eff_bias(ConvW& cw, const float* cond, hipStream_t s) {
    std::vector<float> e(cw.hb);
    for (size_t i = 0; i < e.size(); ++i) e[i] += cond[i];
    cw.b_eff.ensure(e.size() * 4);
    MI_HIP(hipMemcpyAsync(cw.b_eff.p, e.data(), e.size() * 4, hipMemcpyHostToDevice, s));
    MI_HIP(hipStreamSynchronize(s));                    // `e` is a stack temporary
}

// x0: channels-last (B, F, mel_pad) in the engine dtype, held in bT1
void BigVGAN::body(const void* x0, int B, int F, const float* const* cond, float* out_f32, int16_t* out_i16, int mem) {
    const long Tout = (long)F * cfg.hop + 30;
    ls = nullptr;
    DevBuf* IN = &bIN;     // holds the running stage input/output
    DevBuf* X = &bX;
    {
        const float* bias = pre.b.as<float>();
        if (cond) { eff_bias(pre, cond[cfg.n_up], stream); bias = pre.b_eff.as<float>(); }
        ConvGemm p;
        p.dtype = dtype; p.x = x0; p.w = pre.w.p; p.bias = bias; p.out = IN->p;
        p.B = B; p.T_in = F; p.M = F; p.N = cfg.c0; p.Cin = mel_pad; p.taps = 7; p.dil = 1; p.pad = 3;
        p.x_bstride = (long)F * mel_pad; p.x_rstride = mel_pad; p.out_bstride = (long)F * cfg.c0; p.out_rstride = cfg.c0;
        launch_conv_gemm(p, stream);
    }
    int T = F;
    const float inv_nk = 1.0f / (float)cfg.n_kernels;
    for (int i = 0; i < cfg.n_up; ++i) {
        Stage& st = stages[i];
        const int Tn = T * st.u, C = st.cout;
        {   // ConvTranspose1d as a `taps`-tap conv producing u*Cout phase-major channels (taps = k/u)
            const int taps = st.k / st.u;
            const float* bias = st.up.b.as<float>();
            if (cond) { eff_bias(st.up, cond[i], stream); bias = st.up.b_eff.as<float>(); }
            ConvGemm p;
            p.dtype = dtype; p.x = IN->p; p.w = st.up.w.p; p.bias = bias; p.out = X->p;
            p.B = B; p.T_in = T; p.M = T + taps - 1; p.N = st.u * C; p.Cin = st.cin; p.taps = taps; p.dil = 1; p.pad = taps - 1;
            p.x_bstride = (long)T * st.cin; p.x_rstride = st.cin; p.out_bstride = (long)Tn * C; p.out_rstride = C;
            p.epi = EPI_CONVT; p.u = st.u; p.Cout = C; p.padT = (st.k - st.u) / 2; p.T_out = Tn;
            launch_conv_gemm(p, stream);
        }
        // AMP blocks: XS(=IN) = 1/3 * sum_j block_j(X).  The blocks only meet in X (read) and in IN (block 0 writes it, blocks
        // 1.. accumulate, in block order: 16-bit storage rounds after every accumulate).  With side streams block j runs on its
        // own stream with its own scratch, launches issued dilation-major so that all blocks are in flight from the start; the
        // LAST conv of block j waits for the last conv of block j - 1 (events), so the values are those of the one-stream order,
        // bit for bit.  What it buys: the AA launches (VALU / HBM) of one block run beside the conv GEMM (matrix cores, one
        // workgroup per CU with LDS and registers to spare) of another.
        const bool fused = use_fused && C <= fused_max_c;
        const long ns = g_bigvgan_streams;
        const int nside = (cfg.n_kernels > 1 && cfg.n_kernels - 1 <= MAX_SIDE && (ns >= 3 || (ns == 2 && !fused))) ? cfg.n_kernels - 1 : 0;
        if (nside) {
            ensure_side(nside);
            MI_HIP(hipEventRecord(ev_x, stream));
            for (int j = 0; j < nside; ++j) MI_HIP(hipStreamWaitEvent(side[j], ev_x, 0));
        }
        std::vector<const void*> curs(cfg.n_kernels, X->p);
        static const bool dbg_sync = [] { const char* e = std::getenv("MI355TTS_BV_SYNC"); return e && e[0] == '1'; }();
        // issue order: dilation-major with side streams (every block has its own scratch), block-major on one stream (the
        // blocks share T1 / T2 / P / Q there)
        const int n_it = cfg.n_dil * cfg.n_kernels;
        for (int it = 0; it < n_it; ++it) {
            const int l = nside ? it / cfg.n_kernels : it % cfg.n_dil;
            const int j = nside ? it % cfg.n_kernels : it / cfg.n_dil;
            const bool last = l == cfg.n_dil - 1;
            {
                AmpBlock& bk = st.blocks[j];
                const bool sidej = nside && j > 0;
                hipStream_t sj = sidej ? side[j - 1] : stream;
                ls = sj;
                void* t1 = sidej ? sT1[j - 1].p : bT1.p;
                void* t2 = sidej ? sT2[j - 1].p : bT2.p;
                void* pp = sidej ? sP[j - 1].p : bP.p;
                void* qq = sidej ? sQ[j - 1].p : bQ.p;
                const void* cur = curs[j];
                void* dst = last ? IN->p : ((l & 1) ? qq : pp);
                if (fused) {
                    // HBM-bound stages: AA folded into the conv's operand staging (5 tensor passes instead of 9)
                    aa_conv(bk.acts[2 * l], bk.c1[l], cur, t2, B, Tn, C, bk.k, cfg.dil[j][l], nullptr, 1.f, 0);
                    if (nside && last && j > 0) MI_HIP(hipStreamWaitEvent(sj, ev_last[j - 1], 0));
                    aa_conv(bk.acts[2 * l + 1], bk.c2[l], t2, dst, B, Tn, C, bk.k, 1, cur, last ? inv_nk : 1.f, last && j > 0);
                } else {
                    aa(bk.acts[2 * l], cur, t1, B, Tn, C, 0);
                    conv(bk.c1[l], t1, t2, B, Tn, C, C, bk.k, cfg.dil[j][l], nullptr, 1.f, 0);
                    aa(bk.acts[2 * l + 1], t2, t1, B, Tn, C, 0);
                    if (nside && last && j > 0) MI_HIP(hipStreamWaitEvent(sj, ev_last[j - 1], 0));
                    conv(bk.c2[l], t1, dst, B, Tn, C, C, bk.k, 1, cur, last ? inv_nk : 1.f, last && j > 0);
                }
                if (nside && last) MI_HIP(hipEventRecord(ev_last[j], sj));
                if (dbg_sync) MI_HIP(hipDeviceSynchronize());
                curs[j] = dst;
            }
        }
        ls = nullptr;
        if (nside) MI_HIP(hipStreamWaitEvent(stream, ev_last[cfg.n_kernels - 1], 0));
        T = Tn;
    }
    const int cl = cfg.c0 >> cfg.n_up;
    aa(post_act, IN->p, bT1.p, B, T, cl, 1);
    float* of = out_f32 ? (mem == MI_DEVICE ? out_f32 : d_out_f32.as<float>()) : nullptr;
    int16_t* oi = out_i16 ? (mem == MI_DEVICE ? out_i16 : d_out_i16.as<int16_t>()) : nullptr;
    launch_conv_post(bT1.p, post_w.as<float>(), post_bias, B, (int)Tout, cl, dtype, cfg.tanh_final, of, oi, stream);
    if (mem == MI_HOST) {
        if (out_f32) MI_HIP(hipMemcpyAsync(out_f32, of, (size_t)B * Tout * 4, hipMemcpyDeviceToHost, stream));
        if (out_i16) MI_HIP(hipMemcpyAsync(out_i16, oi, (size_t)B * Tout * 2, hipMemcpyDeviceToHost, stream));
    }
    MI_HIP(hipStreamSynchronize(stream));
}

// ---------------------------------------------------------------------------------------------
// unit-level entries (tests): fp32 channels-first host tensors in/out
// ---------------------------------------------------------------------------------------------
struct TmpStream {
    hipStream_t s = nullptr;
    TmpStream() { MI_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); }
    ~TmpStream() { if (s) (void)hipStreamDestroy(s); }
};

void unit_aa_activation1d(const float* x, int B, int C, int T, const float* alpha_log, const float* beta_log,
                          int logscale, int post, int dtype, float* y) {
    TmpStream ts;
    DevBuf dx, dxl, dyl, dy, da, db;
    const size_t es = dtype_size(dtype);
    const int To = T + (post ? 30 : 0);
    upload_f32(dx, x, (size_t)B * C * T, ts.s);
    dxl.ensure((size_t)B * C * T * es); dyl.ensure((size_t)B * C * To * es); dy.ensure((size_t)B * C * To * 4);
    make_snake(alpha_log, beta_log, C, logscale, da, db, ts.s);
    launch_ncl_to_nlc(dx.as<float>(), dxl.p, B, C, T, C, dtype, ts.s);
    AAAct a;
    a.dtype = dtype; a.x = dxl.p; a.y = dyl.p; a.alpha = da.as<float>(); a.inv_beta = db.as<float>();
    a.B = B; a.T = T; a.C = C; a.post = post;
    launch_aa_act(a, ts.s);
    launch_nlc_to_ncl(dyl.p, dy.as<float>(), B, C, To, dtype, ts.s);
    MI_HIP(hipMemcpyAsync(y, dy.p, (size_t)B * C * To * 4, hipMemcpyDeviceToHost, ts.s));
    MI_HIP(hipStreamSynchronize(ts.s));
}

// fused AA-SnakeBeta -> Conv1d(C, C, k, dil, "same") (+ residual): one half of an AMPBlock1 iteration (bigvgan.py:132-140)
void unit_aa_conv1d(const float* x, int B, int C, int T, const float* alpha_log, const float* beta_log, int logscale,
                    const float* w, const float* bias, int k, int dil, const float* res, int dtype, int repeat, float* y) {
    TmpStream ts;
    DevBuf dx, dxl, dw, db, da, dib, dr, drl, dyl, dy;
    const size_t es = dtype_size(dtype), n = (size_t)B * C * T;
    upload_f32(dx, x, n, ts.s);
    dxl.ensure(n * es); dyl.ensure(n * es); dy.ensure(n * 4);
    launch_ncl_to_nlc(dx.as<float>(), dxl.p, B, C, T, C, dtype, ts.s);
    if (res) {
        upload_f32(dr, res, n, ts.s);
        drl.ensure(n * es);
        launch_ncl_to_nlc(dr.as<float>(), drl.p, B, C, T, C, dtype, ts.s);
    }
    make_snake(alpha_log, beta_log, C, logscale, da, dib, ts.s);
    std::vector<float> wl;
    relayout_conv(w, C, C, k, C, wl);
    upload_as(dw, wl.data(), wl.size(), dtype, ts.s);
    upload_f32(db, bias, C, ts.s);
    AAConv a;
    a.dtype = dtype; a.x = dxl.p; a.w = dw.p; a.bias = db.as<float>(); a.snake_alpha = da.as<float>();
    a.snake_inv_beta = dib.as<float>(); a.out = dyl.p; a.res = res ? drl.p : nullptr; a.B = B; a.T = T; a.C = C; a.k = k; a.dil = dil;
    for (int r = 0; r < std::max(repeat, 1); ++r) launch_aa_conv(a, ts.s);      // repeat > 1: tuning / stress runs
    launch_nlc_to_ncl(dyl.p, dy.as<float>(), B, C, T, dtype, ts.s);
    MI_HIP(hipMemcpyAsync(y, dy.p, n * 4, hipMemcpyDeviceToHost, ts.s));
    MI_HIP(hipStreamSynchronize(ts.s));
}

void unit_conv1d(const float* x, int B, int Cin, int T, const float* w, const float* bias, int Cout, int k, int dil,
                 int padding, int groups, int dtype, float* y) {
    MI_REQUIRE(Cin % groups == 0 && Cout % groups == 0, "conv1d: groups");
    TmpStream ts;
    const int vec = 16 / (int)dtype_size(dtype);
    const int cig = Cin / groups, cog = Cout / groups;
    MI_REQUIRE(groups == 1 || cig % vec == 0, "conv1d: grouped conv needs Cin/groups % vec == 0");
    const int Cp = groups == 1 ? round_up(Cin, vec) : Cin;
    const int cigp = groups == 1 ? Cp : cig;
    const int To = T + 2 * padding - dil * (k - 1);
    MI_REQUIRE(To > 0, "conv1d: empty output");
    DevBuf dx, dxl, dw, db, dyl, dy;
    const size_t es = dtype_size(dtype);
    upload_f32(dx, x, (size_t)B * Cin * T, ts.s);
    dxl.ensure((size_t)B * T * Cp * es);
    launch_ncl_to_nlc(dx.as<float>(), dxl.p, B, Cin, T, Cp, dtype, ts.s);
    std::vector<float> wl;
    relayout_conv(w, Cout, cig, k, cigp, wl);       // (Cout, Cin/g, k) -> [co][tap][ci]
    upload_as(dw, wl.data(), wl.size(), dtype, ts.s);
    if (bias) upload_f32(db, bias, Cout, ts.s);
    dyl.ensure((size_t)B * To * Cout * es); dy.ensure((size_t)B * To * Cout * 4);
    ConvGemm p;
    p.dtype = dtype; p.x = dxl.p; p.w = dw.p; p.bias = bias ? db.as<float>() : nullptr; p.out = dyl.p;
    p.B = B; p.G = groups; p.T_in = T; p.M = To; p.N = cog; p.Cin = cigp; p.taps = k; p.dil = dil; p.pad = padding;
    p.x_bstride = (long)T * Cp; p.x_rstride = Cp; p.x_goff = cigp; p.out_bstride = (long)To * Cout; p.out_rstride = Cout;
    SkWorkspace skw;                                    // lets one-tap shapes reach the stream-K kernel (gemm_sk.hip)
    skw.ensure(512, ts.s);
    skw.attach(p);
    DevBuf dw3;
    if (dtype == MI_F32 && k == 1 && groups == 1 && gemm_x3_enabled()) {     // ... and the bf16x3 kernel (gemm_x3.hip)
        dw3.ensure((size_t)3 * Cout * cigp * 2);
        split3_planes(dw.as<float>(), dw3.p, (long)Cout * cigp, ts.s);
        p.w3 = dw3.p;
    }
    DevBuf dxp, dw3p;
    if (p.w3 && gemm_x3p_enabled() && padding == 0 && Cout % 128 == 0 && cigp % 32 == 0) {     // ... and its panel-plane form (gemm_x3p.hip)
        const int np = x3p_planes();
        dxp.ensure((size_t)x3p_bytes((long)B * T, cigp, np)); dw3p.ensure((size_t)x3p_bytes(Cout, cigp, np));
        x3p_split_rows(dxl.as<float>(), Cp, dxp.p, B * T, cigp, ts.s, np);
        x3p_split_rows(dw.as<float>(), cigp, dw3p.p, Cout, cigp, ts.s, np);
        p.xp = dxp.p; p.w3p = dw3p.p; p.np = np;
    }
    launch_conv_gemm(p, ts.s);
    launch_nlc_to_ncl(dyl.p, dy.as<float>(), B, Cout, To, dtype, ts.s);
    MI_HIP(hipMemcpyAsync(y, dy.p, (size_t)B * Cout * To * 4, hipMemcpyDeviceToHost, ts.s));
    MI_HIP(hipStreamSynchronize(ts.s));
}

void unit_conv_transpose1d(const float* x, int B, int Cin, int T, const float* w, const float* bias, int Cout, int k,
                           int stride, int padding, int dtype, float* y) {
    MI_REQUIRE(k % stride == 0 && k / stride <= 2 && (k - stride) % 2 == 0 && padding == (k - stride) / 2,
               "conv_transpose1d: kernel == stride or 2*stride with padding == (kernel - stride)/2 (the BigVGAN upsamplers)");
    const int vec = 16 / (int)dtype_size(dtype);
    MI_REQUIRE(Cin % vec == 0, "conv_transpose1d: Cin % vec");
    TmpStream ts;
    const int To = T * stride;
    DevBuf dx, dxl, dw, db, dyl, dy;
    const size_t es = dtype_size(dtype);
    upload_f32(dx, x, (size_t)B * Cin * T, ts.s);
    dxl.ensure((size_t)B * T * Cin * es);
    launch_ncl_to_nlc(dx.as<float>(), dxl.p, B, Cin, T, Cin, dtype, ts.s);
    std::vector<float> wl;
    relayout_convt(w, Cin, Cout, stride, k, wl);
    upload_as(dw, wl.data(), wl.size(), dtype, ts.s);
    if (bias) upload_f32(db, bias, Cout, ts.s);
    dyl.ensure((size_t)B * To * Cout * es); dy.ensure((size_t)B * To * Cout * 4);
    ConvGemm p;
    p.dtype = dtype; p.x = dxl.p; p.w = dw.p; p.bias = bias ? db.as<float>() : nullptr; p.out = dyl.p;
    const int taps = k / stride;
    p.B = B; p.T_in = T; p.M = T + taps - 1; p.N = stride * Cout; p.Cin = Cin; p.taps = taps; p.dil = 1; p.pad = taps - 1;
    p.x_bstride = (long)T * Cin; p.x_rstride = Cin; p.out_bstride = (long)To * Cout; p.out_rstride = Cout;
    p.epi = EPI_CONVT; p.u = stride; p.Cout = Cout; p.padT = padding; p.T_out = To;
    launch_conv_gemm(p, ts.s);
    launch_nlc_to_ncl(dyl.p, dy.as<float>(), B, Cout, To, dtype, ts.s);
    MI_HIP(hipMemcpyAsync(y, dy.p, (size_t)B * Cout * To * 4, hipMemcpyDeviceToHost, ts.s));
    MI_HIP(hipStreamSynchronize(ts.s));
}

}  // namespace mi
