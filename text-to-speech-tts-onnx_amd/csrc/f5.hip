// f5.hip — F5-TTS on gfx950: weight packing, load-time tables and the three launch sequences
// (preprocess / sampler / decode) that stand in for the reference's three ONNX graphs.
//
//   preprocess  = F5Preprocess.forward   F5_TTS/Export_F5.py:117-141
//   dit_eval    = DiT.forward            F5_TTS/modeling_modified/F5/dit.py:205-220
//   steps       = F5Transformer.forward  F5_TTS/Export_F5.py:167-182   (x nfe-1, all on device)
//   decode      = F5Decode.forward       F5_TTS/Export_F5.py:193-203
//
// Device-resident state: noise (fp32), conditioning (fp32 master + dtype copy inside the `cat`
// GEMM operand), the residual stream X (always fp32), and per-layer operands in the engine dtype.
// Everything that does not depend on the sample is hoisted to load time: time-MLP table, AdaLN
// modulation for all (step, block) pairs, RoPE tables, STFT / mel / ISTFT bases.
// Batching: U utterances of equal length N are laid out as batch 2U (2u = cond, 2u+1 = uncond).
#include "f5.h"
#include <cstdlib>

namespace mi {

static int rup(int a, int b) { return (a + b - 1) / b * b; }

F5Cfg parse_f5_cfg(const int32_t* ci, int ni, const float* cf, int nf) {
    MI_REQUIRE(ci && ni >= 21 && ni <= 24 && cf && (nf == 2 || nf == 3),
               "f5 cfg: expected 21 ints (+ up to 3 optional: fp32 arithmetic, mel type, AdaLN fold) + 2 floats (+ the optional attention score scale)");
    F5Cfg c;
    int i = 0;
    c.dim = ci[i++]; c.depth = ci[i++]; c.heads = ci[i++]; c.dim_head = ci[i++]; c.ff_mult = ci[i++]; c.mel = ci[i++];
    c.text_dim = ci[i++]; c.vocab = ci[i++]; c.conv_layers = ci[i++]; c.conv_mult = ci[i++]; c.pos_k = ci[i++];
    c.pos_g = ci[i++]; c.freq_dim = ci[i++]; c.nfe = ci[i++]; c.max_len = ci[i++]; c.n_fft = ci[i++]; c.hop = ci[i++];
    c.sr = ci[i++]; c.vd = ci[i++]; c.vi = ci[i++]; c.vlayers = ci[i++];
    if (ni > 21) c.f32_arith = ci[21];
    if (ni > 22) c.mel_type = ci[22];
    if (ni > 23) c.ln_fold = ci[23];
    MI_REQUIRE(c.f32_arith == ARITH_DEFAULT || c.f32_arith == ARITH_NATIVE || c.f32_arith == ARITH_PAIRS || c.f32_arith == ARITH_BF16X3,
               "f5 cfg: fp32 arithmetic is -1 (process default), 0 (native fp32 MFMA), 2 (fp16 pairs) or 3 (three bf16 planes)");
    MI_REQUIRE((c.mel_type == 0 || c.mel_type == 1) && c.ln_fold >= -1 && c.ln_fold <= 1, "f5 cfg: mel type is 0 (vocos) or 1 (bigvgan); AdaLN fold is -1, 0 or 1");
    c.cfg_strength = cf[0]; c.sway = cf[1];
    if (nf == 3) { c.score_scale = cf[2]; MI_REQUIRE(c.score_scale > 0.f && c.score_scale <= 1e4f, "f5 cfg: attention score scale"); }
    MI_REQUIRE(c.dim == c.heads * c.dim_head, "f5 cfg: dim != heads*dim_head");
    MI_REQUIRE(c.dim_head == 64, "f5: the attention kernel is built for head_dim 64");
    MI_REQUIRE(c.dim % 32 == 0 && c.dim <= 2048 && c.text_dim % 8 == 0 && c.mel % 4 == 0, "f5 cfg: widths");
    MI_REQUIRE(c.dim % c.pos_g == 0 && (c.dim / c.pos_g) % 8 == 0 && c.pos_k % 2 == 1, "f5 cfg: pos conv");
    MI_REQUIRE(c.nfe >= 2 && c.nfe <= 1024 && c.max_len >= 16 && c.depth > 0, "f5 cfg: sampler");
    MI_REQUIRE(c.n_fft % 8 == 0 && c.n_fft % c.hop == 0 && c.vd % 4 == 0 && c.vi % 4 == 0 && c.vd <= 2048, "f5 cfg: vocoder");
    MI_REQUIRE(c.freq_dim % 2 == 0 && c.conv_mult >= 1 && c.text_dim * c.conv_mult <= 4096, "f5 cfg: text");
    return c;
}

int64_t f5_param_count(const F5Cfg& c) {
    const int64_t d = c.dim, td = c.text_dim, ff = c.ff(), ti = td * c.conv_mult, cin = c.cat_dim();
    int64_t n = d * c.freq_dim + d + d * d + d;
    n += (int64_t)(c.vocab + 1) * td;
    n += (int64_t)c.conv_layers * (td * 7 + td + td + td + ti * td + ti + ti + ti + td * ti + td);
    n += d * cin + d + 2 * (d * (d / c.pos_g) * c.pos_k + d);
    n += (int64_t)c.depth * (6 * d * d + 6 * d + 4 * (d * d + d) + ff * d + ff + d * ff + d);
    n += 2 * d * d + 2 * d + (int64_t)c.mel * d + c.mel;
    const int64_t vd = c.vd, vi = c.vi;
    n += vd * c.mel * 7 + vd + 2 * vd;
    n += (int64_t)c.vlayers * (vd * 7 + vd + 2 * vd + vi * vd + vi + vd * vi + vd);
    n += 2 * vd + (int64_t)(c.n_fft + 2) * vd + (c.n_fft + 2);
    return n;
}

static void up_lin(Lin& L, const float* w, const float* b, int n, int k, int dt, hipStream_t s) {     // host sources
    L.n = n; L.k = k;
    upload_as(L.w, w, (size_t)n * k, dt, s);
    if (b) upload_f32(L.b, b, n, s);
}
static void put_lin(BlobReader& R, Lin& L, const float* w, const float* b, int n, int k, int dt) {     // blob sources
    L.n = n; L.k = k;
    R.put(L.w, w, (size_t)n * k, dt);
    if (b) R.put(L.b, b, n, MI_F32);
}
// Conv1d weight (Co, Ci, k) -> [co][tap][ci]
static std::vector<float> relayout(const float* w, int Co, int Ci, int k) {
    std::vector<float> o((size_t)Co * k * Ci);
    for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < Ci; ++ci)
            for (int j = 0; j < k; ++j) o[((size_t)co * k + j) * Ci + ci] = w[((size_t)co * Ci + ci) * k + j];
    return o;
}
static inline float siluf(float v) { return v / (1.f + expf(-v)); }
static inline float round_f16(float v) { return (float)(f16)v; }

F5::F5(const F5Cfg& c, const float* w, int64_t nw, int dt, int dev, int mem) : cfg(c), dtype(dt), device(dev) {
    arith_kind = dt == MI_F32 ? c.f32_arith : ARITH_DEFAULT;
    arith = arith_for(arith_kind);
    ArithScope arith_scope(arith);        // the load-time launches and the format of the weight planes follow the engine's arithmetic
    np = x3p_planes();
    MI_REQUIRE(dt == MI_F32 || dt == MI_F16 || dt == MI_BF16, "f5: bad dtype");
    MI_REQUIRE(c.score_scale == 1.f || dt == MI_F16, "f5: the attention score scale (reference fp16-transformer form) needs an f16 engine");
    MI_REQUIRE(nw == f5_param_count(c), "f5: weight blob size does not match the config");
    MI_HIP(hipSetDevice(dev));
    MI_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    if (const char* e = std::getenv("MI355TTS_NO_GRAPH")) use_graph = !(e[0] == '1');
    hipStream_t s = stream;
    d_sat.ensure(64);
    MI_HIP(hipMemsetAsync(d_sat.p, 0, 64, s));
    const int d = c.dim, td = c.text_dim, ff = c.ff(), ti = td * c.conv_mult, cin = c.cat_dim();
    {
        const char* e = std::getenv("MI355TTS_CAT_PAD");
        if (dt == MI_F32 && !(e && e[0] == '0')) cat_pad = rup(cin, 64) - cin;
    }
    const float* p = w;
    auto take = [&](size_t n) { const float* r = p; p += n; return r; };          // blob ranges (host or device memory)
    BlobReader R(mem, s);

    // ---- time MLP table (Export_F5.py:145-165), host fp32 -------------------------------------------
    const float* tw0 = R.host(take((size_t)d * c.freq_dim), (size_t)d * c.freq_dim); const float* tb0 = R.host(take(d), d);
    const float* tw2 = R.host(take((size_t)d * d), (size_t)d * d); const float* tb2 = R.host(take(d), d);
    const int steps = c.nfe, half = c.freq_dim / 2;
    std::vector<float> ts(steps);
    for (int i = 0; i < steps; ++i) {
        const float t = (float)i / (float)(steps - 1);
        ts[i] = t + c.sway * (cosf(3.14159265358979323846f * 0.5f * t) - 1.f + t);
    }
    h_delta.resize(steps - 1);
    for (int i = 0; i + 1 < steps; ++i) h_delta[i] = ts[i + 1] - ts[i];
    upload_f32(delta_t, h_delta.data(), steps - 1, s);
    h_time_expand.assign((size_t)steps * d, 0.f);
    {
        const float ef = logf(10000.f) / (float)(half - 1);
        std::vector<float> emb(c.freq_dim), hid(d);
        for (int i = 0; i < steps; ++i) {
            for (int j = 0; j < half; ++j) {
                const float a = ts[i] * (1000.0f * expf((float)j * -ef));
                emb[j] = sinf(a); emb[half + j] = cosf(a);
            }
            for (int o = 0; o < d; ++o) {
                float acc = tb0[o];
                for (int j = 0; j < c.freq_dim; ++j) acc += tw0[(size_t)o * c.freq_dim + j] * emb[j];
                hid[o] = siluf(acc);
            }
            for (int o = 0; o < d; ++o) {
                float acc = tb2[o];
                for (int j = 0; j < d; ++j) acc += tw2[(size_t)o * d + j] * hid[j];
                h_time_expand[(size_t)i * d + o] = acc;
            }
        }
    }
    // ---- text embedding -----------------------------------------------------------------------------
    R.put(text_emb, take((size_t)(c.vocab + 1) * td), (size_t)(c.vocab + 1) * td, MI_F32);
    {   // precompute_freqs_cis(text_dim, max_len) (modules.py:196-207)
        std::vector<float> pos((size_t)c.max_len * td);
        std::vector<float> fr(td / 2);
        for (int j = 0; j < td / 2; ++j) fr[j] = 1.0f / powf(10000.0f, (float)(2 * j) / (float)td);
        for (int n = 0; n < c.max_len; ++n)
            for (int j = 0; j < td / 2; ++j) {
                const float a = (float)n * fr[j];
                pos[(size_t)n * td + j] = cosf(a); pos[(size_t)n * td + td / 2 + j] = sinf(a);
            }
        upload_f32(text_pos, pos.data(), pos.size(), s);
    }
    tblocks.resize(c.conv_layers);
    for (auto& tb : tblocks) {
        R.put(tb.dw_w, take((size_t)td * 7), (size_t)td * 7, MI_F32);
        R.put(tb.dw_b, take(td), td, MI_F32);
        R.put(tb.ln_w, take(td), td, MI_F32);
        R.put(tb.ln_b, take(td), td, MI_F32);
        const float* w1 = take((size_t)ti * td); const float* b1 = take(ti);
        put_lin(R, tb.pw1, w1, b1, ti, td, MI_F32);
        R.put(tb.grn_g, take(ti), ti, MI_F32);
        R.put(tb.grn_b, take(ti), ti, MI_F32);
        const float* w2 = take((size_t)td * ti); const float* b2 = take(td);
        put_lin(R, tb.pw2, w2, b2, td, ti, MI_F32);
    }
    // ---- input embedding -------------------------------------------------------------------------------
    {
        const float* pw = take((size_t)d * cin); const float* pb = take(d);
        if (cat_ld() == cin) put_lin(R, in_proj, pw, pb, d, cin, dt);
        else {
            // fp32 engines: K = 2 * mel + text_dim (712) padded with zero columns to whole 64-deep chunks (768), the row stride of
            // the cat buffer with it (its pad columns are zeroed once, ensure_workspace): the layer then runs on the panel-plane
            // kernel like the block matrices instead of the native-fp32 small-tile kernel (46.9 -> ~25 us per evaluation)
            const int kp = cat_ld();
            const float* hw = R.host(pw, (size_t)d * cin); const float* hb = R.host(pb, d);
            std::vector<float> wp((size_t)d * kp, 0.f);
            for (int o = 0; o < d; ++o) std::copy(hw + (size_t)o * cin, hw + (size_t)(o + 1) * cin, wp.begin() + (size_t)o * kp);
            up_lin(in_proj, wp.data(), hb, d, kp, dt, s);
        }
        const int cg = d / c.pos_g;
        const size_t gw = (size_t)d * cg * c.pos_k;                 // the k31 grouped convs are re-laid out by host code
        const float* w1 = R.host(take(gw), gw); const float* b1 = R.host(take(d), d);
        auto r1 = relayout(w1, d, cg, c.pos_k);
        up_lin(gconv1, r1.data(), b1, d, cg * c.pos_k, dt, s);
        const float* w2 = R.host(take(gw), gw); const float* b2 = R.host(take(d), d);
        auto r2 = relayout(w2, d, cg, c.pos_k);
        up_lin(gconv2, r2.data(), b2, d, cg * c.pos_k, dt, s);
        if (dt != MI_F32 && cg == 64 && c.pos_k >= 8 && c.pos_k <= 127) {
            // 16-bit engines: the same convolution from weight images in the engine's type (gconv16.hip)
            for (Lin* L : {&gconv1, &gconv2}) {
                L->w3p.ensure(gconv16_image_bytes(c.pos_g, c.pos_k));
                gconv16_build_weights(L->w.p, dt, L->w3p.p, c.pos_g, c.pos_k, s);
            }
        }
        if (dt == MI_F32 && cg == 64 && c.pos_k >= 8 && c.pos_k <= 127) {
            // the pair-split form of the position convolution (gconv_pairs.hip) takes its weights pre-split: once here, not per launch
            for (Lin* L : {&gconv1, &gconv2}) {
                L->w3p.ensure(gconv_pairs_planes_bytes(c.pos_g, c.pos_k));
                gconv_pairs_split_weights(L->w.as<float>(), L->w3p.p, c.pos_g, c.pos_k, s);
            }
        }
    }
    // ---- DiT blocks + hoisted AdaLN modulation table -------------------------------------------------
    mod_ld = (long)c.depth * 6 * d + 2 * d;
    mod.ensure((size_t)steps * mod_ld * 4);
    DevBuf d_silu_t, d_tmpw, d_tmpb;
    {
        std::vector<float> st(h_time_expand.size());
        for (size_t i = 0; i < st.size(); ++i) st[i] = siluf(h_time_expand[i]);
        upload_f32(d_silu_t, st.data(), st.size(), s);
    }
    auto mod_gemm = [&](const float* mw, const float* mb, int n, long col) {
        R.put(d_tmpw, mw, (size_t)n * d, MI_F32);
        R.put(d_tmpb, mb, n, MI_F32);
        ConvGemm g;
        g.dtype = MI_F32; g.x = d_silu_t.p; g.w = d_tmpw.p; g.bias = d_tmpb.as<float>(); g.out = mod.as<float>() + col;
        g.B = 1; g.T_in = steps; g.M = steps; g.N = n; g.Cin = d; g.x_rstride = d; g.x_bstride = (long)steps * d;
        g.out_rstride = mod_ld; g.out_bstride = (long)steps * mod_ld;
        launch_conv_gemm(g, s);
        MI_HIP(hipStreamSynchronize(s));
    };
    blocks.resize(c.depth);
    for (int i = 0; i < c.depth; ++i) {
        Block& bk = blocks[i];
        const float* mw = take((size_t)6 * d * d); const float* mb = take((size_t)6 * d);
        mod_gemm(mw, mb, 6 * d, (long)i * 6 * d);
        bk.qkv.n = 3 * d; bk.qkv.k = d;                              // q | k | v rows stacked into one GEMM weight
        bk.qkv.w.ensure((size_t)3 * d * d * dtype_size(dt)); bk.qkv.b.ensure((size_t)3 * d * 4);
        for (int t3 = 0; t3 < 3; ++t3) {
            const float* wt = take((size_t)d * d); const float* bt = take(d);
            if (t3 == 0) { R.put(bk.qkv.w, wt, (size_t)d * d, dt, 0); R.put(bk.qkv.b, bt, d, MI_F32, 0); }
            else { R.put(bk.qkv.w, wt, (size_t)d * d, dt, (size_t)t3 * d * d); R.put(bk.qkv.b, bt, d, MI_F32, (size_t)t3 * d); }
        }
        const float* wo = take((size_t)d * d); const float* bo = take(d);
        put_lin(R, bk.o, wo, bo, d, d, dt);
        const float* w1 = take((size_t)ff * d); const float* b1 = take(ff);
        put_lin(R, bk.ff1, w1, b1, ff, d, dt);
        const float* w2 = take((size_t)d * ff); const float* b2 = take(d);
        put_lin(R, bk.ff2, w2, b2, d, ff, dt);
        if (dt == MI_F32) {
            // the four big matrices of the block once more as three bf16 planes each (gemm_x3.hip: the fallback of the panel-plane
            // kernel); their panel planes are built by set_arith() below, once the number format is settled
            for (Lin* L : {&bk.qkv, &bk.o, &bk.ff1, &bk.ff2}) {
                L->w3.ensure((size_t)3 * L->n * L->k * 2);
                split3_planes(L->w.as<float>(), L->w3.p, (long)L->n * L->k, s);
                launch_absmax(L->w.as<float>(), (long)L->n * L->k, d_sat.as<unsigned>() + 1, s);
            }
        }
    }
    if (dt == MI_F32) {
        // fp16 pairs hold |w| <= 65504 only (the weight operand is not clamped by the kernels): a checkpoint with a
        // larger entry in one of the big matrices gets the exact three-plane bf16 split instead, which spans the fp32 exponent range
        unsigned bits = 0;
        MI_HIP(hipMemcpyAsync(&bits, d_sat.as<unsigned>() + 1, 4, hipMemcpyDeviceToHost, s));
        MI_HIP(hipStreamSynchronize(s));
        float wmax; std::memcpy(&wmax, &bits, 4);
        int kind = arith_kind;
        if (np == 2 && gemm_x3p_enabled() && !(wmax < 65504.f)) kind = ARITH_BF16X3;
        set_arith(kind);
    }
    // ---- AdaLN fold: W (1 + scale) and W shift + b for every (step, block), QKV and FF1 (gemm_epilogue.h) ----
    fold_built = false;
    // On unless switched off.  16-bit engines finish the row statistics with one tiny launch per norm (ln_finalize_kernel) instead
    // of in every consumer epilogue: with 12 / 8 column tiles of 256 the consumers re-read 197 MB of partials per QKV launch at 8
    // utterances (+26 us per launch, profiles/r4/adaln_fold_ab.txt)
    const bool want_fold = c.ln_fold != 0;
    if (want_fold && d >= 1024 && d % 128 == 0 && ff % 64 == 0) {
        ln_blk = (long)6 * d + 2 * ff;
        ln_ld = (long)c.depth * ln_blk;
        ln_tab.ensure((size_t)steps * ln_ld * 4);
        DevBuf G, S, tmpw;
        G.ensure((size_t)steps * d * 4); S.ensure((size_t)steps * d * 4);
        if (dt != MI_F32) tmpw.ensure((size_t)std::max(3 * d, ff) * d * 4);
        for (int i = 0; i < c.depth; ++i) build_ln_tables(i, G, S, tmpw);
        MI_HIP(hipStreamSynchronize(s));
        fold_built = true;
    }
    {
        const float* mw = take((size_t)2 * d * d); const float* mb = take((size_t)2 * d);
        mod_gemm(mw, mb, 2 * d, (long)c.depth * 6 * d);
        const float* pw = take((size_t)c.mel * d); const float* pb = take(c.mel);
        put_lin(R, proj_out, pw, pb, c.mel, d, dt);
        {
            const char* e = std::getenv("MI355TTS_PROJ_PARTS");
            const int parts = e ? std::atoi(e) : 4;
            if (dt == MI_F32 && parts > 1 && parts <= 8 && d % (parts * 32) == 0) {
                proj_parts = parts;
                const int kq = d / parts;
                const float* hw = R.host(pw, (size_t)c.mel * d); const float* hb = R.host(pb, c.mel);
                std::vector<float> wq((size_t)c.mel * d), bq((size_t)parts * c.mel, 0.f);
                for (int q = 0; q < parts; ++q)
                    for (int o = 0; o < c.mel; ++o)
                        std::copy(hw + (size_t)o * d + (size_t)q * kq, hw + (size_t)o * d + (size_t)(q + 1) * kq, wq.begin() + ((size_t)q * c.mel + o) * kq);
                std::copy(hb, hb + c.mel, bq.begin());              // the bias rides on slice 0
                up_lin(proj_out_k, wq.data(), bq.data(), parts * c.mel, kq, dt, s);
            }
        }
    }
    // ---- RoPE tables, rounded through fp16 (Export_F5.py:107-112) --------------------------------------
    {
        const int D = c.dim_head;
        std::vector<float> rc((size_t)c.max_len * D), rs((size_t)c.max_len * D);
        for (int n = 0; n < c.max_len; ++n)
            for (int j = 0; j < D / 2; ++j) {
                const float inv = 1.0f / powf(10000.0f, (float)(2 * j) / (float)D);
                const float a = (float)n * inv;
                const float cc = round_f16(cosf(a)), ss = round_f16(sinf(a));
                rc[(size_t)n * D + 2 * j] = rc[(size_t)n * D + 2 * j + 1] = cc;
                rs[(size_t)n * D + 2 * j] = rs[(size_t)n * D + 2 * j + 1] = ss;
            }
        upload_f32(rope_cos, rc.data(), rc.size(), s);
        upload_f32(rope_sin, rs.data(), rs.size(), s);
        // packed copy: the table values ARE fp16 numbers and both elements of a rotation pair share one angle, so
        // (cos, sin) as two halfs per pair holds the same information in a quarter of the bytes
        std::vector<f16> pk((size_t)c.max_len * D);
        for (int n = 0; n < c.max_len; ++n)
            for (int j = 0; j < D / 2; ++j) {
                pk[(size_t)n * D + 2 * j] = (f16)rc[(size_t)n * D + 2 * j];
                pk[(size_t)n * D + 2 * j + 1] = (f16)rs[(size_t)n * D + 2 * j];
            }
        rope_pack.ensure(pk.size() * sizeof(f16));
        MI_HIP(hipMemcpyAsync(rope_pack.p, pk.data(), pk.size() * sizeof(f16), hipMemcpyHostToDevice, s));
        MI_HIP(hipStreamSynchronize(s));
    }
    // ---- STFT kernels (STFT_Process.py:86-98; fp32 evaluation order of torch) + HTK mel fbank ---------
    {
        const int nf = c.n_fft, nb = c.nb();
        std::vector<float> win(nf);
        const float wstep = (float)(2.0 * M_PI / nf);
        for (int n = 0; n < nf; ++n) win[n] = cosf((float)n * wstep) * -0.5f + 0.5f;      // torch.hann_window (periodic)
        std::vector<float> sw((size_t)2 * nb * nf);
        const float two_pi = (float)(2.0 * M_PI);
        for (int f = 0; f < nb; ++f)
            for (int t = 0; t < nf; ++t) {
                const float om = ((two_pi * (float)f) * (float)t) / (float)nf;
                sw[(size_t)f * nf + t] = cosf(om) * win[t];
                sw[(size_t)(nb + f) * nf + t] = -sinf(om) * win[t];
            }
        if (c.mel_type == 1) {
            // bigvgan-type front end (modules.py:30-72): torch.stft's DFT of the windowed frame — the exact basis, not the fp32
            // evaluation order of STFT_Process's conv kernels (that quirk belongs to the vocos-type graph)
            for (int f = 0; f < nb; ++f)
                for (int t = 0; t < nf; ++t) {
                    const double om = 2.0 * M_PI * (double)((long)f * t % nf) / nf;
                    sw[(size_t)f * nf + t] = (float)(std::cos(om) * (double)win[t]);
                    sw[(size_t)(nb + f) * nf + t] = (float)(-std::sin(om) * (double)win[t]);
                }
        }
        upload_f32(stft_w, sw.data(), sw.size(), s);
        // melscale_fbanks(nb, 0, sr/2, mel, sr, None, 'htk') -> stored [mel][ldm] zero padded
        const int ldm = rup(nb, 8);
        std::vector<float> fb((size_t)c.mel * ldm, 0.f);
        if (c.mel_type == 1) {
            // librosa.filters.mel(sr, n_fft, n_mels, fmin = 0, fmax = sr / 2): slaney scale (linear below 1 kHz, log above), slaney
            // norm (2 / band width) — published definition, librosa is un-vendored (modules.py:18,45)
            const double f_sp = 200.0 / 3, min_log_hz = 1000.0, logstep = std::log(6.4) / 27.0, min_log_mel = min_log_hz / f_sp;
            auto h2m = [&](double f) { return f < min_log_hz ? f / f_sp : min_log_mel + std::log(f / min_log_hz) / logstep; };
            auto m2h = [&](double m) { return m < min_log_mel ? f_sp * m : min_log_hz * std::exp(logstep * (m - min_log_mel)); };
            const double m0 = h2m(0.0), m1 = h2m((double)c.sr / 2);
            std::vector<double> mf(c.mel + 2);
            for (int i = 0; i < c.mel + 2; ++i) mf[i] = m2h(m0 + (m1 - m0) * i / (c.mel + 1));
            for (int m = 0; m < c.mel; ++m) {
                const double enorm = 2.0 / (mf[m + 2] - mf[m]);
                for (int k = 0; k < nb; ++k) {
                    const double fr = (double)c.sr / 2 * k / (nb - 1);
                    const double lower = (fr - mf[m]) / (mf[m + 1] - mf[m]), upper = (mf[m + 2] - fr) / (mf[m + 2] - mf[m + 1]);
                    fb[(size_t)m * ldm + k] = (float)(std::max(0.0, std::min(lower, upper)) * enorm);
                }
            }
        }
        const double m_min = 0.0, m_max = 2595.0 * std::log10(1.0 + (c.sr / 2) / 700.0);
        std::vector<double> fpts(c.mel + 2);
        for (int i = 0; i < c.mel + 2; ++i) {
            const double m = m_min + (m_max - m_min) * i / (c.mel + 1);
            fpts[i] = 700.0 * (std::pow(10.0, m / 2595.0) - 1.0);
        }
        for (int k = 0; k < nb && c.mel_type == 0; ++k) {
            const double fr = (double)(c.sr / 2) * k / (nb - 1);
            for (int m = 0; m < c.mel; ++m) {
                const double down = (fr - fpts[m]) / (fpts[m + 1] - fpts[m]);
                const double up = (fpts[m + 2] - fr) / (fpts[m + 2] - fpts[m + 1]);
                fb[(size_t)m * ldm + k] = (float)std::max(0.0, std::min(down, up));
            }
        }
        upload_f32(fbank, fb.data(), fb.size(), s);
        // ISTFT basis, closed form of window * pinv(fourier_basis*n_fft/hop).T (STFT_Process.py:100-112):
        // stored as GEMM weight [n = sample][r = coefficient], K padded to a multiple of 4.
        const int kp = rup(2 * nb, 4);
        std::vector<float> ib((size_t)nf * kp, 0.f);
        for (int n = 0; n < nf; ++n)
            for (int k = 0; k < nb; ++k) {
                const double sc = ((k == 0 || k == nb - 1) ? 1.0 : 2.0) / nf * ((double)c.hop / nf) * (double)win[n];
                const double ang = 2.0 * M_PI * (double)k * (double)n / nf;
                ib[(size_t)n * kp + k] = (float)(sc * std::cos(ang));
                ib[(size_t)n * kp + nb + k] = (float)(sc * -std::sin(ang));
            }
        istft.n = nf; istft.k = kp;
        upload_f32(istft.w, ib.data(), ib.size(), s);
        // window_sum_inv for max_len frames (STFT_Process.py:114-133), same fp32 accumulation order
        const size_t wl = (size_t)nf + (size_t)c.hop * (c.max_len - 1);
        std::vector<float> ws(wl, 0.f);
        float wmax = 0.f;
        for (int n = 0; n < nf; ++n) wmax = std::max(wmax, fabsf(win[n]));
        std::vector<float> wsq(nf);
        for (int n = 0; n < nf; ++n) { const float wn = win[n] / wmax; wsq[n] = wn * wn; }
        for (int i = 0; i < c.max_len; ++i)
            for (int n = 0; n < nf; ++n) ws[(size_t)i * c.hop + n] += wsq[n];
        for (size_t i = 0; i < wl; ++i) ws[i] = (float)nf / (ws[i] * (float)c.hop + 1e-7f);
        upload_f32(wsi, ws.data(), wl, s);
    }
    // ---- Vocos ----------------------------------------------------------------------------------------------
    {
        const int vd = c.vd, vi = c.vi;
        const size_t ewn = (size_t)vd * c.mel * 7;
        const float* ew = R.host(take(ewn), ewn); const float* eb = R.host(take(vd), vd);
        auto re = relayout(ew, vd, c.mel, 7);
        up_lin(v_embed, re.data(), eb, vd, c.mel * 7, MI_F32, s);
        R.put(v_norm_w, take(vd), vd, MI_F32); R.put(v_norm_b, take(vd), vd, MI_F32);
        vblocks.resize(c.vlayers);
        for (auto& vb_ : vblocks) {
            R.put(vb_.dw_w, take((size_t)vd * 7), (size_t)vd * 7, MI_F32); R.put(vb_.dw_b, take(vd), vd, MI_F32);
            R.put(vb_.n_w, take(vd), vd, MI_F32); R.put(vb_.n_b, take(vd), vd, MI_F32);
            const float* w1 = take((size_t)vi * vd); const float* b1 = take(vi);
            put_lin(R, vb_.pw1, w1, b1, vi, vd, MI_F32);
            const float* w2 = take((size_t)vd * vi); const float* b2 = take(vd);
            put_lin(R, vb_.pw2, w2, b2, vd, vi, MI_F32);
        }
        R.put(v_fnorm_w, take(vd), vd, MI_F32); R.put(v_fnorm_b, take(vd), vd, MI_F32);
        const float* hw = take((size_t)(c.n_fft + 2) * vd); const float* hb = take(c.n_fft + 2);
        put_lin(R, v_head, hw, hb, c.n_fft + 2, vd, MI_F32);
    }
    MI_HIP(hipStreamSynchronize(s));
    MI_REQUIRE(p - w == nw, "f5: weight walk mismatch");
}

F5::~F5() {
    drop_graphs();
    if (stream) (void)hipStreamDestroy(stream);
}

// The engine's fp32 arithmetic (fp32 engines): ARITH_PAIRS = fp16 {hi, lo} pairs (22-bit operands, |a| <= 65504), ARITH_BF16X3 =
// three bf16 planes (exact, whole fp32 exponent range), ARITH_NATIVE = v_mfma_f32_32x32x2_f32, ARITH_DEFAULT = whatever the
// process-wide options say.  Builds the panel planes of the big matrices in that format (the fp32 rows stay resident, so the
// format can change later: take_saturation() -> set_arith(ARITH_BF16X3)).
void F5::set_arith(int kind) {
    arith_kind = kind;
    arith = arith_for(kind);
    ArithScope sc(arith);
    np = x3p_planes();
    drop_graphs();
    if (dtype != MI_F32) return;
    const bool planes = gemm_x3p_enabled();
    auto build_planes = [&](Lin* L) {
        if (!planes || L->n % 128 != 0 || L->k % 32 != 0) { L->w3p.release(); return; }
        L->w3p.ensure((size_t)x3p_bytes(L->n, L->k, np));
        x3p_split_rows(L->w.as<float>(), L->k, L->w3p.p, L->n, L->k, stream, np);
    };
    for (Block& bk : blocks)
        for (Lin* L : {&bk.qkv, &bk.o, &bk.ff1, &bk.ff2}) build_planes(L);
    build_planes(&in_proj);
    if (ws_U > 0) {       // the activation planes are sized by the format
        const long rows = (long)2 * ws_U * ws_N;
        Ap.ensure((size_t)x3p_bytes(rows, cfg.dim, np)); Ap2.ensure((size_t)x3p_bytes(rows, cfg.ff(), np)); ApN.ensure((size_t)x3p_bytes(rows, cfg.dim, np));
    }
    MI_HIP(hipStreamSynchronize(stream));
}

// block i: [ W_qkv (1 + sc_a) | W_qkv sh_a + b_qkv | W_ff1 (1 + sc_m) | W_ff1 sh_m + b_ff1 ] for every step, as GEMMs of the
// (steps x d) modulation slices against the weights the engine multiplies by (16-bit engines: the ROUNDED weights)
void F5::build_ln_tables(int i, DevBuf& G, DevBuf& S, DevBuf& tmpw) {
    const int d = cfg.dim, ff = cfg.ff(), steps = cfg.nfe;
    Block& bk = blocks[i];
    float* tab = ln_tab.as<float>() + (size_t)i * ln_blk;
    auto run = [&](const Lin& L, long col_shift, long col_scale, float* out_p, float* out_c) {
        launch_ln_gather(mod.as<float>(), mod_ld, (long)i * 6 * d + col_scale, (long)i * 6 * d + col_shift, G.as<float>(), S.as<float>(), steps, d, stream);
        const float* w32 = L.w.as<float>();
        if (dtype != MI_F32) { launch_cast_to_f32(L.w.p, dtype, tmpw.as<float>(), (long)L.n * L.k, stream); w32 = tmpw.as<float>(); }
        ConvGemm g;
        g.dtype = MI_F32; g.x = G.p; g.w = w32; g.bias = nullptr; g.out = out_p;
        g.B = 1; g.T_in = steps; g.M = steps; g.N = L.n; g.Cin = d; g.x_rstride = d; g.x_bstride = (long)steps * d;
        g.out_rstride = ln_ld; g.out_bstride = (long)steps * ln_ld;
        launch_conv_gemm(g, stream);
        g.x = S.p; g.bias = L.b.as<float>(); g.out = out_c;
        launch_conv_gemm(g, stream);
        MI_HIP(hipStreamSynchronize(stream));       // G / S / tmpw are reused by the next call
    };
    run(bk.qkv, 0, d, tab, tab + 3 * d);                                    // shift_msa at 0, scale_msa at d (modules.py:303)
    run(bk.ff1, 3 * d, 4 * d, tab + 6 * d, tab + 6 * d + ff);               // shift_mlp at 3d, scale_mlp at 4d
}

bool F5::take_saturation() {
    if (!d_sat.p) return false;
    int flag = 0;
    MI_HIP(hipMemcpy(&flag, d_sat.p, 4, hipMemcpyDeviceToHost));
    if (!flag) return false;
    MI_HIP(hipMemset(d_sat.p, 0, 4));
    return true;
}

// every C-ABI entry, after its stream synchronisation: the stream-K watchdog (it used to be looked at only on the
// paths that validate text ids) and the text-id flag
void F5::finish_call() {
    check_text_ids();
}

void F5::ensure_workspace(int U, int N) {
    if (U <= ws_U && N <= ws_N) return;
    drop_graphs();                         // captured graphs hold raw workspace pointers
    const int Um = std::max(U, ws_U), Nm = std::max(N, ws_N);
    const F5Cfg& c = cfg;
    const size_t es = dtype_size(dtype);
    const size_t rows = (size_t)2 * Um * Nm;
    sk.ensure(1024, stream);     // 64 MB: stream-K slots (64 KB) and the split-tail slabs of gemm_ph8.hip (256 KB)
    {
        // key-sliced attention (attention.hip): used while a launch has fewer than 1024 (fp32) / 512 (16-bit) 128-query
        // workgroups, i.e. fewer than 2048 64-query tiles — whatever the LARGEST batch this handle has seen (the running
        // maxima Um, Nm only size the activations: a handle that first ran 8 utterances must still slice a later single
        // one).  Reserved once: 2048 tiles x 4 slices x (8192 + 256) floats = 277 MB of the 288 GB.
        const long tiles = 2048;
        if (tiles > attn_cnt_n) {
            attn_ws_floats = tiles * 4 * (2 * 32 * 64 + 2 * 64 * 2);
            attn_ws.ensure((size_t)attn_ws_floats * 4);
            attn_cnt.ensure((size_t)tiles * 4);
            MI_HIP(hipMemsetAsync(attn_cnt.p, 0, (size_t)tiles * 4, stream));
            attn_cnt_n = tiles;
        }
    }
    d_noise.ensure((size_t)Um * Nm * c.mel * 4);
    d_cmt.ensure((size_t)Um * Nm * c.cond_dim() * 4);
    d_cmtd.ensure((size_t)Um * Nm * c.cond_dim() * 4);
    if (rows * cat_ld() * es > cat.bytes) {
        cat.ensure(rows * cat_ld() * es);
        if (cat_ld() != c.cat_dim()) MI_HIP(hipMemsetAsync(cat.p, 0, cat.bytes, stream));      // the pad columns stay zero
    }
    h32.ensure(rows * c.dim * 4); hT.ensure(rows * c.dim * es); c1.ensure(rows * c.dim * es);
    X.ensure(rows * c.dim * 4); Ub.ensure(rows * c.dim * es);
    qb.ensure(rows * c.dim * es); Ob.ensure(rows * c.dim * es);
    {   // V may be stored transposed with the key axis padded to a multiple of 8; the pad columns are read (and
        // multiplied by exactly-zero probabilities), so they must hold finite values: zero the buffer once.
        // fp32 engines: K and V^T may be kept as three bf16 planes with the key axis padded to the 64-key stage
        // (ConvGemm::kv_planes, attention.hip KVP): 6 bytes per element instead of 4
        const size_t keypad = (size_t)(Nm + 63) / 64 * 64;
        size_t kbytes = rows * c.dim * es, vbytes = (size_t)2 * Um * (size_t)(Nm + 8) * c.dim * es;
        if (dtype == MI_F32) { kbytes = std::max(kbytes, (size_t)2 * Um * keypad * c.dim * 6); vbytes = std::max(vbytes, (size_t)2 * Um * keypad * c.dim * 6); }
        if (kbytes > kb.bytes) { kb.ensure(kbytes); MI_HIP(hipMemsetAsync(kb.p, 0, kbytes, stream)); }
        if (vbytes > vb.bytes) { vb.ensure(vbytes); MI_HIP(hipMemsetAsync(vb.p, 0, vbytes, stream)); }
    }
    Hff.ensure(rows * c.ff() * es);
    if (dtype == MI_F32) {
        Ap.ensure((size_t)x3p_bytes((long)rows, c.dim, np)); Ap2.ensure((size_t)x3p_bytes((long)rows, c.ff(), np));
        if (fold_built) ApN.ensure((size_t)x3p_bytes((long)rows, c.dim, np));
    }
    if (fold_built) { ln_stats.ensure((rows + 128) * (size_t)(c.dim / LN_BLK) * 2 * 4); if (dtype != MI_F32) ln_fin.ensure((rows + 128) * 8); }
    pred.ensure(rows * c.mel * 4 * proj_parts);
    if (proj_parts > 1) pred_sum.ensure(rows * c.mel * 4);
    // preprocess temporaries
    const int ti = c.text_dim * c.conv_mult;
    p_ids.ensure((size_t)Um * Nm * 4);
    p_tx.ensure((size_t)2 * Um * Nm * c.text_dim * 4); p_ty.ensure((size_t)2 * Um * Nm * c.text_dim * 4);
    p_ty2.ensure((size_t)2 * Um * Nm * ti * 4); p_ss.ensure((size_t)2 * Um * ti * 4);
    // decode temporaries
    const size_t fr = (size_t)Um * Nm;
    v_h.ensure(fr * c.vd * 4); v_z.ensure(fr * c.vd * 4); v_z2.ensure(fr * c.vi * 4);
    v_s.ensure(fr * (c.n_fft + 2) * 4); v_c.ensure(fr * istft.k * 4); v_fr.ensure(fr * c.n_fft * 4);
    v_outf.ensure(fr * c.hop * 4); v_outi.ensure(fr * c.hop * 2);
    ws_U = Um; ws_N = Nm;
}

// One linear layer over rows (the front end, the vocoder, and the DiT layers of the ROWS form).  fp32 layers that have panel
// planes of their weights get their rows split by a separate pass here; the SAME ConvGemm object answers the eligibility
// question and is launched.
void F5::gemm(int dt, const void* x, long xb, long xr, int K, const Lin& L, void* out, int odt, long ob, long orr, int B,
              int M, int act, const void* res, const float* gate) {
    ConvGemm g;
    g.dtype = dt; g.out_dtype = odt; g.x = x; g.w = L.w.p; g.w3 = L.w3.p; g.bias = L.b.p ? L.b.as<float>() : nullptr; g.out = out;
    g.res = res; g.gate = gate; g.gate_bstride = 0;
    g.B = B; g.T_in = M; g.M = M; g.N = L.n; g.Cin = K; g.taps = 1;
    g.x_bstride = xb; g.x_rstride = xr; g.out_bstride = ob; g.out_rstride = orr; g.act = act;
    g.sat = d_sat.as<int>();
    sk.attach(g);
    if (B > 1 && xb == (long)M * xr && ob == (long)M * orr) {      // rows of all batch items are contiguous: one M axis
        g.B = 1; g.T_in = B * M; g.M = B * M;                      // (no per-item tile padding: 2252 rows -> 9 tiles, not 10)
    }
    if (dt == MI_F32 && L.w3p.p && g.B == 1 && Ap.p && gemm_x3p_enabled()) {
        g.xp = K <= cfg.dim ? Ap.p : Ap2.p; g.w3p = L.w3p.p; g.np = np;
        if (gemm_x3p_would_run(g)) x3p_split_rows((const float*)x, xr, const_cast<void*>(g.xp), g.M, K, stream, np, d_sat.as<int>());
        else { g.xp = nullptr; g.w3p = nullptr; }
    }
    launch_conv_gemm(g, stream);
}

// counter-based N(0,1) for the case where the caller does not inject noise (the reference draws it inside
// ORT graph A, Export_F5.py:131, which cannot be reproduced outside ORT anyway)
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static void host_normal(float* out, size_t n, uint64_t seed) {
    for (size_t i = 0; i < n; ++i) {
        const uint64_t a = splitmix64(seed * 0x2545F4914F6CDD1Dull + 2 * i), b = splitmix64(seed * 0x2545F4914F6CDD1Dull + 2 * i + 1);
        const double u1 = ((double)(a >> 11) + 1.0) / 9007199254740993.0, u2 = (double)(b >> 11) / 9007199254740992.0;
        out[i] = (float)(std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2));
    }
}

// Graph A for U utterances at once (audio (U,L), text_ids (U,T), noise_in (U,N,mel) or null): every stage is one launch
// over the batch, nothing waits for the host.  Text ids are validated on the host when they live there; device-resident
// ids are checked by the kernel (flag in p_err, read at the caller's next synchronisation via check_text_ids()).
int F5::preprocess(int U, const int16_t* audio, long L, const int32_t* text_ids, int T, int N,
                   const float* noise_in, uint64_t seed, int mem) {
    const F5Cfg& c = cfg;
    MI_REQUIRE(audio && text_ids && U >= 1 && L >= c.n_fft / 2 + 1 && T >= 0, "f5_preprocess: bad arguments");
    // frames of the prompt: vocos-type mel = reflect pad n_fft / 2, L / hop + 1 frames (Export_F5.py:122-125); bigvgan-type =
    // reflect pad (n_fft - hop) / 2, center = False (modules.py:54-68)
    const int mel_pad = c.mel_type == 1 ? (c.n_fft - c.hop) / 2 : c.n_fft / 2;
    MI_REQUIRE(L + 2 * mel_pad >= c.n_fft && L > mel_pad, "f5_preprocess: audio shorter than one STFT frame");
    const int R = (int)((L + 2 * mel_pad - c.n_fft) / c.hop) + 1;
    MI_REQUIRE(N >= R && N >= T && N <= c.max_len, "f5_preprocess: max_duration must be >= ref frames, >= text length and <= max_signal_length");
    if (mem == MI_HOST)
        for (long i = 0; i < (long)U * T; ++i) {
            const long id = (long)text_ids[i] + 1;
            MI_REQUIRE(id >= 0 && id <= c.vocab, "f5_preprocess: text id out of range");
        }
    MI_HIP(hipSetDevice(device));
    ensure_workspace(U, N);
    hipStream_t s = stream;
    const int nf = c.n_fft, nb = c.nb(), ldm = rup(nb, 8), cd = c.cond_dim();
    const long Lp = L + nf;
    // ---- audio -> STFT -> |.| -> mel -> log ---------------------------------------------------------------
    p_audio.ensure((size_t)U * L * 2); p_pad.ensure((size_t)U * Lp * 4 + 64);
    p_spec.ensure((size_t)U * R * 2 * nb * 4); p_mag.ensure((size_t)U * R * ldm * 4); p_mel.ensure((size_t)U * R * c.mel * 4);
    p_tid.ensure((size_t)U * std::max(T, 1) * 4); p_err.ensure(4);
    const int16_t* da = audio;
    const int32_t* dt_ids = text_ids;
    if (mem == MI_HOST) {
        MI_HIP(hipMemcpyAsync(p_audio.p, audio, (size_t)U * L * 2, hipMemcpyHostToDevice, s));
        da = p_audio.as<int16_t>();
        if (T > 0) MI_HIP(hipMemcpyAsync(p_tid.p, text_ids, (size_t)U * T * 4, hipMemcpyHostToDevice, s));
        dt_ids = p_tid.as<int32_t>();
    }
    MI_HIP(hipMemsetAsync(p_err.p, 0, 4, s));
    stft(da, U, L, mel_pad);
    launch_spec_mag(p_spec.as<float>(), p_mag.as<float>(), U * R, nb, ldm, c.mel_type == 1 ? 1e-9f : 0.f, s);
    {
        ConvGemm g;
        g.dtype = MI_F32; g.x = p_mag.p; g.w = fbank.p; g.out = p_mel.p;
        g.B = 1; g.T_in = U * R; g.M = U * R; g.N = c.mel; g.Cin = ldm; g.x_rstride = ldm; g.out_rstride = c.mel;
        launch_conv_gemm(g, s);
    }
    launch_logmel(p_mel.as<float>(), d_cmt.as<float>(), d_cmtd.as<float>(), U, N, R, c.mel, cd, s);
    // ---- text ids (+1, zero = filler) -> embedding + pos -> ConvNeXtV2 blocks (text / drop branches of all utterances
    //      as batch 2U: slab 2u = text, 2u+1 = drop) ---------------------------------------------------------------
    const int td = c.text_dim, ti = td * c.conv_mult;
    int* dids = p_ids.as<int>();
    launch_text_ids(dt_ids, dids, U, T, N, c.vocab, p_err.as<int>(), s);
    float* tx = p_tx.as<float>(); float* ty = p_ty.as<float>(); float* ty2 = p_ty2.as<float>();
    launch_text_gather(dids, text_emb.as<float>(), text_pos.as<float>(), tx, U, N, td, s);
    const int B2 = 2 * U;
    for (auto& tb : tblocks) {
        launch_dwconv7(tx, ty, tb.dw_w.as<float>(), tb.dw_b.as<float>(), B2, N, td, s);
        launch_rownorm(NORM_LN_AFFINE, ty, ty, MI_F32, tb.ln_w.as<float>(), tb.ln_b.as<float>(), (long)B2 * N, td, 1e-6f, s);
        gemm(MI_F32, ty, (long)N * td, td, td, tb.pw1, ty2, MI_F32, (long)N * ti, ti, B2, N, ACT_GELU_ERF);
        launch_grn(ty2, p_ss.as<float>(), tb.grn_g.as<float>(), tb.grn_b.as<float>(), B2, N, ti, s);
        gemm(MI_F32, ty2, (long)N * ti, ti, ti, tb.pw2, tx, MI_F32, (long)N * td, td, B2, N, ACT_NONE, tx);
        launch_mask_rows(dids, tx, B2, N, td, s);
    }
    for (int u = 0; u < U; ++u) {
        launch_copy2d(tx + (size_t)(2 * u) * N * td, td, d_cmt.as<float>() + (size_t)u * N * cd + c.mel, cd, N, td, MI_F32, s);
        launch_copy2d(tx + (size_t)(2 * u + 1) * N * td, td, d_cmtd.as<float>() + (size_t)u * N * cd + c.mel, cd, N, td, MI_F32, s);
    }
    // ---- noise ------------------------------------------------------------------------------------------------
    const size_t nn = (size_t)N * c.mel;
    if (noise_in) {
        MI_HIP(hipMemcpyAsync(d_noise.p, noise_in, (size_t)U * nn * 4, mem == MI_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s));
    } else {
        h_noise.resize((size_t)U * nn);                 // member: must outlive the asynchronous copy
        for (int u = 0; u < U; ++u) host_normal(h_noise.data() + (size_t)u * nn, nn, seed + (uint64_t)u);
        MI_HIP(hipMemcpyAsync(d_noise.p, h_noise.data(), h_noise.size() * 4, hipMemcpyHostToDevice, s));
    }
    return R;
}

// STFT-B (STFT_Process.py:144-157): reflect pad n_fft/2, frames of n_fft at stride hop against the hann*cos / -hann*sin
// kernels as one framed GEMM -> p_spec [u][frame][re(nb) | im(nb)]
void F5::stft(const int16_t* audio_dev, int U, long L, int pad) {
    const F5Cfg& c = cfg;
    if (pad < 0) pad = c.n_fft / 2;
    const int nf = c.n_fft, nb = c.nb(), R = (int)((L + 2 * pad - nf) / c.hop) + 1;
    const long Lp = L + 2 * pad;
    launch_pad_reflect(audio_dev, p_pad.as<float>(), U, L, pad, stream);
    ConvGemm g;      // framed GEMM: row f = padded[f*hop : f*hop + n_fft]
    g.dtype = MI_F32; g.x = p_pad.p; g.w = stft_w.p; g.out = p_spec.p;
    g.B = U; g.T_in = R; g.M = R; g.N = 2 * nb; g.Cin = nf; g.x_rstride = c.hop; g.x_bstride = Lp;
    g.out_rstride = 2 * nb; g.out_bstride = (long)R * 2 * nb;
    launch_conv_gemm(g, stream);
}

// after a stream synchronisation: did the text-id kernel see an id outside the embedding table?
void F5::check_text_ids() {
    if (sk.tripped()) {             // a stream-K fix-up spin hit its bound (gemm_x3p.hip): the results of this call are not to be trusted
        recover();
        MI_REQUIRE(false, "f5: a split-tile hand-off timed out on the device (workspace reset; results of this call discarded)");
    }
    if (!p_err.p) return;
    int flag = 0;
    MI_HIP(hipMemcpy(&flag, p_err.p, 4, hipMemcpyDeviceToHost));
    if (flag != 0) {
        MI_HIP(hipMemset(p_err.p, 0, 4));       // reported once: later calls on the handle with valid inputs must not inherit it
        MI_REQUIRE(false, "f5_preprocess: text id out of range");
    }
}

void F5::load_cond(const float* noise, const float* cmt, const float* cmtd, int U, int N, int mem) {
    const F5Cfg& c = cfg;
    MI_REQUIRE(U > 0 && N > 0 && N <= c.max_len, "f5: bad batch / length");
    MI_HIP(hipSetDevice(device));
    ensure_workspace(U, N);
    const hipMemcpyKind kind = mem == MI_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    if (noise && noise != d_noise.p) MI_HIP(hipMemcpyAsync(d_noise.p, noise, (size_t)U * N * c.mel * 4, kind, stream));
    if (cmt && cmt != d_cmt.p) MI_HIP(hipMemcpyAsync(d_cmt.p, cmt, (size_t)U * N * c.cond_dim() * 4, kind, stream));
    if (cmtd && cmtd != d_cmtd.p) MI_HIP(hipMemcpyAsync(d_cmtd.p, cmtd, (size_t)U * N * c.cond_dim() * 4, kind, stream));
}

// cat[(2u+br), n, mel:] = (br == 0 ? cat_mel_text : cat_mel_text_drop)[u, n, :]   (engine dtype)
void F5::build_cat_cond(int U, int N) {
    const F5Cfg& c = cfg;
    const size_t es = dtype_size(dtype);
    const int cd = c.cond_dim(), ld = cat_ld();
    for (int u = 0; u < U; ++u)
        for (int br = 0; br < 2; ++br) {
            const float* src = (br == 0 ? d_cmt.as<float>() : d_cmtd.as<float>()) + (size_t)u * N * cd;
            char* dst = (char*)cat.p + ((size_t)(2 * u + br) * N * ld + c.mel) * es;
            launch_copy2d(src, cd, dst, ld, N, cd, dtype, stream);
        }
}

void F5::dit_eval(int U, int N, int k) {
    const F5Cfg& c = cfg;
    MI_REQUIRE(k >= 0 && k < c.nfe, "f5: time step out of range");
    hipStream_t s = stream;
    const int B = 2 * U, d = c.dim, ld = cat_ld(), ff = c.ff(), H = c.heads, D = c.dim_head;
    const long rows = (long)B * N;
    const float* modk = mod.as<float>() + (size_t)k * mod_ld;
    // ---- input embedding: proj(cat(x, cond)) ; + Mish(GConv(Mish(GConv(.)))) ---------------------------
    launch_cat_noise(d_noise.as<float>(), cat.p, U, N, c.mel, ld, dtype, s);
    gemm(dtype, cat.p, (long)N * ld, ld, ld, in_proj, h32.p, MI_F32, (long)N * d, d, B, N);
    const void* hin = h32.p;
    if (dtype != MI_F32) { launch_copy2d(h32.as<float>(), d, hT.p, d, rows, d, dtype, s); hin = hT.p; }
    {
        ConvGemm g;
        g.dtype = dtype; g.x = hin; g.w = gconv1.w.p; g.bias = gconv1.b.as<float>(); g.out = c1.p;
        g.B = B; g.G = c.pos_g; g.T_in = N; g.M = N; g.N = d / c.pos_g; g.Cin = d / c.pos_g; g.taps = c.pos_k; g.pad = c.pos_k / 2;
        g.x_bstride = (long)N * d; g.x_rstride = d; g.x_goff = d / c.pos_g; g.out_bstride = (long)N * d; g.out_rstride = d;
        g.act = ACT_MISH; g.sat = d_sat.as<int>(); g.gcp_w = gconv1.w3p.p;
        launch_conv_gemm(g, s);
        g.x = c1.p; g.w = gconv2.w.p; g.bias = gconv2.b.as<float>(); g.out = X.p; g.out_dtype = MI_F32; g.res = h32.p; g.gcp_w = gconv2.w3p.p;
        launch_conv_gemm(g, s);
    }
    // ---- transformer blocks ----------------------------------------------------------------------------------
    // One ConvGemm per linear layer, built ONCE: the same object answers "which kernel takes this" and is launched (ADVICE r3:
    // the panel-plane decision used to be taken on a hand-built copy and re-derived at launch).  Three forms:
    //   FOLD    the AdaLN fold: no row-norm launches; O / FF2 epilogues leave x o (1 + scale) + row statistics, QKV / FF1 finish
    //           the LayerNorm in theirs (gemm_epilogue.h).  fp32 engines: operands as panel planes; 16-bit engines: rows
    //   PLANES  fp32 engines without the fold: panel planes written by rownorm_x3p / attention / the FF1 epilogue (round 3)
    //   ROWS    everything else: row-norm launches, operands as rows (F5::gemm may still split them for the panel-plane kernel)
    const bool f32 = dtype == MI_F32;
    auto lin = [&](const Lin& L, int K, int odt, void* out) {
        ConvGemm g;
        g.dtype = dtype; g.out_dtype = odt; g.w = L.w.p; g.w3 = L.w3.p; g.bias = L.b.p ? L.b.as<float>() : nullptr; g.out = out;
        g.B = 1; g.T_in = (int)rows; g.M = (int)rows; g.N = L.n; g.Cin = K; g.taps = 1;
        g.x_bstride = rows * K; g.x_rstride = K; g.out_bstride = rows * L.n; g.out_rstride = L.n;
        g.sat = d_sat.as<int>();
        sk.attach(g);
        return g;
    };
    auto with_planes = [&](ConvGemm& g, const Lin& L, const void* xp) { g.xp = xp; g.w3p = L.w3p.p; g.np = np; };
    auto qkv_gemm = [&](const Block& bk) {
        ConvGemm g = lin(bk.qkv, d, -1, qb.p);
        g.out2 = kb.p; g.out3 = vb.p; g.rows_per_item = N;           // batch flattened into M
        g.epi = EPI_QKV_ROPE; g.rope_cos = rope_cos.as<float>(); g.rope_sin = rope_sin.as<float>(); g.rope_pack = rope_pack.p; g.heads = H; g.head_dim = D;
        g.v_ld = attention_v_ld(N, dtype);
        return g;
    };
    const int kvp_fmt = attention_kv_planes_format();       // K / V^T pre-split for the attention kernel: 2 fp16 planes or 3 bf16 planes
    // which form: decided on block 0's layers (every block has the same shapes and the same weight formats)
    bool planes = false, fold = false;
    {
        const Block& b0 = blocks[0];
        if (f32 && Ap.p && gemm_x3p_enabled() && b0.qkv.w3p.p && b0.o.w3p.p && b0.ff1.w3p.p && b0.ff2.w3p.p && attention_can_write_planes(N, B * H, dtype)) {
            ConvGemm gq = qkv_gemm(b0); gq.x = Ub.p; with_planes(gq, b0.qkv, Ap.p);
            ConvGemm go = lin(b0.o, d, MI_F32, X.p); go.x = Ob.p; go.res = X.p; go.gate = modk; with_planes(go, b0.o, Ap.p);
            ConvGemm g1 = lin(b0.ff1, d, dtype, Hff.p); g1.x = Ub.p; g1.act = ACT_GELU_TANH; with_planes(g1, b0.ff1, Ap.p);
            ConvGemm g2 = lin(b0.ff2, ff, MI_F32, X.p); g2.x = Hff.p; g2.res = X.p; g2.gate = modk; with_planes(g2, b0.ff2, Ap2.p);
            planes = gemm_x3p_would_run(gq) && gemm_x3p_would_run(go) && gemm_x3p_can_write_planes(g1) && gemm_x3p_would_run(g2);
        }
        if (fold_built && cfg.ln_fold != 0 && (planes || !f32) && (!f32 || ApN.p) && ln_stats.p && (f32 || ln_fin.p)) {
            ConvGemm gq = qkv_gemm(b0); gq.x = Ub.p;
            ConvGemm go = lin(b0.o, d, MI_F32, X.p); go.x = Ob.p; go.res = X.p; go.gate = modk;
            ConvGemm g1 = lin(b0.ff1, d, dtype, Hff.p); g1.x = Ub.p; g1.act = ACT_GELU_TANH;
            ConvGemm g2 = lin(b0.ff2, ff, MI_F32, X.p); g2.x = Hff.p; g2.res = X.p; g2.gate = modk;
            if (f32) { with_planes(gq, b0.qkv, ApN.p); with_planes(go, b0.o, Ap.p); with_planes(g1, b0.ff1, ApN.p); with_planes(g2, b0.ff2, Ap2.p); }
            fold = gemm_ln_fold_ok(gq) && gemm_ln_fold_ok(go) && gemm_ln_fold_ok(g1) && gemm_ln_fold_ok(g2);
        }
    }
    float* stats = ln_stats.as<float>();
    const float* lnk = fold ? ln_tab.as<float>() + (size_t)k * ln_ld : nullptr;
    const bool fin = fold && !f32;               // 16-bit engines: the statistics are finished by one tiny launch, not per column tile
    auto finalize = [&] { if (fin) launch_ln_finalize(stats, ln_fin.as<float>(), rows, d, 1e-6f, s); };
    auto consume = [&](ConvGemm& g) { g.ln_stats_in = fin ? ln_fin.as<float>() : stats; g.ln_final = fin ? 1 : 0; g.ln_dim = d; g.ln_eps = 1e-6f; };
    if (fold) {     // block 0's attention norm: the residual row comes from the position convolution, whose epilogue has no fold
        launch_ln_prologue(X.as<float>(), f32 ? ApN.p : Ub.p, dtype, np, stats, modk + d, rows, d, d_sat.as<int>(), s);
        finalize();
    }
    for (int i = 0; i < c.depth; ++i) {
        const Block& bk = blocks[i];
        const float* m = modk + (size_t)i * 6 * d;       // shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
        if (fold) {
            const float* lt = lnk + (size_t)i * ln_blk;   // W_qkv (1 + sc_a) | W_qkv sh_a + b | W_ff1 (1 + sc_m) | W_ff1 sh_m + b
            const bool kvp = f32 && attention_takes_kv_planes(N, B * H, dtype);
            {
                ConvGemm g = qkv_gemm(bk);
                g.x = Ub.p; g.bias = nullptr;
                if (f32) with_planes(g, bk.qkv, ApN.p);
                consume(g); g.ln_p = lt; g.ln_c = lt + 3 * d;
                if (kvp) { g.kv_planes = kvp_fmt; g.k_ld = g.v_ld = (long)((N + 63) / 64 * 64); }
                launch_conv_gemm(g, s);
            }
            launch_attention(qb.p, kb.p, vb.p, Ob.p, B * H, H, N, dtype, s, attn_ws.as<float>(), attn_ws_floats, attn_cnt.as<int>(), attn_cnt_n,
                             f32 ? Ap.p : nullptr, kvp ? kvp_fmt : 0, np, cfg.score_scale != 1.f ? cfg.score_scale : 0.f);
            {
                ConvGemm g = lin(bk.o, d, MI_F32, X.p);
                g.x = Ob.p; g.res = X.p; g.gate = m + 2 * d;
                if (f32) with_planes(g, bk.o, Ap.p);
                g.ln_scale = m + 4 * d; g.ln_out = f32 ? ApN.p : Ub.p; g.ln_out_np = np; g.ln_stats_out = stats;       // -> the FF1 norm
                launch_conv_gemm(g, s);
                finalize();
            }
            {
                ConvGemm g = lin(bk.ff1, d, dtype, Hff.p);
                g.x = Ub.p; g.bias = nullptr; g.act = ACT_GELU_TANH;
                if (f32) { with_planes(g, bk.ff1, ApN.p); g.out_planes = Ap2.p; }
                consume(g); g.ln_p = lt + 6 * d; g.ln_c = lt + 6 * d + ff;
                launch_conv_gemm(g, s);
            }
            {
                ConvGemm g = lin(bk.ff2, ff, MI_F32, X.p);
                g.x = Hff.p; g.res = X.p; g.gate = m + 5 * d;
                if (f32) with_planes(g, bk.ff2, Ap2.p);
                if (i + 1 < c.depth) {                   // -> the next block's attention norm (the last block's row goes to AdaLN-final below)
                    g.ln_scale = m + 6 * d + d; g.ln_out = f32 ? ApN.p : Ub.p; g.ln_out_np = np; g.ln_stats_out = stats;
                }
                launch_conv_gemm(g, s);
                if (i + 1 < c.depth) finalize();
            }
            continue;
        }
        if (planes) {
            const bool kvp = attention_takes_kv_planes(N, B * H, dtype);
            launch_rownorm_x3p(X.as<float>(), Ap.p, m + d, m, rows, d, 1e-6f, s, np, d_sat.as<int>());
            {
                ConvGemm g = qkv_gemm(bk);
                g.x = Ub.p; with_planes(g, bk.qkv, Ap.p);
                if (kvp) { g.kv_planes = kvp_fmt; g.k_ld = g.v_ld = (long)((N + 63) / 64 * 64); }
                MI_REQUIRE(gemm_x3p_would_run(g), "f5: the QKV layer left the panel-plane kernel between the decision and the launch");
                launch_conv_gemm(g, s);
            }
            launch_attention(qb.p, kb.p, vb.p, Ob.p, B * H, H, N, dtype, s, attn_ws.as<float>(), attn_ws_floats, attn_cnt.as<int>(), attn_cnt_n,
                             Ap.p, kvp ? kvp_fmt : 0, np, 0.f);
            {
                ConvGemm g = lin(bk.o, d, MI_F32, X.p);
                g.x = Ob.p; g.res = X.p; g.gate = m + 2 * d; with_planes(g, bk.o, Ap.p);
                MI_REQUIRE(gemm_x3p_would_run(g), "f5: the O projection left the panel-plane kernel between the decision and the launch");
                launch_conv_gemm(g, s);
            }
            launch_rownorm_x3p(X.as<float>(), Ap.p, m + 4 * d, m + 3 * d, rows, d, 1e-6f, s, np, d_sat.as<int>());
            {
                ConvGemm g = lin(bk.ff1, d, dtype, Hff.p);
                g.x = Ub.p; g.act = ACT_GELU_TANH; with_planes(g, bk.ff1, Ap.p); g.out_planes = Ap2.p;
                MI_REQUIRE(gemm_x3p_would_run(g), "f5: the FF1 layer left the panel-plane kernel between the decision and the launch");
                launch_conv_gemm(g, s);
            }
            {
                ConvGemm g = lin(bk.ff2, ff, MI_F32, X.p);
                g.x = Hff.p; g.res = X.p; g.gate = m + 5 * d; with_planes(g, bk.ff2, Ap2.p);
                MI_REQUIRE(gemm_x3p_would_run(g), "f5: the FF2 layer left the panel-plane kernel between the decision and the launch");
                launch_conv_gemm(g, s);
            }
            continue;
        }
        // ROWS
        launch_rownorm(NORM_LN_MOD, X.as<float>(), Ub.p, dtype, m + d, m, rows, d, 1e-6f, s);
        {
            ConvGemm g = qkv_gemm(bk);
            g.x = Ub.p; g.x_bstride = (long)B * N * d;
            launch_conv_gemm(g, s);
        }
        launch_attention(qb.p, kb.p, vb.p, Ob.p, B * H, H, N, dtype, s, attn_ws.as<float>(), attn_ws_floats, attn_cnt.as<int>(), attn_cnt_n,
                         nullptr, 0, np, cfg.score_scale != 1.f ? cfg.score_scale : 0.f);
        gemm(dtype, Ob.p, (long)N * d, d, d, bk.o, X.p, MI_F32, (long)N * d, d, B, N, ACT_NONE, X.p, m + 2 * d);
        launch_rownorm(NORM_LN_MOD, X.as<float>(), Ub.p, dtype, m + 4 * d, m + 3 * d, rows, d, 1e-6f, s);
        gemm(dtype, Ub.p, (long)N * d, d, d, bk.ff1, Hff.p, dtype, (long)N * ff, ff, B, N, ACT_GELU_TANH);
        gemm(dtype, Hff.p, (long)N * ff, ff, ff, bk.ff2, X.p, MI_F32, (long)N * d, d, B, N, ACT_NONE, X.p, m + 5 * d);
    }
    // ---- AdaLN-final (scale, shift order: modules.py:323) + proj_out ------------------------------------------
    const float* mf = modk + (size_t)c.depth * 6 * d;
    launch_rownorm(NORM_LN_MOD, X.as<float>(), Ub.p, dtype, mf, mf + d, rows, d, 1e-6f, s);
    if (proj_parts > 1) {
        ConvGemm g;                                              // K slices as groups: slice q multiplies columns [q d / parts, ...) of every row
        g.dtype = dtype; g.out_dtype = MI_F32; g.x = Ub.p; g.w = proj_out_k.w.p; g.bias = proj_out_k.b.as<float>(); g.out = pred.p;
        g.B = 1; g.G = proj_parts; g.T_in = (int)rows; g.M = (int)rows; g.N = c.mel; g.Cin = d / proj_parts; g.taps = 1;
        g.x_bstride = rows * d; g.x_rstride = d; g.x_goff = d / proj_parts;
        g.out_bstride = rows * proj_parts * c.mel; g.out_rstride = (long)proj_parts * c.mel;
        launch_conv_gemm(g, s);
    } else
        gemm(dtype, Ub.p, (long)N * d, d, d, proj_out, pred.p, MI_F32, (long)N * c.mel, c.mel, B, N);
}

const float* F5::pred_rows(int U, int N) {
    if (proj_parts <= 1) return pred.as<float>();
    launch_sum_parts(pred.as<float>(), pred_sum.as<float>(), (long)2 * U * N, cfg.mel, proj_parts, stream);
    return pred_sum.as<float>();
}

void F5::steps_eager(int U, int N, int k0, int nsteps) {
    for (int k = k0; k < k0 + nsteps; ++k) {
        dit_eval(U, N, k);
        launch_cfg_update(d_noise.as<float>(), pred.as<float>(), U, N, cfg.mel, cfg.cfg_strength, delta_t.as<float>(), k, stream, proj_parts);
    }
}

void F5::drop_graphs() {
    for (auto& kv : graphs)
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    graphs.clear();
}

// Called by the C-ABI when a call on this handle threw (capi.hip F5_CHECK).  The inter-workgroup hand-offs of the GEMM and
// attention kernels keep persistent flags / ticket counters in the workspace that every COMPLETED launch leaves at zero;
// after a faulted or aborted launch (or a capture that failed half way) they may be set, and the next launch would consume
// stale slabs or spin.  Nothing here throws.
void F5::recover() {
    (void)hipSetDevice(device);
    (void)hipStreamSynchronize(stream);
    (void)hipGetLastError();
    drop_graphs();
    sk.reset(stream);
    if (attn_cnt.p && attn_cnt_n > 0) {
        (void)hipMemsetAsync(attn_cnt.p, 0, (size_t)attn_cnt_n * 4, stream);
        (void)hipStreamSynchronize(stream);
    }
    // flags raised by the call that failed belong to it: a stale text-id flag would fail the next (valid) call, a stale range-watch
    // flag would re-run it and switch the engine to bf16x3 for good
    if (p_err.p) (void)hipMemset(p_err.p, 0, 4);
    if (d_sat.p) (void)hipMemset(d_sat.p, 0, 4);
}

// The reference drives 31 host round trips (F5-TTS-ONNX-Inference.py:291-304).  Here the whole loop is ~5000 kernel
// launches on one stream with every operand resident in HBM; from the second use of a shape on it is captured once
// into a hipGraph and replayed, which removes the per-launch host cost (launch-bound at batch 1).
void F5::steps(int U, int N, int k0, int nsteps) {
    MI_REQUIRE(k0 >= 0 && nsteps >= 0 && k0 + nsteps <= cfg.nfe - 1, "f5: step range exceeds the NFE grid");
    if (!use_graph || prof_mask() != 0 || nsteps < 2) { steps_eager(U, N, k0, nsteps); return; }
    if (graph_epoch != option_epoch()) { drop_graphs(); graph_epoch = option_epoch(); }
    GraphEntry& e = graphs[{U, N, k0, nsteps}];
    if (e.exec) { MI_HIP(hipGraphLaunch(e.exec, stream)); return; }
    if (e.uses++ == 0) { steps_eager(U, N, k0, nsteps); return; }      // first use: eager (also warms one-time allocations)
    hipGraph_t graph = nullptr;
    MI_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    try {
        steps_eager(U, N, k0, nsteps);
    } catch (...) {
        (void)hipStreamEndCapture(stream, &graph);
        if (graph) (void)hipGraphDestroy(graph);
        throw;
    }
    MI_HIP(hipStreamEndCapture(stream, &graph));
    hipError_t err = hipGraphInstantiate(&e.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (err != hipSuccess) { e.exec = nullptr; use_graph = false; steps_eager(U, N, k0, nsteps); return; }
    MI_HIP(hipGraphLaunch(e.exec, stream));
}

long F5::decode(const float* den, int U, int N, int R, float* out_f, int16_t* out_i) {
    const F5Cfg& c = cfg;
    const int F = N - R;
    MI_REQUIRE(R >= 0 && F >= 1, "f5_decode: max_duration leaves no generated frame");
    if (F == 1) return 0;        // one generated frame: the reference's graph C returns (N - R - 1) * hop = 0 samples (Export_F5.py:414)
    hipStream_t s = stream;
    const int vd = c.vd, vi = c.vi, nf = c.n_fft, nb = c.nb();
    const long rows = (long)U * F;
    float* h = v_h.as<float>(); float* z = v_z.as<float>(); float* z2 = v_z2.as<float>();
    {   // embed: Conv1d(mel -> vd, k7, pad 3) over the generated frames only (zero padded at the slice edges)
        ConvGemm g;
        g.dtype = MI_F32; g.x = den + (size_t)R * c.mel; g.w = v_embed.w.p; g.bias = v_embed.b.as<float>(); g.out = h;
        g.B = U; g.T_in = F; g.M = F; g.N = vd; g.Cin = c.mel; g.taps = 7; g.pad = 3;
        g.x_bstride = (long)N * c.mel; g.x_rstride = c.mel; g.out_bstride = (long)F * vd; g.out_rstride = vd;
        launch_conv_gemm(g, s);
    }
    launch_rownorm(NORM_L2, h, h, MI_F32, v_norm_w.as<float>(), v_norm_b.as<float>(), rows, vd, 0.f, s);
    for (auto& vbk : vblocks) {
        launch_dwconv7(h, z, vbk.dw_w.as<float>(), vbk.dw_b.as<float>(), U, F, vd, s);
        launch_rownorm(NORM_L2, z, z, MI_F32, vbk.n_w.as<float>(), vbk.n_b.as<float>(), rows, vd, 0.f, s);
        gemm(MI_F32, z, (long)F * vd, vd, vd, vbk.pw1, z2, MI_F32, (long)F * vi, vi, U, F, ACT_GELU_ERF);
        gemm(MI_F32, z2, (long)F * vi, vi, vi, vbk.pw2, h, MI_F32, (long)F * vd, vd, U, F, ACT_NONE, h);
    }
    launch_rownorm(NORM_L2, h, z, MI_F32, v_fnorm_w.as<float>(), v_fnorm_b.as<float>(), rows, vd, 0.f, s);
    gemm(MI_F32, z, (long)F * vd, vd, vd, v_head, v_s.p, MI_F32, (long)F * 2 * nb, 2 * nb, U, F);
    launch_vocos_head(v_s.as<float>(), v_c.as<float>(), rows, nb, istft.k, s);
    gemm(MI_F32, v_c.p, (long)F * istft.k, istft.k, istft.k, istft, v_fr.p, MI_F32, (long)F * nf, nf, U, F);
    launch_istft_ola(v_fr.as<float>(), wsi.as<float>(), U, F, nf, c.hop, out_f, out_i, s);
    return (long)(F - 1) * c.hop;
}

}  // namespace mi
