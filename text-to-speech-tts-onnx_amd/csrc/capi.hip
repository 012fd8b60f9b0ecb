// capi.hip — extern "C" boundary of libmi355tts.so (declarations: include/mi355tts.h).
#include "common.h"
#include "bigvgan.h"
#include "f5.h"
#include "gpt.h"
#include "cond.h"
#include <cstdlib>
#include <algorithm>
#include <mutex>
#include <exception>

using namespace mi;

// calls on one handle are serialised (one HIP stream + one workspace per handle); different handles run concurrently
struct mi_bigvgan { BigVGAN* impl; std::mutex mu; };
struct mi_f5 { F5* impl; std::mutex mu; };
struct mi_gpt { Gpt* impl; std::mutex mu; };
struct mi_cond { Cond* impl; std::mutex mu; };

// Every C-ABI entry runs under the SHARED side of the option lock; mi_set_option takes the exclusive side.  The dispatch options
// stay process-wide (tools, A/B runs, tests), but a change can never land in the middle of a call on another thread: it waits for
// the calls in flight, and calls that start afterwards see the whole new table (the header's promise that
// different handles may be driven from different threads holds with mi_set_option in the picture).
template <typename F> static int guard(F&& f, bool exclusive = false) {
    try {
        std::shared_lock<std::shared_mutex> rd(option_lock(), std::defer_lock);
        std::unique_lock<std::shared_mutex> wr(option_lock(), std::defer_lock);
        if (exclusive) wr.lock(); else rd.lock();
        f();
        return MI_OK;
    } catch (const mi::Error& e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        set_last_error("out of host memory");
        return MI_ENOMEM;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return MI_EINVAL;
    }
}

// Runs `body` (launches + a stream synchronisation).  fp16-pair engines: if a producer flagged an operand at the fp16 range limit
// (x3_split.h sat_publish) the results of the call are not fp32-equivalent — the engine switches itself to the exact three-plane
// bf16 split (permanently; mi_f5_info reports it) and the body runs again.  `body` must not have overwritten its inputs.
template <typename F> static void f5_run_checked(F5& e, F&& body) {
    body();
    if (e.dtype == MI_F32 && e.np == 2 && e.take_saturation()) {
        ++e.sat_events;
        e.set_arith(ARITH_BF16X3);
        ArithScope sc(e.arith);          // (the caller's scope still holds the pair override)
        body();
        (void)e.take_saturation();
    } else if (e.dtype == MI_F16 && e.fold_built && e.cfg.ln_fold != 0 && e.take_saturation()) {
        // f16 engine: the AdaLN fold's A operand (the unnormalised residual row o (1 + scale)) left the fp16 range — the row-norm
        // path's operand LN(x) (1 + scale) + shift is bounded by the norm: fold off for this engine, permanently, and run again
        ++e.sat_events;
        e.cfg.ln_fold = 0;
        e.drop_graphs();
        body();
        (void)e.take_saturation();
    }
}

extern "C" {

const char* mi_version(void) { return "mi355tts 0.1 (gfx950)"; }
const char* mi_last_error(void) { return last_error().c_str(); }

int mi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mi_init(int device) {
    return guard([&] {
        int n = 0;
        MI_HIP(hipGetDeviceCount(&n));
        MI_REQUIRE(device >= 0 && device < n, "mi_init: no such HIP device (libmi355tts needs an MI355X; there is no CPU fallback)");
        MI_HIP(hipSetDevice(device));
        hipDeviceProp_t prop;
        MI_HIP(hipGetDeviceProperties(&prop, device));
        MI_REQUIRE(std::strncmp(prop.gcnArchName, "gfx950", 6) == 0, "mi_init: device is not gfx950");
        MI_HIP(hipFree(nullptr));
    });
}

int64_t mi_bigvgan_param_count(const int32_t* cfg, int n_cfg) {
    int64_t n = -1;
    int rc = guard([&] { n = bigvgan_param_count(parse_bigvgan_cfg(cfg, n_cfg)); });
    return rc == MI_OK ? n : (int64_t)rc;
}

mi_bigvgan* mi_bigvgan_create(const int32_t* cfg, int n_cfg, const float* weights, int64_t n_weights, int dtype,
                              int device) {
    mi_bigvgan* h = nullptr;
    int rc = guard([&] {
        MI_REQUIRE(weights != nullptr, "mi_bigvgan_create: null weights");
        BigVGANCfg g = parse_bigvgan_cfg(cfg, n_cfg);
        BigVGAN* impl = new BigVGAN(g, weights, n_weights, dtype, device);
        h = new mi_bigvgan; h->impl = impl;
    });
    return rc == MI_OK ? h : nullptr;
}

// a device-resident blob whose re-layouts are host code (BigVGAN conv / GPT Conv1D transposes): read back once at load
static std::vector<float> read_back(const float* dev, int64_t n, int device) {
    MI_REQUIRE(n > 0, "create_mem: empty blob");
    MI_HIP(hipSetDevice(device));
    std::vector<float> h((size_t)n);
    MI_HIP(hipMemcpy(h.data(), dev, (size_t)n * 4, hipMemcpyDeviceToHost));
    return h;
}

mi_bigvgan* mi_bigvgan_create_mem(const int32_t* cfg, int n_cfg, const float* weights, int64_t n_weights, int dtype,
                                  int device, int mem) {
    if (mem == MI_HOST) return mi_bigvgan_create(cfg, n_cfg, weights, n_weights, dtype, device);
    mi_bigvgan* h = nullptr;
    int rc = guard([&] {
        MI_REQUIRE(weights != nullptr && mem == MI_DEVICE, "mi_bigvgan_create_mem: null weights / bad mem kind");
        BigVGANCfg g = parse_bigvgan_cfg(cfg, n_cfg);
        MI_REQUIRE(n_weights == bigvgan_param_count(g), "bigvgan: weight blob size does not match the config");
        std::vector<float> hw = read_back(weights, n_weights, device);
        BigVGAN* impl = new BigVGAN(g, hw.data(), n_weights, dtype, device);
        h = new mi_bigvgan; h->impl = impl;
    });
    return rc == MI_OK ? h : nullptr;
}

void mi_bigvgan_destroy(mi_bigvgan* h) {
    if (!h) return;
    delete h->impl;
    delete h;
}

int64_t mi_bigvgan_out_len(const mi_bigvgan* h, int frames) {
    if (!h || frames <= 0) return MI_EINVAL;
    return (int64_t)frames * h->impl->cfg.hop + 30;
}

int mi_bigvgan_forward(mi_bigvgan* h, const float* mel, int B, int frames, int16_t* out, int mem) {
    return guard([&] {
        MI_REQUIRE(h && h->impl, "mi_bigvgan_forward: null handle");
        std::lock_guard<std::mutex> lk_(h->mu);
        MI_REQUIRE(mem == MI_HOST || mem == MI_DEVICE, "mi_bigvgan_forward: bad mem kind");
        h->impl->run(mel, B, frames, nullptr, out, mem);
    });
}

int mi_bigvgan_forward_f32(mi_bigvgan* h, const float* mel, int B, int frames, float* out, int mem) {
    return guard([&] {
        MI_REQUIRE(h && h->impl, "mi_bigvgan_forward_f32: null handle");
        std::lock_guard<std::mutex> lk_(h->mu);
        MI_REQUIRE(mem == MI_HOST || mem == MI_DEVICE, "mi_bigvgan_forward_f32: bad mem kind");
        h->impl->run(mel, B, frames, out, nullptr, mem);
    });
}

int mi_bigvgan_forward_latent(mi_bigvgan* h, const float* latent, int T_codes, const float* conds, int64_t n_conds,
                              int16_t* out, float* out_f32, int mem) {
    return guard([&] {
        MI_REQUIRE(h && h->impl, "mi_bigvgan_forward_latent: null handle");
        std::lock_guard<std::mutex> lk_(h->mu);
        MI_REQUIRE(mem == MI_HOST || mem == MI_DEVICE, "mi_bigvgan_forward_latent: bad mem kind");
        h->impl->run_latent(latent, T_codes, conds, (long)n_conds, out_f32, out, mem);
    });
}

int mi_aa_activation1d(const float* x, int B, int C, int T, const float* alpha_log, const float* beta_log,
                       int logscale, int post, int dtype, float* y) {
    return guard([&] {
        MI_REQUIRE(x && y && alpha_log && beta_log && B > 0 && C > 0 && T > 0, "mi_aa_activation1d: bad arguments");
        unit_aa_activation1d(x, B, C, T, alpha_log, beta_log, logscale, post, dtype, y);
    });
}

int mi_aa_conv1d(const float* x, int B, int C, int T, const float* alpha_log, const float* beta_log, int logscale,
                 const float* w, const float* bias, int k, int dilation, const float* res, int dtype, float* y) {
    return guard([&] {
        MI_REQUIRE(x && y && w && bias && alpha_log && beta_log && B > 0 && C > 0 && T > 0 && k > 0 && (k & 1) && dilation > 0,
                   "mi_aa_conv1d: bad arguments");
        unit_aa_conv1d(x, B, C, T, alpha_log, beta_log, logscale, w, bias, k, dilation, res, dtype, 1, y);
    });
}

int mi_conv1d(const float* x, int B, int Cin, int T, const float* w, const float* bias, int Cout, int k, int dilation,
              int padding, int groups, int dtype, float* y) {
    return guard([&] {
        MI_REQUIRE(x && w && y && B > 0 && Cin > 0 && Cout > 0 && T > 0 && k > 0 && dilation > 0 && groups > 0 && padding >= 0,
                   "mi_conv1d: bad arguments");
        unit_conv1d(x, B, Cin, T, w, bias, Cout, k, dilation, padding, groups, dtype, y);
    });
}

int mi_conv_transpose1d(const float* x, int B, int Cin, int T, const float* w, const float* bias, int Cout, int k,
                        int stride, int padding, int dtype, float* y) {
    return guard([&] {
        MI_REQUIRE(x && w && y && B > 0 && Cin > 0 && Cout > 0 && T > 0, "mi_conv_transpose1d: bad arguments");
        unit_conv_transpose1d(x, B, Cin, T, w, bias, Cout, k, stride, padding, dtype, y);
    });
}

// ---------------------------------------------------------------------------------------------------
// F5-TTS
// ---------------------------------------------------------------------------------------------------
int64_t mi_f5_param_count(const int32_t* cfg_i, int n_i, const float* cfg_f, int n_f) {
    int64_t n = -1;
    int rc = guard([&] { n = f5_param_count(parse_f5_cfg(cfg_i, n_i, cfg_f, n_f)); });
    return rc == MI_OK ? n : (int64_t)rc;
}

mi_f5* mi_f5_create(const int32_t* cfg_i, int n_i, const float* cfg_f, int n_f, const float* weights, int64_t n_weights,
                    int dtype, int device) {
    mi_f5* h = nullptr;
    int rc = guard([&] {
        MI_REQUIRE(weights != nullptr, "mi_f5_create: null weights");
        F5Cfg c = parse_f5_cfg(cfg_i, n_i, cfg_f, n_f);
        F5* impl = new F5(c, weights, n_weights, dtype, device);
        h = new mi_f5; h->impl = impl;
    });
    return rc == MI_OK ? h : nullptr;
}

mi_f5* mi_f5_create_mem(const int32_t* cfg_i, int n_i, const float* cfg_f, int n_f, const float* weights, int64_t n_weights,
                        int dtype, int device, int mem) {
    mi_f5* h = nullptr;
    int rc = guard([&] {
        MI_REQUIRE(weights != nullptr && (mem == MI_HOST || mem == MI_DEVICE), "mi_f5_create_mem: null weights / bad mem kind");
        F5Cfg c = parse_f5_cfg(cfg_i, n_i, cfg_f, n_f);
        F5* impl = new F5(c, weights, n_weights, dtype, device, mem);
        h = new mi_f5; h->impl = impl;
    });
    return rc == MI_OK ? h : nullptr;
}

void mi_f5_destroy(mi_f5* h) {
    if (!h) return;
    delete h->impl;
    delete h;
}

// runs F5::recover() when the call leaves by an exception (while the handle lock is still held)
struct F5Recover {
    F5* e; int n;
    explicit F5Recover(F5* e_) : e(e_), n(std::uncaught_exceptions()) {}
    ~F5Recover() { if (std::uncaught_exceptions() > n) e->recover(); }
};
#define F5_CHECK(h, mem, name)                                                     \
    MI_REQUIRE((h) && (h)->impl, name ": null handle");                            \
    std::lock_guard<std::mutex> lk_((h)->mu);                                      \
    F5Recover rec_((h)->impl);                                                     \
    ArithScope arith_((h)->impl->arith);                                           \
    MI_REQUIRE((mem) == MI_HOST || (mem) == MI_DEVICE, name ": bad mem kind")

static void copy_out(void* dst, const void* src, size_t bytes, int mem, hipStream_t s) {
    if (!dst) return;
    MI_HIP(hipMemcpyAsync(dst, src, bytes, mem == MI_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, s));
}

int mi_f5_tables(mi_f5* h, float* time_expand, float* delta_t) {
    return guard([&] {
        MI_REQUIRE(h && h->impl, "mi_f5_tables: null handle");
        std::lock_guard<std::mutex> lk_(h->mu);
        F5& e = *h->impl;
        if (time_expand) std::memcpy(time_expand, e.h_time_expand.data(), e.h_time_expand.size() * 4);
        if (delta_t) std::memcpy(delta_t, e.h_delta.data(), e.h_delta.size() * 4);
    });
}

int64_t mi_f5_info(mi_f5* h, const char* key) {
    int64_t v = -1;
    int rc = guard([&] {
        MI_REQUIRE(h && h->impl && key, "mi_f5_info: null handle / key");
        std::lock_guard<std::mutex> lk_(h->mu);
        F5& e = *h->impl;
        ArithScope as(e.arith);
        const std::string k(key);
        if (k == "f32_arithmetic") v = e.dtype != MI_F32 ? -1 : !gemm_x3_enabled() ? ARITH_NATIVE : (gemm_x3p_enabled() ? e.np : ARITH_BF16X3);
        else if (k == "saturation_events") v = e.sat_events;
        else if (k == "adaln_fold") v = (e.fold_built && e.cfg.ln_fold != 0) ? 1 : 0;
        else MI_REQUIRE(false, "mi_f5_info: unknown key");
    });
    return rc == MI_OK ? v : (int64_t)rc;
}

int mi_f5_preprocess(mi_f5* h, const int16_t* audio, int64_t L, const int32_t* text_ids, int64_t T, int64_t max_duration,
                     const float* noise_in, uint64_t seed, float* noise, float* rope_cos, float* rope_sin,
                     float* cat_mel_text, float* cat_mel_text_drop, int64_t* ref_signal_len, int mem) {
    return guard([&] {
        F5_CHECK(h, mem, "mi_f5_preprocess");
        F5& e = *h->impl;
        MI_REQUIRE(max_duration > 0 && max_duration <= e.cfg.max_len && T >= 0 && T < (1 << 20) && L > 0 && L < (1L << 31),
                   "mi_f5_preprocess: sizes");
        const int N = (int)max_duration;
        const int R = e.preprocess(1, audio, L, text_ids, (int)T, N, noise_in, seed, mem);
        hipStream_t s = e.stream;
        const size_t cd = e.cfg.cond_dim(), D = e.cfg.dim_head;
        copy_out(noise, e.d_noise.p, (size_t)N * e.cfg.mel * 4, mem, s);
        copy_out(rope_cos, e.rope_cos.p, (size_t)N * D * 4, mem, s);
        copy_out(rope_sin, e.rope_sin.p, (size_t)N * D * 4, mem, s);
        copy_out(cat_mel_text, e.d_cmt.p, (size_t)N * cd * 4, mem, s);
        copy_out(cat_mel_text_drop, e.d_cmtd.p, (size_t)N * cd * 4, mem, s);
        MI_HIP(hipStreamSynchronize(s));
        e.finish_call();
        if (ref_signal_len) *ref_signal_len = R;
    });
}

int mi_f5_transformer_step(mi_f5* h, float* noise, const float* cmt, const float* cmtd, int U, int64_t N,
                           int32_t* time_step, int fuse, int mem) {
    return guard([&] {
        F5_CHECK(h, mem, "mi_f5_transformer_step");
        MI_REQUIRE(noise && cmt && cmtd && time_step && fuse >= 1 && U >= 1 && N > 0, "mi_f5_transformer_step: bad arguments");
        F5& e = *h->impl;
        f5_run_checked(e, [&] {
            e.load_cond(noise, cmt, cmtd, U, (int)N, mem);
            e.build_cat_cond(U, (int)N);
            e.steps(U, (int)N, *time_step, fuse);
            MI_HIP(hipStreamSynchronize(e.stream));
        });
        copy_out(noise, e.d_noise.p, (size_t)U * N * e.cfg.mel * 4, mem, e.stream);
        MI_HIP(hipStreamSynchronize(e.stream));
        e.finish_call();
        *time_step += fuse;
    });
}

int mi_f5_sample(mi_f5* h, float* noise, const float* cmt, const float* cmtd, int U, int64_t N, int k0, int n_steps, int mem) {
    return guard([&] {
        F5_CHECK(h, mem, "mi_f5_sample");
        MI_REQUIRE(noise && cmt && cmtd && U >= 1 && N > 0, "mi_f5_sample: bad arguments");
        F5& e = *h->impl;
        f5_run_checked(e, [&] {
            e.load_cond(noise, cmt, cmtd, U, (int)N, mem);
            e.build_cat_cond(U, (int)N);
            e.steps(U, (int)N, k0, n_steps);
            MI_HIP(hipStreamSynchronize(e.stream));
        });
        copy_out(noise, e.d_noise.p, (size_t)U * N * e.cfg.mel * 4, mem, e.stream);
        MI_HIP(hipStreamSynchronize(e.stream));
        e.finish_call();
    });
}

int mi_f5_dit_eval(mi_f5* h, const float* noise, const float* cmt, const float* cmtd, int U, int64_t N, int k, float* pred, int mem) {
    return guard([&] {
        F5_CHECK(h, mem, "mi_f5_dit_eval");
        MI_REQUIRE(noise && cmt && cmtd && pred && U >= 1 && N > 0, "mi_f5_dit_eval: bad arguments");
        F5& e = *h->impl;
        f5_run_checked(e, [&] {
            e.load_cond(noise, cmt, cmtd, U, (int)N, mem);
            e.build_cat_cond(U, (int)N);
            e.dit_eval(U, (int)N, k);
            MI_HIP(hipStreamSynchronize(e.stream));
        });
        copy_out(pred, e.pred_rows(U, (int)N), (size_t)2 * U * N * e.cfg.mel * 4, mem, e.stream);
        MI_HIP(hipStreamSynchronize(e.stream));
        e.finish_call();
    });
}

int mi_f5_decode(mi_f5* h, const float* denoised, int U, int64_t N, int64_t ref_signal_len, int16_t* out, float* out_f32,
                 int64_t* out_len, int mem) {
    return guard([&] {
        F5_CHECK(h, mem, "mi_f5_decode");
        MI_REQUIRE(denoised && (out || out_f32) && U >= 1 && N > 0 && ref_signal_len >= 0 && ref_signal_len < N,
                   "mi_f5_decode: bad arguments");
        F5& e = *h->impl;
        e.load_cond(denoised, nullptr, nullptr, U, (int)N, mem);
        const long len = e.decode(e.d_noise.as<float>(), U, (int)N, (int)ref_signal_len, out_f32 ? e.v_outf.as<float>() : nullptr,
                                  out ? e.v_outi.as<int16_t>() : nullptr);
        copy_out(out, e.v_outi.p, (size_t)U * len * 2, mem, e.stream);
        copy_out(out_f32, e.v_outf.p, (size_t)U * len * 4, mem, e.stream);
        MI_HIP(hipStreamSynchronize(e.stream));
        e.finish_call();
        if (out_len) *out_len = len;
    });
}

int mi_f5_synthesize(mi_f5* h, int U, const int16_t* audio, int64_t L, const int32_t* text_ids, int64_t T,
                     int64_t max_duration, const float* noise_in, uint64_t seed, int16_t* out, int64_t* out_len, int mem) {
    return guard([&] {
        F5_CHECK(h, mem, "mi_f5_synthesize");
        MI_REQUIRE(audio && text_ids && out && U >= 1 && max_duration > 0, "mi_f5_synthesize: bad arguments");
        F5& e = *h->impl;
        const int N = (int)max_duration;
        long len = 0;
        f5_run_checked(e, [&] {
            const int R = e.preprocess(U, audio, L, text_ids, (int)T, N, noise_in, seed, mem);
            e.build_cat_cond(U, N);
            e.steps(U, N, 0, e.cfg.nfe - 1);
            len = e.decode(e.d_noise.as<float>(), U, N, R, nullptr, e.v_outi.as<int16_t>());
            copy_out(out, e.v_outi.p, (size_t)U * len * 2, mem, e.stream);
            MI_HIP(hipStreamSynchronize(e.stream));
        });
        e.finish_call();
        if (out_len) *out_len = len;
    });
}

// A -> loop, and the generated frames handed on as a mel for a vocoder engine: (U, 100, N - R) fp32, channels first — the
// `mel_features` layout of mi_bigvgan_forward (BigVGAN-v2 24khz_100band_256x shares F5's 100 bands / hop 256 / 24 kHz; the
// reference's bigvgan-type front end is modules.py:30-72).  *n_frames receives N - R.
int mi_f5_synthesize_mel(mi_f5* h, int U, const int16_t* audio, int64_t L, const int32_t* text_ids, int64_t T,
                         int64_t max_duration, const float* noise_in, uint64_t seed, float* mel_out, int64_t* n_frames, int mem) {
    return guard([&] {
        F5_CHECK(h, mem, "mi_f5_synthesize_mel");
        MI_REQUIRE(audio && text_ids && mel_out && U >= 1 && max_duration > 0, "mi_f5_synthesize_mel: bad arguments");
        F5& e = *h->impl;
        const int N = (int)max_duration;
        int F = 0;
        f5_run_checked(e, [&] {
            const int R = e.preprocess(U, audio, L, text_ids, (int)T, N, noise_in, seed, mem);
            const int mel = e.cfg.mel;
            F = N - R;
            MI_REQUIRE(F >= 1, "mi_f5_synthesize_mel: max_duration leaves no generated frames");
            e.build_cat_cond(U, N);
            e.steps(U, N, 0, e.cfg.nfe - 1);
            float* dst = mel_out;
            if (mem == MI_HOST) { e.v_outf.ensure((size_t)U * mel * F * 4); dst = e.v_outf.as<float>(); }
            for (int u = 0; u < U; ++u)
                launch_nlc_to_ncl(e.d_noise.as<float>() + ((size_t)u * N + R) * mel, dst + (size_t)u * mel * F, 1, mel, F, MI_F32, e.stream);
            if (mem == MI_HOST) copy_out(mel_out, dst, (size_t)U * mel * F * 4, mem, e.stream);
            MI_HIP(hipStreamSynchronize(e.stream));
        });
        e.finish_call();
        if (n_frames) *n_frames = F;
    });
}

int mi_f5_stft(mi_f5* h, const int16_t* audio, int64_t L, float* spec, int mem) {
    return guard([&] {
        F5_CHECK(h, mem, "mi_f5_stft");
        F5& e = *h->impl;
        MI_REQUIRE(audio && spec && L >= e.cfg.n_fft / 2 + 1 && L < (1L << 31), "mi_f5_stft: bad arguments");
        MI_HIP(hipSetDevice(e.device));
        const int nf = e.cfg.n_fft, nb = e.cfg.nb(), R = (int)(L / e.cfg.hop) + 1;
        e.p_audio.ensure((size_t)L * 2); e.p_pad.ensure((size_t)(L + nf) * 4 + 64); e.p_spec.ensure((size_t)R * 2 * nb * 4);
        const int16_t* da = audio;
        if (mem == MI_HOST) {
            MI_HIP(hipMemcpyAsync(e.p_audio.p, audio, (size_t)L * 2, hipMemcpyHostToDevice, e.stream));
            da = e.p_audio.as<int16_t>();
        }
        e.stft(da, 1, L);
        copy_out(spec, e.p_spec.p, (size_t)R * 2 * nb * 4, mem, e.stream);
        MI_HIP(hipStreamSynchronize(e.stream));
    });
}

// ---------------------------------------------------------------------------------------------------------------
// IndexTTS GPT
// ---------------------------------------------------------------------------------------------------------------
int64_t mi_gpt_param_count(const int32_t* cfg, int n_cfg) {
    int64_t n = -1;
    int rc = guard([&] { n = gpt_param_count(parse_gpt_cfg(cfg, n_cfg)); });
    return rc == MI_OK ? n : (int64_t)rc;
}

mi_gpt* mi_gpt_create(const int32_t* cfg, int n_cfg, const float* weights, int64_t n_weights, int dtype, int device) {
    mi_gpt* h = nullptr;
    int rc = guard([&] {
        MI_REQUIRE(weights != nullptr, "mi_gpt_create: null weights");
        GptCfg c = parse_gpt_cfg(cfg, n_cfg);
        Gpt* impl = new Gpt(c, weights, n_weights, dtype, device);
        h = new mi_gpt; h->impl = impl;
    });
    return rc == MI_OK ? h : nullptr;
}

mi_gpt* mi_gpt_create_mem(const int32_t* cfg, int n_cfg, const float* weights, int64_t n_weights, int dtype, int device,
                          int mem) {
    if (mem == MI_HOST) return mi_gpt_create(cfg, n_cfg, weights, n_weights, dtype, device);
    mi_gpt* h = nullptr;
    int rc = guard([&] {
        MI_REQUIRE(weights != nullptr && mem == MI_DEVICE, "mi_gpt_create_mem: null weights / bad mem kind");
        GptCfg c = parse_gpt_cfg(cfg, n_cfg);
        MI_REQUIRE(n_weights == gpt_param_count(c), "gpt: weight blob size does not match the config");
        std::vector<float> hw = read_back(weights, n_weights, device);
        Gpt* impl = new Gpt(c, hw.data(), n_weights, dtype, device);
        h = new mi_gpt; h->impl = impl;
    });
    return rc == MI_OK ? h : nullptr;
}

void mi_gpt_destroy(mi_gpt* h) {
    if (!h) return;
    delete h->impl;
    delete h;
}

#define GPT_CHECK(h, mem, name)                                                    \
    MI_REQUIRE((h) && (h)->impl, name ": null handle");                            \
    std::lock_guard<std::mutex> lk_((h)->mu);                                      \
    MI_REQUIRE((mem) == MI_HOST || (mem) == MI_DEVICE, name ": bad mem kind");     \
    MI_HIP(hipSetDevice((h)->impl->device))

// input staged to the device when it lives on the host
static const void* stage_in(DevBuf& buf, const void* src, size_t bytes, int mem, hipStream_t s) {
    if (mem == MI_DEVICE) return src;
    buf.ensure(bytes);
    MI_HIP(hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, s));
    return buf.p;
}

int mi_gpt_text_embed(mi_gpt* h, const int32_t* text_ids, int n, float* out, int mem) {
    return guard([&] {
        GPT_CHECK(h, mem, "mi_gpt_text_embed");
        Gpt& e = *h->impl;
        MI_REQUIRE(out && n >= 0 && (n == 0 || text_ids), "mi_gpt_text_embed: null argument");
        MI_REQUIRE(n + 2 <= e.cfg.max_text_pos, "mi_gpt_text_embed: text is longer than the text position table");
        if (mem == MI_HOST)
            for (int i = 0; i < n; ++i)
                MI_REQUIRE(text_ids[i] >= 0 && text_ids[i] < e.cfg.text_tokens, "mi_gpt_text_embed: text id out of range");
        const size_t ob = (size_t)(n + 2) * e.cfg.hidden * 4;
        const int32_t* ids = n ? (const int32_t*)stage_in(e.io_a, text_ids, (size_t)n * 4, mem, e.stream) : nullptr;
        float* o = out;
        if (mem == MI_HOST) { e.io_b.ensure(ob); o = e.io_b.as<float>(); }
        e.text_embed(ids, n, o);
        if (mem == MI_HOST) copy_out(out, o, ob, mem, e.stream);
        MI_HIP(hipStreamSynchronize(e.stream));
    });
}

int mi_gpt_mel_embed(mi_gpt* h, int32_t gpt_id, int64_t gen_len, float* out, int mem) {
    return guard([&] {
        GPT_CHECK(h, mem, "mi_gpt_mel_embed");
        Gpt& e = *h->impl;
        MI_REQUIRE(out != nullptr, "mi_gpt_mel_embed: null output");
        const size_t ob = (size_t)e.cfg.hidden * 4;
        float* o = out;
        if (mem == MI_HOST) { e.io_b.ensure(ob); o = e.io_b.as<float>(); }
        e.mel_embed(gpt_id, gen_len, o);
        if (mem == MI_HOST) copy_out(out, o, ob, mem, e.stream);
        MI_HIP(hipStreamSynchronize(e.stream));
    });
}

int mi_gpt_reset(mi_gpt* h) {
    return guard([&] {
        MI_REQUIRE(h && h->impl, "mi_gpt_reset: null handle");
        std::lock_guard<std::mutex> lk_(h->mu);
        MI_HIP(hipSetDevice(h->impl->device));
        h->impl->reset();
    });
}

int64_t mi_gpt_history_len(mi_gpt* h) {
    if (!h || !h->impl) return MI_EINVAL;
    return h->impl->history;
}

int mi_gpt_step(mi_gpt* h, const float* hidden_state, int ids_len, const float* repeat_penality, int attention_mask,
                float* last_hidden_state, int32_t* max_logit_id, float* logits, int mem) {
    return guard([&] {
        GPT_CHECK(h, mem, "mi_gpt_step");
        Gpt& e = *h->impl;
        const GptCfg& c = e.cfg;
        MI_REQUIRE(hidden_state && ids_len >= 1, "mi_gpt_step: hidden_state must hold at least one row");
        MI_REQUIRE(e.history + ids_len <= c.max_seq, "mi_gpt_step: history_len + ids_len exceeds the KV cache (max_seq)");
        hipStream_t s = e.stream;
        const hipMemcpyKind in = mem == MI_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
        MI_HIP(hipMemcpyAsync(e.X.p, hidden_state, (size_t)ids_len * c.hidden * 4, in, s));
        if (repeat_penality) MI_HIP(hipMemcpyAsync(e.pen.p, repeat_penality, (size_t)c.mel_codes * 4, in, s));
        else {
            std::vector<float> ones(c.mel_codes, 1.f);
            MI_HIP(hipMemcpyAsync(e.pen.p, ones.data(), ones.size() * 4, hipMemcpyHostToDevice, s));
            MI_HIP(hipStreamSynchronize(s));
        }
        std::vector<int32_t> w(GS_WORDS, 0);
        w[GS_HIST] = e.history;
        e.set_state(w);
        e.forward_rows(ids_len, attention_mask ? 1 : 0);
        copy_out(last_hidden_state, e.last.p, (size_t)c.hidden * 4, mem, s);
        copy_out(logits, e.logits.p, (size_t)c.mel_codes * 4, mem, s);      // un-penalised; see below
        copy_out(max_logit_id, e.state.as<int32_t>() + GS_TOKEN, 4, mem, s);
        w = e.get_state();                                                  // synchronises; history := hist + ids_len
    });
}

int mi_gpt_kv_read(mi_gpt* h, int layer, float* keys, float* values, int mem) {
    return guard([&] {
        GPT_CHECK(h, mem, "mi_gpt_kv_read");
        Gpt& e = *h->impl;
        const size_t n = (size_t)e.cfg.hidden * e.history * 4;
        if (n == 0) return;
        float *k = keys, *v = values;
        if (mem == MI_HOST) {
            e.io_a.ensure(n); e.io_b.ensure(n);
            k = keys ? e.io_a.as<float>() : nullptr; v = values ? e.io_b.as<float>() : nullptr;
        }
        e.kv_read(layer, k, v);
        if (mem == MI_HOST) { copy_out(keys, k, n, mem, e.stream); copy_out(values, v, n, mem, e.stream); }
        MI_HIP(hipStreamSynchronize(e.stream));
    });
}

int mi_gpt_kv_write(mi_gpt* h, int layer, const float* keys, const float* values, int hist, int mem) {
    return guard([&] {
        GPT_CHECK(h, mem, "mi_gpt_kv_write");
        Gpt& e = *h->impl;
        MI_REQUIRE(hist >= 0 && hist < e.cfg.max_seq, "mi_gpt_kv_write: history does not fit the KV cache (max_seq)");
        MI_REQUIRE(hist == 0 || (keys && values), "mi_gpt_kv_write: null keys/values");
        const size_t n = (size_t)e.cfg.hidden * hist * 4;
        const float* k = hist ? (const float*)stage_in(e.io_a, keys, n, mem, e.stream) : nullptr;
        const float* v = hist ? (const float*)stage_in(e.io_b, values, n, mem, e.stream) : nullptr;
        e.kv_write(layer, k, v, hist);
    });
}

int mi_gpt_generate(mi_gpt* h, const float* prompt, int P, int max_new, const int32_t* stop_ids, int n_stop,
                    float repeat_value, int penalty_range, float* repeat_penality, int32_t* tokens, float* hidden,
                    int32_t* n_out, int mem) {
    return guard([&] {
        GPT_CHECK(h, mem, "mi_gpt_generate");
        Gpt& e = *h->impl;
        const GptCfg& c = e.cfg;
        MI_REQUIRE(prompt && P >= 1 && n_out, "mi_gpt_generate: null argument");
        MI_REQUIRE(n_stop >= 0 && n_stop <= GS_WORDS - GS_STOP0 && (n_stop == 0 || stop_ids), "mi_gpt_generate: at most 6 stop ids");
        *n_out = 0;
        if (max_new <= 0) return;                                        // `while num_decode < generate_limit` never runs
        MI_REQUIRE(P + max_new - 1 <= c.max_seq, "mi_gpt_generate: prompt + max_new exceeds the KV cache (max_seq)");
        MI_REQUIRE(max_new <= c.max_mel_pos, "mi_gpt_generate: max_new exceeds the mel position table");
        hipStream_t s = e.stream;
        const hipMemcpyKind in = mem == MI_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
        MI_HIP(hipMemcpyAsync(e.X.p, prompt, (size_t)P * c.hidden * 4, in, s));
        if (repeat_penality) MI_HIP(hipMemcpyAsync(e.pen.p, repeat_penality, (size_t)c.mel_codes * 4, in, s));
        else {
            std::vector<float> ones(c.mel_codes, 1.f);
            MI_HIP(hipMemcpyAsync(e.pen.p, ones.data(), ones.size() * 4, hipMemcpyHostToDevice, s));
            MI_HIP(hipStreamSynchronize(s));
        }
        e.set_rep_value(repeat_value);
        std::vector<int32_t> w(GS_WORDS, 0);
        w[GS_GEN_LEN] = 0; w[GS_NSTOP] = n_stop; w[GS_RANGE] = penalty_range; w[GS_UPDATE_PEN] = 1;
        if (mem == MI_HOST) for (int i = 0; i < n_stop; ++i) w[GS_STOP0 + i] = stop_ids[i];
        else MI_HIP(hipMemcpy(&w[GS_STOP0], stop_ids, (size_t)n_stop * 4, hipMemcpyDeviceToHost));
        e.set_state(w);
        e.forward_rows(P, 1);                                            // prompt pass: first token
        int left = max_new - 1;
        // the stop test needs the host: look at the state every `chunk` tokens (steps after a stop are no-ops)
        const int chunk = 16;
        for (;;) {
            w = e.get_state();
            if (w[GS_DONE] || left <= 0) break;
            const int n = left < chunk ? left : chunk;
            e.decode_steps(n);
            left -= n;
        }
        const int n = w[GS_NDEC];
        *n_out = n;
        copy_out(tokens, e.toks.p, (size_t)n * 4, mem, s);
        copy_out(hidden, e.hid.p, (size_t)n * c.hidden * 4, mem, s);
        copy_out(repeat_penality, e.pen.p, (size_t)c.mel_codes * 4, mem, s);
        MI_HIP(hipStreamSynchronize(s));
    });
}

int mi_gpt_generate_batch(mi_gpt* h, int nb, const float* prompts, const int32_t* prompt_rows, const int32_t* max_new,
                          const int32_t* stop_ids, int n_stop, float repeat_value, int penalty_range,
                          float* repeat_penality, int32_t* tokens, float* hidden, int cap, int32_t* n_out, int mem) {
    return guard([&] {
        GPT_CHECK(h, mem, "mi_gpt_generate_batch");
        Gpt& e = *h->impl;
        const GptCfg& c = e.cfg;
        MI_REQUIRE(prompts && prompt_rows && max_new && n_out && cap >= 1, "mi_gpt_generate_batch: null argument");
        MI_REQUIRE(nb >= 1 && nb <= c.max_batch, "mi_gpt_generate_batch: batch exceeds the handle's max_batch");
        MI_REQUIRE(n_stop >= 0 && n_stop <= GS_WORDS - GS_STOP0 && (n_stop == 0 || stop_ids), "mi_gpt_generate_batch: at most 6 stop ids");
        hipStream_t s = e.stream;
        const hipMemcpyKind in = mem == MI_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
        std::vector<int32_t> stops(n_stop);
        if (n_stop) {
            if (mem == MI_HOST) std::copy(stop_ids, stop_ids + n_stop, stops.begin());
            else MI_HIP(hipMemcpy(stops.data(), stop_ids, (size_t)n_stop * 4, hipMemcpyDeviceToHost));
        }
        size_t row0 = 0;
        for (int b = 0; b < nb; ++b) {
            MI_REQUIRE(prompt_rows[b] >= 1 && max_new[b] >= 0 && max_new[b] <= cap, "mi_gpt_generate_batch: prompt_rows / max_new");
            MI_REQUIRE(max_new[b] == 0 || prompt_rows[b] + max_new[b] - 1 <= c.max_seq, "mi_gpt_generate_batch: prompt + max_new exceeds the KV cache (max_seq)");
            MI_REQUIRE(max_new[b] <= c.max_mel_pos, "mi_gpt_generate_batch: max_new exceeds the mel position table");
        }
        if (repeat_penality) MI_HIP(hipMemcpyAsync(e.pen.p, repeat_penality, (size_t)nb * c.mel_codes * 4, in, s));
        else {
            std::vector<float> ones((size_t)nb * c.mel_codes, 1.f);
            MI_HIP(hipMemcpyAsync(e.pen.p, ones.data(), ones.size() * 4, hipMemcpyHostToDevice, s));
            MI_HIP(hipStreamSynchronize(s));
        }
        e.set_rep_value(repeat_value);
        bool any = false;
        for (int b = 0; b < nb; ++b) {                               // prompt passes, one sentence at a time
            std::vector<int32_t> w(GS_WORDS, 0);
            w[GS_NSTOP] = n_stop; w[GS_RANGE] = penalty_range; w[GS_UPDATE_PEN] = 1; w[GS_LIMIT] = max_new[b];
            for (int i = 0; i < n_stop; ++i) w[GS_STOP0 + i] = stops[i];
            if (max_new[b] == 0) w[GS_DONE] = 1;
            e.set_state(w, b);
            if (max_new[b] > 0) {
                MI_HIP(hipMemcpyAsync(e.X.p, prompts + row0 * c.hidden, (size_t)prompt_rows[b] * c.hidden * 4, in, s));
                e.forward_rows(prompt_rows[b], 1, b);
                any = true;
            }
            row0 += prompt_rows[b];
        }
        std::vector<int32_t> all((size_t)nb * GS_WORDS);
        auto read_states = [&] {
            MI_HIP(hipMemcpyAsync(all.data(), e.state.p, all.size() * 4, hipMemcpyDeviceToHost, s));
            MI_HIP(hipStreamSynchronize(s));
            bool done = true;
            for (int b = 0; b < nb; ++b) done &= all[(size_t)b * GS_WORDS + GS_DONE] != 0;
            return done;
        };
        int guard_steps = 0;
        while (any && !read_states()) {
            e.decode_batch_steps(nb, 16);
            guard_steps += 16;
            MI_REQUIRE(guard_steps <= c.max_seq + 32, "mi_gpt_generate_batch: decode loop did not terminate");
        }
        if (!any) read_states();
        e.history = all[GS_HIST];
        for (int b = 0; b < nb; ++b) {
            const int n = all[(size_t)b * GS_WORDS + GS_NDEC];
            n_out[b] = n;
            copy_out(tokens ? tokens + (size_t)b * cap : nullptr, e.toks.as<int32_t>() + (size_t)b * c.max_seq, (size_t)n * 4, mem, s);
            copy_out(hidden ? hidden + (size_t)b * cap * c.hidden : nullptr, e.hid.as<float>() + (size_t)b * c.max_seq * c.hidden,
                     (size_t)n * c.hidden * 4, mem, s);
        }
        copy_out(repeat_penality, e.pen.p, (size_t)nb * c.mel_codes * 4, mem, s);
        MI_HIP(hipStreamSynchronize(s));
    });
}

int mi_bench_conv_gemm(int dtype, int B, int T, int Cin, int N, int taps, int dil, int with_res, int iters, double* ms) {
    return guard([&] {
        MI_REQUIRE(ms && B > 0 && T > 0 && Cin > 0 && N > 0 && taps > 0 && iters > 0, "mi_bench_conv_gemm: bad arguments");
        hipStream_t s;
        MI_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        const size_t es = dtype_size(dtype);
        DevBuf x, w, o, r, bias;
        const size_t nx = (size_t)B * T * Cin, nw = (size_t)N * taps * Cin, no = (size_t)B * T * N;
        std::vector<float> hx(nx), hw(nw), hb(N, 0.01f);
        uint32_t st = 12345;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
        for (auto& v : hx) v = rnd();
        for (auto& v : hw) v = rnd() * 0.1f;
        if (const char* z = std::getenv("MI355TTS_BENCH_ZERO")) if (z[0] == '1') { std::fill(hx.begin(), hx.end(), 0.f); std::fill(hw.begin(), hw.end(), 0.f); }   // data-dependent power / clock check
        // MI355TTS_BENCH_WSETS=n: cycle through n copies of the weights (n * bytes > 256 MiB MALL => every launch
        // streams cold weights from HBM, like consecutive layers of a real model)
        int nsets = 1;
        if (const char* e = std::getenv("MI355TTS_BENCH_WSETS")) nsets = std::max(1, std::atoi(e));
        upload_as(x, hx.data(), nx, dtype, s); upload_as(w, hw.data(), nw, dtype, s); upload_f32(bias, hb.data(), N, s);
        DevBuf wsets;
        if (nsets > 1) {
            wsets.ensure((size_t)nsets * nw * es);
            for (int i = 0; i < nsets; ++i)
                MI_HIP(hipMemcpyAsync((char*)wsets.p + (size_t)i * nw * es, w.p, nw * es, hipMemcpyDeviceToDevice, s));
        }
        o.ensure(no * es); r.ensure(no * es);
        MI_HIP(hipMemsetAsync(r.p, 0, no * es, s));
        ConvGemm p;
        p.dtype = dtype; p.x = x.p; p.w = w.p; p.bias = bias.as<float>(); p.out = o.p; p.res = with_res ? r.p : nullptr;
        p.B = B; p.T_in = T; p.M = T; p.N = N; p.Cin = Cin; p.taps = taps; p.dil = dil; p.pad = (taps * dil - dil) / 2;
        p.x_bstride = (long)T * Cin; p.x_rstride = Cin; p.out_bstride = (long)T * N; p.out_rstride = N;
        SkWorkspace skw;
        skw.ensure(1024, s);
        skw.attach(p);
        DevBuf w3;
        if (dtype == MI_F32 && taps == 1 && gemm_x3_enabled()) {      // the bf16x3 kernel needs the weight planes
            w3.ensure((size_t)3 * nw * 2);
            split3_planes((const float*)w.p, w3.p, (long)nw, s);
            p.w3 = w3.p;
        }
        DevBuf xp, w3p;
        if (p.w3 && gemm_x3p_enabled() && N % 128 == 0 && Cin % 32 == 0 && p.pad == 0) {       // ... and the panel-plane form (gemm_x3p.hip)
            const int np = x3p_planes();
            xp.ensure((size_t)x3p_bytes((long)B * T, Cin, np)); w3p.ensure((size_t)x3p_bytes(N, Cin, np));
            x3p_split_rows((const float*)x.p, Cin, xp.p, B * T, Cin, s, np);
            x3p_split_rows((const float*)w.p, Cin, w3p.p, N, Cin, s, np);
            p.xp = xp.p; p.w3p = w3p.p; p.np = np;
        }
        // panel-plane launches with MI355TTS_BENCH_WSETS: n copies of the weight PLANES (the kernel never reads p.w)
        DevBuf w3psets;
        const size_t w3pb = p.w3p ? (size_t)x3p_bytes(N, Cin, p.np) : 0;
        if (nsets > 1 && p.w3p) {
            w3psets.ensure((size_t)nsets * w3pb);
            for (int i = 0; i < nsets; ++i) MI_HIP(hipMemcpyAsync((char*)w3psets.p + (size_t)i * w3pb, w3p.p, w3pb, hipMemcpyDeviceToDevice, s));
        }
        for (int i = 0; i < 3; ++i) launch_conv_gemm(p, s);
        hipEvent_t e0, e1;
        MI_HIP(hipEventCreate(&e0)); MI_HIP(hipEventCreate(&e1));
        MI_HIP(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) {
            if (nsets > 1) p.w = (char*)wsets.p + (size_t)(i % nsets) * nw * es;
            if (nsets > 1 && p.w3p) p.w3p = (char*)w3psets.p + (size_t)(i % nsets) * w3pb;
            launch_conv_gemm(p, s);
        }
        MI_HIP(hipEventRecord(e1, s));
        MI_HIP(hipEventSynchronize(e1));
        float t = 0.f;
        MI_HIP(hipEventElapsedTime(&t, e0, e1));
        *ms = (double)t / iters;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
    });
}

// ---------------------------------------------------------------------------------------------------------------
// IndexTTS graph A (cond.hip)
// ---------------------------------------------------------------------------------------------------------------
int64_t mi_indextts_cond_param_count(const int32_t* cfg, int n_cfg) {
    int64_t n = -1;
    int rc = guard([&] { n = cond_param_count(parse_cond_cfg(cfg, n_cfg)); });
    return rc == MI_OK ? n : (int64_t)rc;
}

mi_cond* mi_indextts_cond_create(const int32_t* cfg, int n_cfg, const float* weights, int64_t n_weights, int device) {
    mi_cond* h = nullptr;
    int rc = guard([&] {
        MI_REQUIRE(weights != nullptr, "mi_indextts_cond_create: null weights");
        Cond* impl = cond_create(parse_cond_cfg(cfg, n_cfg), weights, n_weights, device);
        h = new mi_cond; h->impl = impl;
    });
    return rc == MI_OK ? h : nullptr;
}

void mi_indextts_cond_destroy(mi_cond* h) {
    if (!h) return;
    cond_destroy(h->impl);
    delete h;
}

int mi_indextts_cond_run(mi_cond* h, const int16_t* audio, int64_t L, float* conds, float* conds_latent, float* mel, int mem) {
    return guard([&] {
        MI_REQUIRE(h && h->impl, "mi_indextts_cond_run: null handle");
        std::lock_guard<std::mutex> lk_(h->mu);
        MI_REQUIRE(mem == MI_HOST || mem == MI_DEVICE, "mi_indextts_cond_run: bad mem kind");
        cond_run(h->impl, audio, (long)L, conds, conds_latent, mel, mem);
    });
}

int mi_set_option(const char* key, int64_t value) {
    return guard([&] {
        MI_REQUIRE(key != nullptr, "mi_set_option: null key");
        MI_REQUIRE(gemm_set_option(key, (long)value) || gpt_set_option(key, (long)value) || aa_conv_set_option(key, (long)value) ||
                       attn_set_option(key, (long)value) || bigvgan_set_option(key, (long)value), "mi_set_option: unknown key");
        option_epoch_bump();
    }, /*exclusive=*/true);
}

int mi_device_pci_bus_id(int device, char* buf, int cap) {
    return guard([&] {
        MI_REQUIRE(buf && cap >= 16, "mi_device_pci_bus_id: buffer of at least 16 bytes");
        MI_HIP(hipDeviceGetPCIBusId(buf, cap, device));
    });
}

int mi_prof_enable(int family_mask) { prof_enable((unsigned)family_mask); return MI_OK; }
int mi_prof_reset(void) { return guard([&] { prof_reset(); }); }
int mi_prof_kernel_count(void) { return prof_kernel_count(); }
int mi_prof_kernel_get(int index, char* name, int name_cap, char* family, int family_cap, double* ms, int64_t* launches,
                       double* bytes, double* flops) {
    return guard([&] {
        std::string n;
        int f = 0;
        double m = 0, b = 0, fl = 0;
        int64_t l = 0;
        MI_REQUIRE(name && name_cap > 1 && prof_kernel_get(index, &n, &f, &m, &l, &b, &fl), "mi_prof_kernel_get: bad index");
        std::snprintf(name, (size_t)name_cap, "%s", n.c_str());
        static const char* fams[] = {"conv_gemm", "aa_act", "conv_post", "attn", "norm", "other"};
        if (family && family_cap > 1) std::snprintf(family, (size_t)family_cap, "%s", fams[f]);
        if (ms) *ms = m;
        if (launches) *launches = l;
        if (bytes) *bytes = b;
        if (flops) *flops = fl;
    });
}
int mi_prof_get(const char* family, double* ms, int64_t* launches, double* bytes, double* flops) {
    return guard([&] {
        MI_REQUIRE(family != nullptr, "mi_prof_get: null family");
        const int f = prof_family(family);
        MI_REQUIRE(f >= 0, "mi_prof_get: unknown kernel family");
        prof_get(f, ms, launches, bytes, flops);
    });
}

}  // extern "C"
