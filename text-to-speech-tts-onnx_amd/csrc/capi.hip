// capi.hip — extern "C" boundary of libmi355tts.so (declarations: include/mi355tts.h).
#include "common.h"
#include "bigvgan.h"

using namespace mi;

struct mi_bigvgan { BigVGAN* impl; };

template <typename F> static int guard(F&& f) {
    try {
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
        h = new mi_bigvgan{new BigVGAN(g, weights, n_weights, dtype, device)};
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
        MI_REQUIRE(mem == MI_HOST || mem == MI_DEVICE, "mi_bigvgan_forward: bad mem kind");
        h->impl->run(mel, B, frames, nullptr, out, mem);
    });
}

int mi_bigvgan_forward_f32(mi_bigvgan* h, const float* mel, int B, int frames, float* out, int mem) {
    return guard([&] {
        MI_REQUIRE(h && h->impl, "mi_bigvgan_forward_f32: null handle");
        MI_REQUIRE(mem == MI_HOST || mem == MI_DEVICE, "mi_bigvgan_forward_f32: bad mem kind");
        h->impl->run(mel, B, frames, out, nullptr, mem);
    });
}

int mi_aa_activation1d(const float* x, int B, int C, int T, const float* alpha_log, const float* beta_log,
                       int logscale, int post, int dtype, float* y) {
    return guard([&] {
        MI_REQUIRE(x && y && alpha_log && beta_log && B > 0 && C > 0 && T > 0, "mi_aa_activation1d: bad arguments");
        unit_aa_activation1d(x, B, C, T, alpha_log, beta_log, logscale, post, dtype, y);
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

int mi_prof_enable(int family_mask) { prof_enable((unsigned)family_mask); return MI_OK; }
int mi_prof_reset(void) { return guard([&] { prof_reset(); }); }
int mi_prof_get(const char* family, double* ms, int64_t* launches, double* bytes, double* flops) {
    return guard([&] {
        MI_REQUIRE(family != nullptr, "mi_prof_get: null family");
        const int f = prof_family(family);
        MI_REQUIRE(f >= 0, "mi_prof_get: unknown kernel family");
        prof_get(f, ms, launches, bytes, flops);
    });
}

}  // extern "C"
