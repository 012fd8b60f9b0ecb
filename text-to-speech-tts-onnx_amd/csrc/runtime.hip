// runtime.hip — error state, uploads, and the event-based per-family kernel profiler.
#include "common.h"
#include <mutex>
#include <atomic>

namespace mi {

static thread_local std::string g_err;
void set_last_error(const std::string& m) { g_err = m; }
const std::string& last_error() { return g_err; }

static std::atomic<long> g_option_epoch{0};
long option_epoch() { return g_option_epoch.load(); }
void option_epoch_bump() { g_option_epoch.fetch_add(1); }

void upload_f32(DevBuf& dst, const float* src, size_t n, hipStream_t s) {
    dst.ensure(n * 4);
    MI_HIP(hipMemcpyAsync(dst.p, src, n * 4, hipMemcpyHostToDevice, s));
    MI_HIP(hipStreamSynchronize(s));
}

static inline uint16_t f32_to_bf16_bits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
    const uint32_t lsb = (u >> 16) & 1u;
    u += 0x7fffu + lsb;                                                          // RNE
    return (uint16_t)(u >> 16);
}

void upload_as(DevBuf& dst, const float* src, size_t n, int dt, hipStream_t s) {
    if (dt == MI_F32) { upload_f32(dst, src, n, s); return; }
    std::vector<uint16_t> tmp(n);
    if (dt == MI_F16) {
        for (size_t i = 0; i < n; ++i) { f16 h = (f16)src[i]; std::memcpy(&tmp[i], &h, 2); }
    } else {
        for (size_t i = 0; i < n; ++i) tmp[i] = f32_to_bf16_bits(src[i]);
    }
    dst.ensure(n * 2);
    MI_HIP(hipMemcpyAsync(dst.p, tmp.data(), n * 2, hipMemcpyHostToDevice, s));
    MI_HIP(hipStreamSynchronize(s));
}

// ---------------------------------------------------------------------------------------------
// profiler
// ---------------------------------------------------------------------------------------------
struct Pending { int fam; hipEvent_t e0, e1; };
static std::mutex g_pm;
static unsigned g_prof_mask = 0;
static std::vector<Pending> g_pending;
static double g_ms[FAM_COUNT], g_bytes[FAM_COUNT], g_flops[FAM_COUNT];
static int64_t g_launches[FAM_COUNT];

ProfScope::ProfScope(int family, hipStream_t stream, double bytes, double flops) : fam(family), s(stream) {
    on = (g_prof_mask >> fam) & 1u;
    if (!on) return;
    {
        std::lock_guard<std::mutex> lk(g_pm);
        g_bytes[fam] += bytes; g_flops[fam] += flops; g_launches[fam] += 1;
    }
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { on = false; return; }
    (void)hipEventRecord(e0, s);
}
ProfScope::~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(e1, s);
    std::lock_guard<std::mutex> lk(g_pm);
    g_pending.push_back({fam, e0, e1});
}
void prof_collect() {
    std::lock_guard<std::mutex> lk(g_pm);
    for (auto& p : g_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.e1) == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess)
            g_ms[p.fam] += ms;
        (void)hipEventDestroy(p.e0);
        (void)hipEventDestroy(p.e1);
    }
    g_pending.clear();
}
void prof_enable(unsigned mask) { g_prof_mask = mask; }
unsigned prof_mask() { return g_prof_mask; }
void prof_reset() {
    prof_collect();
    std::lock_guard<std::mutex> lk(g_pm);
    for (int i = 0; i < FAM_COUNT; ++i) { g_ms[i] = g_bytes[i] = g_flops[i] = 0; g_launches[i] = 0; }
}
int prof_family(const char* name) {
    static const char* names[FAM_COUNT] = {"conv_gemm", "aa_act", "conv_post", "attn", "norm", "other"};
    for (int i = 0; i < FAM_COUNT; ++i)
        if (!std::strcmp(name, names[i])) return i;
    return -1;
}
void prof_get(int fam, double* ms, int64_t* launches, double* bytes, double* flops) {
    prof_collect();
    std::lock_guard<std::mutex> lk(g_pm);
    if (ms) *ms = g_ms[fam];
    if (launches) *launches = g_launches[fam];
    if (bytes) *bytes = g_bytes[fam];
    if (flops) *flops = g_flops[fam];
}

}  // namespace mi
