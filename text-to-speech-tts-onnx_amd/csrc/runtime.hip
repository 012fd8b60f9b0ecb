// runtime.hip — error state, uploads, and the event-based per-family kernel profiler.
#include "common.h"
#include <mutex>
#include <atomic>
#include <algorithm>

namespace mi {

static thread_local std::string g_err;
void set_last_error(const std::string& m) { g_err = m; }
const std::string& last_error() { return g_err; }

static std::atomic<long> g_option_epoch{0};
long option_epoch() { return g_option_epoch.load(); }
void option_epoch_bump() { g_option_epoch.fetch_add(1); }
std::shared_mutex& option_lock() { static std::shared_mutex m; return m; }

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

template <typename TO>
__global__ __launch_bounds__(256) void convert_f32_kernel(const float* __restrict__ src, TO* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = (TO)src[i];
}

const float* BlobReader::host(const float* p, size_t n) {
    if (mem == MI_HOST) return p;
    keep.emplace_back(n);
    MI_HIP(hipMemcpy(keep.back().data(), p, n * 4, hipMemcpyDeviceToHost));
    return keep.back().data();
}

void BlobReader::put(DevBuf& dst, const float* p, size_t n, int dt, size_t dst_off) {
    const size_t es = dtype_size(dt);
    if (dst_off == 0) dst.ensure(n * es);
    MI_REQUIRE((dst_off + n) * es <= dst.bytes, "blob reader: destination too small");
    if (mem == MI_HOST) {
        if (dst_off == 0) { upload_as(dst, p, n, dt, s); return; }
        DevBuf tmp;
        upload_as(tmp, p, n, dt, s);
        MI_HIP(hipMemcpyAsync((char*)dst.p + dst_off * es, tmp.p, n * es, hipMemcpyDeviceToDevice, s));
        MI_HIP(hipStreamSynchronize(s));
        return;
    }
    if (n == 0) return;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
    if (dt == MI_F32) MI_HIP(hipMemcpyAsync((char*)dst.p + dst_off * 4, p, n * 4, hipMemcpyDeviceToDevice, s));
    else if (dt == MI_F16) hipLaunchKernelGGL(convert_f32_kernel<f16>, dim3(blocks), dim3(256), 0, s, p, dst.as<f16>() + dst_off, n);
    else hipLaunchKernelGGL(convert_f32_kernel<bf16>, dim3(blocks), dim3(256), 0, s, p, dst.as<bf16>() + dst_off, n);
    MI_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// profiler
// ---------------------------------------------------------------------------------------------
struct KStat { double ms = 0, bytes = 0, flops = 0; int64_t launches = 0; int fam = 0; };
struct Pending { int fam; hipEvent_t e0, e1; const std::string* kname; double bytes, flops; };
static std::mutex g_pm;
static unsigned g_prof_mask = 0;
static std::vector<Pending> g_pending;
static double g_ms[FAM_COUNT], g_bytes[FAM_COUNT], g_flops[FAM_COUNT];
static int64_t g_launches[FAM_COUNT];
static std::map<std::string, KStat> g_kstat;          // per kernel instantiation (node addresses are stable)
static thread_local const std::string* g_cur_kernel = nullptr;
static const char* const g_fam_names[FAM_COUNT] = {"conv_gemm", "aa_act", "conv_post", "attn", "norm", "other"};

void prof_set_kernel(const char* expr, const char* t, const char* to) {
    if (!g_prof_mask) return;
    std::string n(expr);
    while (!n.empty() && n.front() == '(') n.erase(0, 1);
    while (!n.empty() && n.back() == ')') n.pop_back();
    if (t && to) {
        const size_t p = n.find("<T, TO");
        if (p != std::string::npos) n.replace(p, 6, std::string("<") + t + ", " + to);
    } else if (t) {
        const size_t p = n.find("<T");
        if (p != std::string::npos) n.replace(p, 2, std::string("<") + t);
    }
    std::lock_guard<std::mutex> lk(g_pm);
    g_cur_kernel = &g_kstat.emplace(n, KStat{}).first->first;
}

ProfScope::ProfScope(int family, hipStream_t stream, double bytes, double flops) : fam(family), s(stream) {
    on = (g_prof_mask >> fam) & 1u;
    if (!on) return;
    bytes_ = bytes; flops_ = flops;
    g_cur_kernel = nullptr;
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
    const std::string* kn = g_cur_kernel;
    if (!kn) kn = &g_kstat.emplace(g_fam_names[fam], KStat{}).first->first;      // launcher did not name its kernel
    g_cur_kernel = nullptr;
    g_pending.push_back({fam, e0, e1, kn, bytes_, flops_});
}
void prof_collect() {
    std::lock_guard<std::mutex> lk(g_pm);
    for (auto& p : g_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.e1) == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            g_ms[p.fam] += ms;
            KStat& k = g_kstat[*p.kname];
            k.ms += ms; k.bytes += p.bytes; k.flops += p.flops; k.launches += 1; k.fam = p.fam;
        }
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
    for (auto& kv : g_kstat) kv.second = KStat{};         // keep the nodes: pending name pointers stay valid
}
int prof_kernel_count() {
    prof_collect();
    std::lock_guard<std::mutex> lk(g_pm);
    int n = 0;
    for (auto& kv : g_kstat) n += kv.second.launches > 0;
    return n;
}
bool prof_kernel_get(int idx, std::string* name, int* fam, double* ms, int64_t* launches, double* bytes, double* flops) {
    prof_collect();
    std::lock_guard<std::mutex> lk(g_pm);
    for (auto& kv : g_kstat) {
        if (kv.second.launches == 0) continue;
        if (idx-- == 0) {
            *name = kv.first; *fam = kv.second.fam; *ms = kv.second.ms; *launches = kv.second.launches;
            *bytes = kv.second.bytes; *flops = kv.second.flops;
            return true;
        }
    }
    return false;
}
int prof_family(const char* name) {
    for (int i = 0; i < FAM_COUNT; ++i)
        if (!std::strcmp(name, g_fam_names[i])) return i;
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
