// Do MFMA and VALU work of DIFFERENT waves of one SIMD overlap on gfx950?  (hipcc --offload-arch=gfx950 -O3; run on the GPU box)
// Workgroup = 8 waves = two per SIMD (wave w and w + 4 share SIMD w % 4).  Waves 0-3: NM v_mfma_f32_32x32x16_f16 in NCH independent
// accumulator chains.  Waves 4-7: NV v_fma_f32 in 8 independent chains (inline asm: no packing).  Every wave reports its own
// s_memtime cycles (shader clock), so the numbers do not depend on the clock the power manager picks.
//   mode 1: MFMA waves only   2: VALU waves only   3: both   4: both, MFMA waves at s_setprio 3
//   mode 5: ONE wave per SIMD issues both, 8 v_fma_f32 behind every MFMA
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));
#define PK8 _Pragma("unroll") for (int i = 0; i < 8; i += 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(float2v*)&x[i]) : "v"(c2a), "v"(c2b))
#define EXP8 _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]))
#define FMA8 _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(0.999f), "v"(0.001f))

template <int NCH>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int NM, int NV, int mode) {
    const int wave = threadIdx.x >> 6;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 0.5f + 0.001f * (threadIdx.x + i);
    f32x16 acc[NCH];
    for (int c = 0; c < NCH; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    __syncthreads();
    const float2v c2a = {0.999f, 0.999f}, c2b = {0.001f, 0.001f};
    const long long t0 = __builtin_readcyclecounter();
    if (mode >= 6) {            // VALU only: 6 fma on all eight waves ; 7 / 8 v_pk_fma_f32 on four / eight waves (NV / 2 instructions of 2) ; 9 / 10 v_exp_f32 on four / eight
        const bool on = wave >= 4 || mode == 6 || mode == 8 || mode == 10;
        if (on) {
            if (mode == 6) for (int it = 0; it < NV / 8; ++it) { FMA8; }
            else if (mode <= 8) for (int it = 0; it < NV / 8; ++it) { PK8; }
            else for (int it = 0; it < NV / 8; ++it) { EXP8; }
        }
    } else if (mode == 5) {
        if (wave < 4)
            for (int it = 0; it < NM / NCH; ++it)
#pragma unroll
                for (int c = 0; c < NCH; ++c) { acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0); FMA8; }
    } else if (wave < 4) {
        if (mode != 2) {
            if (mode == 4) __builtin_amdgcn_s_setprio(3);
            for (int it = 0; it < NM / NCH; ++it)
#pragma unroll
                for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
        }
    } else if (mode != 1) {
        for (int it = 0; it < NV / 8; ++it) { FMA8; }
    }
    float r = 0.f;
    for (int c = 0; c < NCH; ++c) r += acc[c][0];
    for (int i = 0; i < 8; ++i) r += x[i];
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int NCH>
static void run(const char* name, int mode, int NM, int NV) {
    float* d; long long* c; (void)hipMalloc(&d, 4096); (void)hipMalloc(&c, 256 * 8 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<NCH>, dim3(256), dim3(512), 0, 0, d, c, NM, NV, mode);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(256 * 8);
    (void)hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w] / 1024.0;
    printf("%-44s chains %d : MFMA waves %9.0f cycles (%5.1f per MFMA)   VALU waves %9.0f cycles (%5.2f per v_fma_f32)\n", name, NCH, m, m / NM, v, v / NV);
    (void)hipFree(d); (void)hipFree(c);
}

int main() {
    const int NM = 8192, NV = 65536;           // 8 v_fma_f32 per MFMA: 32 + 32 cycles of issue if nothing overlaps
    run<4>("MFMA waves only", 1, NM, NV);
    run<4>("VALU waves only", 2, NM, NV);
    run<4>("both, different waves of a SIMD", 3, NM, NV);
    run<4>("both, MFMA waves at s_setprio 3", 4, NM, NV);
    run<1>("MFMA waves only, ONE dependent chain", 1, NM, NV);
    run<1>("both, ONE dependent chain", 3, NM, NV);
    run<1>("both, ONE dependent chain, s_setprio 3", 4, NM, NV);
    run<4>("one wave issues both (8 FMA per MFMA)", 5, NM, NV);
    run<1>("one wave issues both, ONE dependent chain", 5, NM, NV);
    run<1>("v_fma_f32 on BOTH waves of a SIMD", 6, NM, NV);
    run<1>("v_pk_fma_f32, one wave per SIMD (per instruction: x2)", 7, NM, NV);
    run<1>("v_pk_fma_f32, both waves (per instruction: x2)", 8, NM, NV);
    run<1>("v_exp_f32, one wave per SIMD", 9, NM, NV);
    run<1>("v_exp_f32, both waves", 10, NM, NV);
    return 0;
}
