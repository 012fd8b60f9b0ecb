// valu_mfma_coexec.hip — does an fp32 VALU chain give bit-identical results while ANOTHER wave on the same SIMD issues MFMAs?
// Every lane of every VALU wave computes the same chain from the same inputs, so all outputs must be equal.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_mfma_coexec.hip -o coexec && ./coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int OP>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int iters, int mfma_on, float* sink) {
    const int wave = threadIdx.x >> 6;
    // odd workgroups: MFMA role (all four waves) ; even workgroups: VALU role.  Two workgroups share a CU (LDS-limited).
    extern __shared__ float lds[];
    lds[threadIdx.x & 255] = 0.f;
    // 8 waves per workgroup = 2 per SIMD: waves 0-3 run the VALU chain, waves 4-7 issue MFMAs next to them
    const bool mfma_role = wave >= 4;
    if (wave >= 4 && !mfma_on) return;
    if (mfma_role) {
        f32x16 acc = {0};
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (i + 1)); }
        for (int it = 0; it < iters * 4; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += acc[r];
        if (s == 12345.678f) sink[0] = s;
        return;
    }
    float x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3];
    float r0 = 0.f, r1 = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) {            // scalar fma chain
#pragma unroll
            for (int u = 0; u < 16; ++u) { r0 = fmaf(x0, r0, x1); r1 = fmaf(x2, r1, x3); }
        } else if (OP == 1) {     // packed fma chain
            f2 a = {x0, x2}, c = {x1, x3}, r = {r0, r1};
#pragma unroll
            for (int u = 0; u < 16; ++u) r = __builtin_elementwise_fma(a, r, c);
            r0 = r.x; r1 = r.y;
        } else if (OP == 2) {     // v_sin (trans)
#pragma unroll
            for (int u = 0; u < 8; ++u) { r0 = __builtin_amdgcn_sinf(r0 * x0 + x1); r1 = __builtin_amdgcn_sinf(r1 * x2 + x3); }
        } else if (OP == 3) {     // mul + add (unfused)
#pragma unroll
            for (int u = 0; u < 16; ++u) { r0 = __fadd_rn(__fmul_rn(x0, r0), x1); r1 = __fadd_rn(__fmul_rn(x2, r1), x3); }
        } else if (OP == 4) {     // packed fma with SGPR-pair sources (uniform values kept in scalar registers)
            const float s0 = __builtin_amdgcn_readfirstlane(x0), s1 = __builtin_amdgcn_readfirstlane(x2);
            const float c0 = __builtin_amdgcn_readfirstlane(x1), c1 = __builtin_amdgcn_readfirstlane(x3);
            f2 a = {s0, s1}, c = {c0, c1}, r = {r0, r1};
#pragma unroll
            for (int u = 0; u < 16; ++u) r = __builtin_elementwise_fma(a, r, c);
            r0 = r.x; r1 = r.y;
        } else if (OP == 5) {     // values travelling through v_writelane / v_readlane (what an SGPR spill does)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                r0 = fmaf(x0, r0, x1);
                const int bits = __builtin_amdgcn_readlane(__float_as_int(r0), 17);
                int w = __float_as_int(r1);
                asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, 5" : "+v"(w) : "s"(bits));
                r1 = fmaf(x2, __int_as_float(__builtin_amdgcn_readlane(w, 5)), x3);
            }
        } else if (OP == 6) {     // LDS round trip of packed bf16 pairs + shift/and unpack (the AA staging pattern)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const __bf16 a = (__bf16)fmaf(x0, r0, x1), b = (__bf16)fmaf(x2, r1, x3);
                unsigned short ua, ub; __builtin_memcpy(&ua, &a, 2); __builtin_memcpy(&ub, &b, 2);
                ((volatile unsigned*)lds)[threadIdx.x & 255] = (unsigned)ua | ((unsigned)ub << 16);
                const unsigned v = ((volatile unsigned*)lds)[threadIdx.x & 255];
                r0 = __uint_as_float(v << 16); r1 = __uint_as_float(v & 0xffff0000u);
            }
        } else if (OP == 7) {     // op_sel broadcast of the HIGH register of a pair (v_pk_fma ... op_sel:[0,1,0])
            f2 r = {r0, r1};
            const f2 t = {x0, x2};
#pragma unroll
            for (int u = 0; u < 16; ++u) r = __builtin_elementwise_fma(f2{x1, x3}, f2{t.y, t.y}, r * f2{x0, x0});
            r0 = r.x; r1 = r.y;
        }
    }
    out[(size_t)blockIdx.x * 256 + (threadIdx.x & 255)] = r0 + r1 * 1.0000001f;
    (void)wave;
}

template <int OP> static void run(const char* name, int mfma_on, size_t lds_bytes) {
    const int blocks = 2048, iters = 2000;
    float *out, *in, *sink;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&in, 16); hipMalloc(&sink, 4);
    const float hin[4] = {0.9993f, 0.3171f, 0.9871f, 0.2133f};
    hipMemcpy(in, hin, 16, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    std::vector<float> h(blocks * 256);
    long bad = 0, total = 0; unsigned ref = 0; int rowhist[4] = {0};
    for (int rep = 0; rep < 5; ++rep) {
        hipMemset(out, 0, blocks * 256 * 4);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), lds_bytes, 0, out, in, iters, mfma_on, sink);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, blocks * 256 * 4, hipMemcpyDeviceToHost);
        for (int bI = 0; bI < blocks; ++bI) {
            for (int t = 0; t < 256; ++t) {
                unsigned u; memcpy(&u, &h[(size_t)bI * 256 + t], 4);
                if (total == 0) ref = u;
                ++total;
                if (u != ref) { ++bad; ++rowhist[(t & 63) >> 4]; }
            }
        }
    }
    printf("%-12s mfma_on=%d lds=%zu KB: %ld of %ld lane results differ from the first (16-lane row histogram %d %d %d %d)\n", name, mfma_on,
           lds_bytes / 1024, bad, total, rowhist[0], rowhist[1], rowhist[2], rowhist[3]);
    hipFree(out); hipFree(in); hipFree(sink);
}

int main() {
    for (int mf = 0; mf < 2; ++mf)
        for (size_t lds : {(size_t)100 * 1024}) {      // 60 KB: two workgroups per CU ; 100 KB: one
            run<0>("fma", mf, lds); run<1>("pk_fma", mf, lds); run<2>("sin", mf, lds); run<3>("mul+add", mf, lds);
            run<4>("pk_fma sgpr", mf, lds); run<5>("lane rw", mf, lds); run<6>("lds bf16", mf, lds); run<7>("pk op_sel", mf, lds);
        }
    return 0;
}
