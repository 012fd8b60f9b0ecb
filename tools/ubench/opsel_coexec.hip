// opsel_coexec.hip — minimal form of the aa_conv run-to-run difference (round 3; tools only, compiled on the GPU box):
// is `v_pk_fma_f32 vD, vT, v[2n:2n+1], vD op_sel:[0,1,0]` (both halves read the HIGH register of the pair) exact while a
// wave of ANOTHER workgroup runs MFMAs on the same SIMD?  Per-lane DISTINCT data (the round-2 probe valu_mfma_coexec.hip
// used wave-uniform values, which cannot show a lane / register mix-up), two chains per lane that must agree bit for bit:
//   A: x in the high register, low register = 1000, op_sel:[0,1,0]
//   B: x in the low register,  op_sel_hi:[1,0,1]                     (the encoding every other sample of the AA run got)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/opsel_coexec.hip -o opsel && ./opsel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// mode bit0: odd workgroups run MFMAs ; bit1: the VALU role re-reads its x pair from LDS every iteration (ds_read_b32 +
// shift / and unpack of a bf16 pair, as the AA phase does)
__global__ __launch_bounds__(256) void k(unsigned* bad, unsigned* rowhist, const float* in, int iters, int mode, float* sink) {
    extern __shared__ unsigned lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    if ((mode & 1) && (blockIdx.x & 1)) {
        f32x16 acc = {0};
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (tid + i)); b[i] = (__bf16)(0.002f * (i + 1)); }
        for (int it = 0; it < iters * 6; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += acc[r];
        if (s == 12345.678f) sink[0] = s;
        return;
    }
    const float xa = in[2 * tid], xb = in[2 * tid + 1];
    {   // bf16 pair of (xa, xb) in LDS
        const unsigned ua = __float_as_uint(xa) >> 16, ub = __float_as_uint(xb) & 0xffff0000u;
        lds[tid] = ua | ub;
    }
    __syncthreads();
    const f2 t0 = {in[1024], in[1025]}, t1 = {in[1026], in[1027]};
    f2 rA = {0.f, 0.f}, rB = {0.f, 0.f};
    float x0 = __uint_as_float(__float_as_uint(xa) & 0xffff0000u), x1 = __uint_as_float(__float_as_uint(xb) & 0xffff0000u);
    for (int it = 0; it < iters; ++it) {
        if (mode & 2) {
            const unsigned v = ((volatile unsigned*)lds)[tid];
            x0 = __uint_as_float(v << 16); x1 = __uint_as_float(v & 0xffff0000u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            f2 pa = {1000.f, u & 1 ? x0 : x1};
            f2 pb = {u & 1 ? x0 : x1, -1000.f};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(rA) : "v"(u & 2 ? t0 : t1), "v"(pa));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(rB) : "v"(u & 2 ? t0 : t1), "v"(pb));
        }
        rA *= f2{0.5f, 0.5f}; rB *= f2{0.5f, 0.5f};
    }
    if (__float_as_uint(rA.x) != __float_as_uint(rB.x) || __float_as_uint(rA.y) != __float_as_uint(rB.y)) {
        atomicAdd(bad, 1u);
        atomicAdd(rowhist + (lane >> 4), 1u);
    }
}

int main() {
    const int blocks = 4096, iters = 4000;
    std::vector<float> hin(1028);
    unsigned s = 777;
    for (auto& v : hin) { s = s * 1664525u + 1013904223u; v = (float)((s >> 8) & 0xffff) / 65536.f * 4.f - 2.f; }
    hin[1024] = 0.0041f; hin[1025] = -0.0173f; hin[1026] = 0.0522f; hin[1027] = -0.0932f;
    float *in, *sink; unsigned *bad, *rows;
    hipMalloc(&in, hin.size() * 4); hipMalloc(&sink, 4); hipMalloc(&bad, 4); hipMalloc(&rows, 16);
    hipMemcpy(in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (size_t ldsb : {(size_t)70 * 1024, (size_t)100 * 1024})          // 70 KB: two workgroups per CU ; 100 KB: one
        for (int mode = 0; mode < 4; ++mode) {
            unsigned tb = 0, tr[4] = {0, 0, 0, 0};
            for (int rep = 0; rep < 10; ++rep) {
                hipMemset(bad, 0, 4); hipMemset(rows, 0, 16);
                hipLaunchKernelGGL(k, dim3(blocks), dim3(256), ldsb, 0, bad, rows, in, iters, mode, sink);
                hipDeviceSynchronize();
                unsigned hb, hr[4];
                hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(hr, rows, 16, hipMemcpyDeviceToHost);
                tb += hb; for (int i = 0; i < 4; ++i) tr[i] += hr[i];
            }
            printf("lds %3zu KB (%s workgroup(s) per CU)  mfma neighbours %d  lds re-read %d : %u lanes with chain A != chain B (16-lane rows: %u %u %u %u)\n",
                   ldsb / 1024, ldsb < 80 * 1024 ? "two" : "one", mode & 1, (mode >> 1) & 1, tb, tr[0], tr[1], tr[2], tr[3]);
        }
    return 0;
}
