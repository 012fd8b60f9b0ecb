// fillrate.hip — per-CU L2 -> LDS fill rate on gfx950: LDS-DMA vs register staging (tuning aid, not product code).
//   hipcc --offload-arch=gfx950 -O3 -o fillrate fillrate.hip && ./fillrate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// every block streams `iters` x 32 KB out of its own `span` bytes (wrapping), 512 threads, 4 x 16 B per thread per step
template <int MODE>
__global__ __launch_bounds__(512) void fill(const uint4* __restrict__ src, long span16, int iters, float* sink) {
    __shared__ __attribute__((aligned(1024))) uint4 smem[4 * 2048];          // 4 stages x 32 KB
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint4* base = src + (long)blockIdx.x * span16;
    float acc = 0.f;
    long off = 0;
    if (MODE == 1 || MODE == 5) {              // register ring: loads run three steps ahead of the ds_writes
        uint4 r[3][4];
        auto ld = [&](uint4 (&d)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = base[off + (wave * 4 + j) * 64 + lane];
            off += 2048; if (off + 2048 > span16) off = 0;
        };
        ld(r[0]); ld(r[1]); ld(r[2]);
        for (int it = 0; it < iters; it += 3) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                uint4* st = smem + ((it + u) & 3) * 2048;
                if (MODE == 5) {
                    const uint4* rd = smem + ((it + u + 2) & 3) * 2048;
#pragma unroll
                    for (int j = 0; j < 12; ++j) { uint4 v = rd[((wave * 12 + j) * 64 + lane) & 2047]; acc += __uint_as_float(v.x ^ v.w); }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) st[(wave * 4 + j) * 64 + lane] = r[u][j];
                ld(r[u]);
            }
        }
    } else
    for (int it = 0; it < iters; ++it) {
        uint4* st = smem + (it & 3) * 2048;
        if (MODE == 0) {                       // LDS-DMA, 3 steps in flight
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off + (wave * 4 + j) * 64 + lane),
                                                 (lds_void*)(st + (wave * 4 + j) * 64), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        } else if (MODE == 1) {                // registers -> ds_write_b128
            uint4 r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = base[off + (wave * 4 + j) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) st[(wave * 4 + j) * 64 + lane] = r[j];
        } else if (MODE == 2) {                // loads only
            uint4 r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = base[off + (wave * 4 + j) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += __uint_as_float(r[j].x ^ r[j].y ^ r[j].z ^ r[j].w);
        } else if (MODE == 3) {                // DMA + concurrent ds_read_b128 of another stage (6 reads per DMA instr ~ GEMM ratio 3:1 bytes)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off + (wave * 4 + j) * 64 + lane),
                                                 (lds_void*)(st + (wave * 4 + j) * 64), 16, 0, 0);
            const uint4* rd = smem + ((it + 2) & 3) * 2048;
#pragma unroll
            for (int j = 0; j < 12; ++j) { uint4 v = rd[((wave * 12 + j) * 64 + lane) & 2047]; acc += __uint_as_float(v.x ^ v.w); }
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        } else if (MODE == 4) {                // ds_read only (same reads as mode 3)
            const uint4* rd = smem + ((it + 2) & 3) * 2048;
#pragma unroll
            for (int j = 0; j < 12; ++j) { uint4 v = rd[((wave * 12 + j) * 64 + lane) & 2047]; acc += __uint_as_float(v.x ^ v.w); }
        } else if (MODE == 5) {                // registers -> ds_write + concurrent ds_reads
            uint4 r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = base[off + (wave * 4 + j) * 64 + lane];
            const uint4* rd = smem + ((it + 2) & 3) * 2048;
#pragma unroll
            for (int j = 0; j < 12; ++j) { uint4 v = rd[((wave * 12 + j) * 64 + lane) & 2047]; acc += __uint_as_float(v.x ^ v.w); }
#pragma unroll
            for (int j = 0; j < 4; ++j) st[(wave * 4 + j) * 64 + lane] = r[j];
        }
        off += 2048;
        if (off + 2048 > span16) off = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE != 2 && MODE != 4) acc += __uint_as_float(smem[tid].x);
    if (acc == 12345.678f) sink[0] = acc;
}

int main() {
    const int blocks = 256, iters = 4096;
    const char* names[] = {"lds-dma (3 in flight)", "regs -> ds_write_b128", "global loads only", "lds-dma + ds_reads", "ds_reads only", "regs -> ds_write + ds_reads"};
    for (long span : {65536L, 1L << 20, 8L << 20}) {               // bytes per block: L1/L2-hot, L2-ish (256 MB total = MALL), MALL/HBM
        uint4* src; float* sink;
        CK(hipMalloc(&src, blocks * span)); CK(hipMalloc(&sink, 4));
        CK(hipMemset(src, 1, blocks * span));
        for (int mode = 0; mode < 6; ++mode) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            auto run = [&](int it) {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(fill<0>, dim3(blocks), dim3(512), 0, 0, src, span / 16, it, sink); break;
                    case 1: hipLaunchKernelGGL(fill<1>, dim3(blocks), dim3(512), 0, 0, src, span / 16, it, sink); break;
                    case 2: hipLaunchKernelGGL(fill<2>, dim3(blocks), dim3(512), 0, 0, src, span / 16, it, sink); break;
                    case 3: hipLaunchKernelGGL(fill<3>, dim3(blocks), dim3(512), 0, 0, src, span / 16, it, sink); break;
                    case 4: hipLaunchKernelGGL(fill<4>, dim3(blocks), dim3(512), 0, 0, src, span / 16, it, sink); break;
                    default: hipLaunchKernelGGL(fill<5>, dim3(blocks), dim3(512), 0, 0, src, span / 16, it, sink); break;
                }
            };
            run(64);
            CK(hipEventRecord(e0)); run(iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double bytes = (mode == 4 ? 0.0 : (double)blocks * iters * 32768.0);
            const double rbytes = (mode >= 3) ? (double)blocks * iters * 8 * 12 * 1024.0 : 0.0;
            printf("span %8ld B/block  %-30s %8.3f ms  fill %7.1f GB/s/CU (%5.1f B/clk @2.4GHz)  lds reads %7.1f GB/s/CU\n", span, names[mode], ms,
                   bytes / ms / 1e6 / blocks, bytes / ms / 1e6 / blocks / 2.4, rbytes / ms / 1e6 / blocks);
        }
        CK(hipFree(src)); CK(hipFree(sink));
    }
    return 0;
}
