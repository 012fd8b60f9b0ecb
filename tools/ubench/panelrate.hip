// panelrate.hip — L2 -> LDS fill rate of the implicit-GEMM operand pattern on gfx950 (tuning aid, not product code):
// 256 workgroups as a 32 x 8 tile grid, each streams its A panel (256 rows) and B panel (256 rows) of a row-major
// matrix with row stride S bytes, ROWB bytes of every row per step, by LDS-DMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int ROWB, int XCDMAP>
__global__ __launch_bounds__(512) void panel(const char* __restrict__ A, const char* __restrict__ B, long S, int steps, float* sink) {
    __shared__ __attribute__((aligned(1024))) char smem[131072];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int L = blockIdx.x;
    int rowt, nt;
    if (XCDMAP == 0) { nt = L / 32; rowt = L % 32; }                 // row tiles fastest (what the kernels do)
    else { const int x = L & 7, j = L >> 3; rowt = x * 4 + (j & 3); nt = j >> 2; }   // XCD x owns row tiles 4x..4x+3, all 8 column tiles
    constexpr int LPR = ROWB / 16;                                   // lanes per row
    constexpr int RPI = 64 / LPR;                                    // rows per wave instruction
    constexpr int NI = 512 / RPI / 8;                                // instructions per wave per step
    const char* base = wave < 4 ? A + (long)rowt * 256 * S : B + (long)nt * 256 * S;
    const int r0 = (wave & 3) * 64;                                  // this wave's 64 rows of the panel
    const int lrow = lane / LPR, lcol = (lane % LPR) * 16;
    for (int st = 0; st < steps; ++st) {
        char* dst = smem + (st & (ROWB == 64 ? 3 : 1)) * (512 * ROWB);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int row = r0 + j * RPI + lrow;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (long)row * S + (long)st * ROWB + lcol),
                                             (lds_void*)(dst + (wave * NI + j) * 1024), 16, 0, 0);
        }
        if (ROWB == 64) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (smem[tid] == 77 && smem[tid + 1000] == 99) sink[0] = 1.f;
}

int main() {
    float* sink; CK(hipMalloc(&sink, 4));
    for (long S : {16384L, 16384L + 128, 1536L, 8448L * 2}) {
        const long rows = 32 * 256;
        char *A, *B;
        CK(hipMalloc(&A, rows * S)); CK(hipMalloc(&B, rows * S));
        CK(hipMemset(A, 1, rows * S)); CK(hipMemset(B, 1, rows * S));
        for (int v = 0; v < 4; ++v) {
            const int rowb = (v & 1) ? 128 : 64, xm = v >> 1;
            const int steps = (int)(S / rowb) & ~3;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            auto run = [&]() {
                if (v == 0) hipLaunchKernelGGL((panel<64, 0>), dim3(256), dim3(512), 0, 0, A, B, S, steps, sink);
                if (v == 1) hipLaunchKernelGGL((panel<128, 0>), dim3(256), dim3(512), 0, 0, A, B, S, steps, sink);
                if (v == 2) hipLaunchKernelGGL((panel<64, 1>), dim3(256), dim3(512), 0, 0, A, B, S, steps, sink);
                if (v == 3) hipLaunchKernelGGL((panel<128, 1>), dim3(256), dim3(512), 0, 0, A, B, S, steps, sink);
            };
            run();
            const int reps = 10;
            CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) run(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double bytes = 256.0 * steps * 512 * rowb * reps;
            printf("S %6ld  rowB %3d  xcdmap %d  steps %4d: %8.3f ms/launch  %7.1f GB/s/CU (%5.1f B/clk @2.4GHz)  %6.2f TB/s chip\n", S, rowb, xm, steps,
                   ms / reps, bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.4, bytes / ms / 1e9);
        }
        CK(hipFree(A)); CK(hipFree(B));
    }
    return 0;
}
