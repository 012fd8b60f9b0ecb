// permlane32c.hip — the half exchange through inline asm: both registers come back (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float xor32_max(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
__global__ void k(float* o) {
    const int l = threadIdx.x;
    float v = (float)(l * 7 % 64) + 0.25f * l;
    o[l] = fmaxf(v, __shfl_xor(v, 32));
    o[64 + l] = xor32_max(v);
}
int main() {
    float* d; (void)hipMalloc(&d, 128 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[128]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) if (h[l] != h[64 + l]) ok = 0;
    printf("inline-asm permlane32 max == shuffle max in every lane: %d   (lane 5: %g %g, lane 40: %g %g)\n", ok, h[5], h[69], h[40], h[104]);
    return 0;
}
