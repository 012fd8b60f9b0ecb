// permlane32b.hip — decode __builtin_amdgcn_permlane32_swap(a, b): a = lane, b = 100 + lane (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* o) {
    const int l = threadIdx.x;
    float a = (float)l, b = 100.f + (float)l;
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    o[l] = __builtin_bit_cast(float, r[0]);
    o[64 + l] = __builtin_bit_cast(float, r[1]);
}
int main() {
    float* d; (void)hipMalloc(&d, 128 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[128]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l : {0, 1, 31, 32, 33, 63}) printf("lane %2d: r0 %5.0f r1 %5.0f\n", l, h[l], h[64 + l]);
    return 0;
}
