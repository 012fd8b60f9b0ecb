// permlane32.hip — what __builtin_amdgcn_permlane32_swap(v, v) returns per lane (tools only).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/permlane32.hip -o tools/ubench/bin/permlane32 && tools/ubench/bin/permlane32
#include <hip/hip_runtime.h>
#include <cstdio>
template <bool two_regs> __global__ void k(float* o) {
    const int l = threadIdx.x;
    float v = (float)(l * 3 % 64) + 0.5f * l;          // distinct per lane
    o[l] = v;
    unsigned w = __builtin_bit_cast(unsigned, v);
    if (two_regs) asm volatile("v_mov_b32 %0, %1" : "=v"(w) : "v"(v));          // an opaque copy: two registers for sure
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), w, false, false);
    o[64 + l] = __builtin_bit_cast(float, r[0]);
    o[128 + l] = __builtin_bit_cast(float, r[1]);
    o[192 + l] = __shfl_xor(v, 32);
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 4);
  for (int pass = 0; pass < 2; ++pass) {
    if (pass) hipLaunchKernelGGL(k<true>, dim3(1), dim3(64), 0, 0, d); else hipLaunchKernelGGL(k<false>, dim3(1), dim3(64), 0, 0, d);
    printf("%s\n", pass ? "-- swap(v, opaque copy of v)" : "-- swap(v, v)");
    float h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok_max = 1;
    for (int l = 0; l < 64; ++l) {
        const float want = h[192 + l];
        const int r0_is_partner = h[64 + l] == want, r1_is_partner = h[128 + l] == want, r0_is_own = h[64 + l] == h[l], r1_is_own = h[128 + l] == h[l];
        if (l % 16 == 0 || l == 63) printf("lane %2d own %6.1f partner %6.1f r0 %6.1f r1 %6.1f  (r0 own %d partner %d | r1 own %d partner %d)\n", l, h[l], want, h[64 + l], h[128 + l], r0_is_own, r0_is_partner, r1_is_own, r1_is_partner);
        const float m = h[64 + l] > h[128 + l] ? h[64 + l] : h[128 + l];
        const float wm = h[l] > want ? h[l] : want;
        if (m != wm) ok_max = 0;
    }
    printf("max(r0, r1) == max(own, partner) in every lane: %d\n", ok_max);
  }
    return 0;
}
