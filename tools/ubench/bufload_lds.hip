// bufload_lds.hip — semantics check of buffer_load_dwordx4 ... lds on gfx950 (tuning aid): out-of-range lanes must write
// zeros to LDS (that is what lets the GEMM drop its per-lane validity selects), with the offset split between voffset
// and soffset in every combination.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const float* x, float* out, int nbytes, const int* voffs, const int* soffs, int ncase) {
    __shared__ __attribute__((aligned(1024))) float smem[64 * 4];
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
    for (int c = 0; c < ncase; ++c) {
        for (int i = threadIdx.x; i < 256; i += 64) smem[i] = -7.f;
        __syncthreads();
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)smem, 16, voffs[c] + (int)threadIdx.x * 16, soffs[c], 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        out[c * 2 + 0] = smem[0];            // lane 0, first float
        out[c * 2 + 1] = smem[63 * 4 + 3];   // lane 63, last float
        __syncthreads();
    }
}
int main() {
    const int n = 4096;                      // floats; buffer = 16 KB, values = index
    std::vector<float> h(n * 2);
    for (int i = 0; i < n * 2; ++i) h[i] = (float)i;
    float *x, *out; int *vo, *so;
    hipMalloc(&x, n * 2 * 4); hipMemcpy(x, h.data(), n * 2 * 4, hipMemcpyHostToDevice);
    // cases: (voffset, soffset) in bytes; buffer declared as n floats (16384 B) although 2n are allocated
    std::vector<int> v = {0, 16384 - 1024, 16384 - 512, 16384, -1024, 0, 0, 8192, -2048, 1024};
    std::vector<int> s = {0, 0, 0, 0, 0, 16384 - 1024, 16384, 8192, 1024, -1024};
    const int nc = (int)v.size();
    hipMalloc(&out, nc * 8); hipMalloc(&vo, nc * 4); hipMalloc(&so, nc * 4);
    hipMemcpy(vo, v.data(), nc * 4, hipMemcpyHostToDevice); hipMemcpy(so, s.data(), nc * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x + 0, out, n * 4, vo, so, nc);
    std::vector<float> o(nc * 2);
    hipMemcpy(o.data(), out, nc * 8, hipMemcpyDeviceToHost);
    for (int c = 0; c < nc; ++c)
        printf("voffset %7d soffset %7d : lane0 -> %8.0f   lane63 last -> %8.0f   (in-range expectation %d / %d)\n", v[c], s[c], o[c * 2], o[c * 2 + 1],
               (v[c] + s[c]) / 4, (v[c] + s[c]) / 4 + 63 * 4 + 3);
    return 0;
}
