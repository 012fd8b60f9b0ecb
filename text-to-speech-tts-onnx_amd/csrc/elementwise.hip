// elementwise.hip — layout conversion at the C-ABI boundary and the BigVGAN conv_post tail.
#include "common.h"

namespace mi {

// (B,C,T) fp32 channels-first -> (B,T,Cpad) T channels-last, zero padded channels.
// 32x32 LDS transpose so both the HBM read (along T) and write (along C) are coalesced.
template <typename T>
__global__ __launch_bounds__(256) void ncl_to_nlc_kernel(const float* __restrict__ x, T* __restrict__ y, int C,
                                                         int Tn, int Cpad) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* xb = x + (long)b * C * Tn;
    T* yb = y + (long)b * Tn * Cpad;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + i * 8, t = t0 + tx;
        tile[ty + i * 8][tx] = (c < C && t < Tn) ? xb[(long)c * Tn + t] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty + i * 8, c = c0 + tx;
        if (t < Tn && c < Cpad) yb[(long)t * Cpad + c] = from_f32<T>(tile[tx][ty + i * 8]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void nlc_to_ncl_kernel(const T* __restrict__ x, float* __restrict__ y, int C,
                                                         int Tn) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const T* xb = x + (long)b * Tn * C;
    float* yb = y + (long)b * C * Tn;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty + i * 8, c = c0 + tx;
        tile[ty + i * 8][tx] = (t < Tn && c < C) ? to_f32(xb[(long)t * C + c]) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + i * 8, t = t0 + tx;
        if (c < C && t < Tn) yb[(long)c * Tn + t] = tile[tx][ty + i * 8];
    }
}

void launch_ncl_to_nlc(const float* x, void* y, int B, int C, int T, int Cpad, int dtype, hipStream_t s) {
    dim3 grid((T + 31) / 32, (Cpad + 31) / 32, B);
    ProfScope ps(FAM_OTHER, s, (double)B * T * (C * 4.0 + Cpad * (double)dtype_size(dtype)), 0);
    if (dtype == MI_F32) hipLaunchKernelGGL(ncl_to_nlc_kernel<float>, grid, dim3(256), 0, s, x, (float*)y, C, T, Cpad);
    else if (dtype == MI_F16) hipLaunchKernelGGL(ncl_to_nlc_kernel<f16>, grid, dim3(256), 0, s, x, (f16*)y, C, T, Cpad);
    else hipLaunchKernelGGL(ncl_to_nlc_kernel<bf16>, grid, dim3(256), 0, s, x, (bf16*)y, C, T, Cpad);
    MI_HIP(hipGetLastError());
}

void launch_nlc_to_ncl(const void* x, float* y, int B, int C, int T, int dtype, hipStream_t s) {
    dim3 grid((T + 31) / 32, (C + 31) / 32, B);
    ProfScope ps(FAM_OTHER, s, (double)B * T * C * (4.0 + (double)dtype_size(dtype)), 0);
    if (dtype == MI_F32) hipLaunchKernelGGL(nlc_to_ncl_kernel<float>, grid, dim3(256), 0, s, (const float*)x, y, C, T);
    else if (dtype == MI_F16) hipLaunchKernelGGL(nlc_to_ncl_kernel<f16>, grid, dim3(256), 0, s, (const f16*)x, y, C, T);
    else hipLaunchKernelGGL(nlc_to_ncl_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)x, y, C, T);
    MI_HIP(hipGetLastError());
}

// conv_post (C -> 1, k=7, pad 3) + tanh|clamp + optional int16 conversion.
//   BigVGAN.forward tail (bigvgan.py:403-408) and BIGVGAN.forward (Export_BigVGAN.py:44-49):
//   y = tanh(conv(x)) ; i16 = trunc(clamp(y*32767, -32768, 32767)).
// 256 outputs per workgroup; the (256+6) x C input span is one contiguous HBM read, staged in LDS
// with an odd row stride so the per-thread row walks are bank-conflict free.
template <typename T>
__global__ __launch_bounds__(256) void conv_post_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                        float bias, int Tn, int C, int use_tanh,
                                                        float* __restrict__ out_f32, int16_t* __restrict__ out_i16) {
    extern __shared__ float lds[];
    const int LD = C | 1;
    float* ws = lds;                  // 7*C
    float* xs = lds + 7 * C;          // 262 * LD
    const int b = blockIdx.y, t0 = blockIdx.x * 256, tid = threadIdx.x;
    const T* xb = x + (long)b * Tn * C;
    for (int i = tid; i < 7 * C; i += 256) ws[i] = w[i];
    const int n = 262 * C;
    for (int i = tid; i < n; i += 256) {
        const int row = i / C, c = i - row * C;
        const int t = t0 - 3 + row;
        xs[row * LD + c] = (t >= 0 && t < Tn) ? to_f32(xb[(long)t * C + c]) : 0.f;
    }
    __syncthreads();
    const int t = t0 + tid;
    if (t >= Tn) return;
    float acc = bias;
    for (int j = 0; j < 7; ++j) {
        const float* xr = xs + (tid + j) * LD;
        const float* wr = ws + j * C;
        for (int c = 0; c < C; ++c) acc = fmaf(xr[c], wr[c], acc);
    }
    float v = use_tanh ? tanhf(acc) : fminf(fmaxf(acc, -1.f), 1.f);
    if (out_f32) out_f32[(long)b * Tn + t] = v;
    if (out_i16) {
        float q = v * 32767.0f;
        if (use_tanh) q = fminf(fmaxf(q, -32768.0f), 32767.0f);
        out_i16[(long)b * Tn + t] = (int16_t)q;     // truncation toward zero == torch .to(int16)
    }
}

void launch_conv_post(const void* x, const float* w, float bias, int B, int T, int C, int dtype, int use_tanh,
                      float* out_f32, int16_t* out_i16, hipStream_t s) {
    dim3 grid((T + 255) / 256, B);
    const size_t lds = (size_t)(7 * C + 262 * (C | 1)) * 4;
    MI_REQUIRE(lds <= 160 * 1024, "conv_post: channel count too large");
    ProfScope ps(FAM_CONV_POST, s, (double)B * T * (C * (double)dtype_size(dtype) + 2.0), 2.0 * B * T * 7.0 * C);
    if (dtype == MI_F32)
        hipLaunchKernelGGL(conv_post_kernel<float>, grid, dim3(256), lds, s, (const float*)x, w, bias, T, C, use_tanh, out_f32, out_i16);
    else if (dtype == MI_F16)
        hipLaunchKernelGGL(conv_post_kernel<f16>, grid, dim3(256), lds, s, (const f16*)x, w, bias, T, C, use_tanh, out_f32, out_i16);
    else
        hipLaunchKernelGGL(conv_post_kernel<bf16>, grid, dim3(256), lds, s, (const bf16*)x, w, bias, T, C, use_tanh, out_f32, out_i16);
    MI_HIP(hipGetLastError());
}

}  // namespace mi
