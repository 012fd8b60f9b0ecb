// f5_kernels.h — launchers of the non-GEMM F5 kernels (f5_kernels.hip, attention.hip).
#pragma once
#include "common.h"

namespace mi {

enum { NORM_LN_MOD = 0, NORM_LN_AFFINE = 1, NORM_L2 = 2 };
// x fp32 [rows][D] -> y (out_dtype) ; LN_MOD: LN(x)*(1+a)+b ; LN_AFFINE: LN(x)*a+b ; L2: a*x/||x||+b
void launch_rownorm(int mode, const float* x, void* y, int out_dtype, const float* a, const float* b, long rows, int D,
                    float eps, hipStream_t s);
// LN_MOD with the result as gemm_x3p.hip panel planes (three-way bf16 split of the fp32 value) instead of fp32 rows
void launch_rownorm_x3p(const float* x, void* planes, const float* a, const float* b, long rows, int D, float eps, hipStream_t s,
                        int np = 3, int* sat = nullptr);       // sat: range watch of the fp16-pair split (x3_split.h)
// AdaLN fold (ConvGemm::ln_*, gemm_epilogue.h): the producer side as a pass of its own, for the first block of an evaluation —
// aout = x o (1 + scale) as panel planes of np planes (a_dtype MI_F32) or rows of a_dtype, stats[row][D / 32][2] = partial (sum, sum^2)
void launch_ln_prologue(const float* x, void* aout, int a_dtype, int np, float* stats, const float* scale, long rows, int D, int* sat,
                        hipStream_t s);
// ... and, for the 16-bit engines, the partials of every row summed once into fin[row] = (rstd, mean * rstd) (ConvGemm::ln_final)
void launch_ln_finalize(const float* partials, float* fin, long rows, int D, float eps, hipStream_t s);
void launch_ln_gather(const float* mod, long mod_ld, long col_scale, long col_shift, float* G, float* S, int steps, int D, hipStream_t s);
void launch_cast_to_f32(const void* src, int dtype, float* dst, long n, hipStream_t s);
void launch_absmax(const float* x, long n, unsigned* out_bits, hipStream_t s);       // atomicMax of the bit pattern of max |x|
void launch_dwconv7(const float* x, float* y, const float* w, const float* bias, int B, int T, int C, hipStream_t s);
void launch_grn(float* y, float* ss_scratch, const float* gamma, const float* beta, int B, int T, int C, hipStream_t s);
// ids [U][N]; out slabs 2u (text) / 2u+1 (drop)
void launch_text_gather(const int* ids, const float* emb, const float* pos, float* out, int U, int N, int C, hipStream_t s);
void launch_text_ids(const int32_t* in, int* out, int U, int T, int N, int vocab, int* err, hipStream_t s);
void launch_mask_rows(const int* ids, float* x, int V, int N, int C, hipStream_t s);
void launch_copy2d(const float* src, long lds_, void* dst, long ldd, long rows, int cols, int out_dtype, hipStream_t s);
void launch_pad_reflect(const int16_t* a, float* out, int U, long L, int half, hipStream_t s);
void launch_spec_mag(const float* spec, float* mag, int F, int nb, int ldm, float eps, hipStream_t s);      // sqrt(re^2 + im^2 + eps)
void launch_logmel(const float* melraw, float* cmt, float* cmtd, int U, int N, int R, int M, int ld, hipStream_t s);
void launch_vocos_head(const float* sp, float* c, long rows, int nb, int ldc, hipStream_t s);
void launch_istft_ola(const float* frames, const float* wsi, int U, int F, int nfft, int hop, float* out_f,
                      int16_t* out_i, hipStream_t s);
void launch_cat_noise(const float* noise, void* cat, int U, int N, int M, int ldc, int dtype, hipStream_t s);
void launch_cfg_update(float* noise, const float* pred, int U, int N, int M, float cfg, const float* dt, int k, hipStream_t s, int parts = 1);
void launch_sum_parts(const float* in, float* out, long rows, int M, int parts, hipStream_t s);

// attention.hip: softmax_fp32(q k^T) v, no mask, no scale (q/k are pre-scaled): modules.py:467
//   q,k [BH][N][64], v [BH][N][64] (fp32, native / q.k-split kernels) or transposed [BH][64][v_ld] (16-bit, and the fp32
//   kernel with both products split; v_ld = attention_v_ld(N, dtype))
//   -> o [B][N][H*64] (dtype), B = BH / H
// ws / cnt: optional workspace of the key-sliced fp32 kernel ((2*32*64 + 256) floats per 64-query tile and slice; one zeroed
// counter per tile); without them every query tile is one workgroup
void launch_attention(const void* q, const void* k, const void* v, void* o, int BH, int H, int N, int dtype, hipStream_t s,
                      float* ws = nullptr, long ws_floats = 0, int* cnt = nullptr, long cnt_n = 0, void* o_planes = nullptr,
                      int kv_planes = 0, int o_np = 3, float ref_fp16_scale = 0.f);
// ref_fp16_scale (f16 engines; 0 = off): the score rounding points of the reference's fp16-transformer export — q k scores
// rounded to fp16, then x ref_fp16_scale (= 100, undoing the extra x0.1 folded into q and k) in fp32 (F5/fp16/modules.py:467)
// kv_planes (fp32 engines, both products split): k and v are the pre-split bf16 planes the QKV epilogue wrote (ConvGemm::kv_planes:
// k [BH][3][ld][64], v [BH][3][64][ld], ld = N rounded up to 64, pad keys of v zero) — ask attention_takes_kv_planes() first
bool attention_takes_kv_planes(int N, int BH, int dtype);
int attention_kv_planes_format();        // 2: fp16 {hi, lo} pairs (option "attn_f32_planes", default) | 3: three bf16 planes — pass it as kv_planes
// o_planes (fp32 engines, both products split): the output as gemm_x3p.hip panel planes of the [B * N][H * 64] matrix instead
// of rows in o — ask attention_can_write_planes() first
bool attention_can_write_planes(int N, int BH, int dtype);
// row length of the transposed V the attention kernel in use expects (0: V untransposed, [BH][N][64]) — ask per launch: the
// fp32 answer follows the attn_f32_x3 option
long attention_v_ld(int N, int dtype);

}  // namespace mi
