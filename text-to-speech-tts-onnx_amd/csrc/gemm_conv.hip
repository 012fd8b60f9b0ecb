// gemm_conv.hip — implicit-GEMM convolution / linear kernel on CDNA4 MFMA (gfx950).
//
// One kernel family serves every dense contraction on the hot path:
//   * BigVGAN Conv1d k in {3,7,11}, dilation {1,3,5}         (bigvgan.py:56-88, 136-139)
//   * BigVGAN ConvTranspose1d stride u, k = 2u  (EPI_CONVT)   (bigvgan.py:300-315, 391)
//   * DiT / Vocos / text-embed Linear layers (taps = 1)       (modules.py:329-340, 459-468)
//   * grouped position-embedding conv k31 g16 (G > 1)         (modules.py:167-190)
//   * STFT / ISTFT / mel as framed GEMMs (x_rstride = hop)     (STFT_Process.py:153-166)
//
// Activations are channels-last (B, T, C) so that the GEMM K axis (tap, ci) is contiguous in
// 16-byte vectors for the MFMA A operand and the N axis (output channel) is the lane axis of the
// 32x32 accumulator tile => 64/128-byte coalesced row stores.
//
//   M = output time steps, N = output channels, K = taps * Cin
//   A[m][kk] = x[b, m - pad + tap*dil, ci]   (zero outside [0, T_in))      kk = tap*Cin + ci
//   B[kk][n] = w[n, kk]
//
// MFMA: f32 -> v_mfma_f32_32x32x2_f32 (exact fp32), f16/bf16 -> v_mfma_f32_32x32x16_{f16,bf16};
// fp32 accumulate always.  4 waves per workgroup, each owning a (WM x WN) sub-tile of 32x32 MFMA
// tiles.  K is streamed in KC-wide chunks: global -> registers (next chunk, issued before the
// MFMAs of the current one) -> LDS -> fragments.
#include <atomic>
#include "common.h"
#include <mutex>
#include "mfma.h"
#include "gemm_epilogue.h"
#include <cstdlib>

namespace mi {

template <typename T, typename TO, int BM, int BN, int WGM, int WGN, int KC>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const ConvGemmDev p) {
    using MF = Mfma<T>;
    constexpr int KP = MF::KP;
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int KV = KC / VEC;          // 16-byte vectors per tile row
    constexpr int RPP = 256 / KV;         // tile rows filled per pass of the 256 threads
    constexpr int AP = (BM + RPP - 1) / RPP, BP = (BN + RPP - 1) / RPP;
    constexpr int LDA = KC + (sizeof(T) == 4 ? 1 : 8);   // +pad: conflict-free fragment reads
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    static_assert(WGM * WGN == 4 && WM % 32 == 0 && WN % 32 == 0, "tile");
    static_assert(KC % (2 * KP) == 0, "KC");

    __shared__ __attribute__((aligned(16))) T smem[(BM + BN) * LDA];
    T* As = smem;
    T* Bs = smem + BM * LDA;

    const int tid = threadIdx.x;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int b = blockIdx.z / p.G, g = blockIdx.z % p.G;
    const T* xb = (const T*)p.x + (long)b * p.x_bstride + (long)g * p.x_goff;
    const T* wg = (const T*)p.w + (long)g * p.N * p.K;

    const int kv = tid % KV, r0 = tid / KV;
    uint4 areg[AP], breg[BP];

    auto load_regs = [&](int kk0) {
        const int kk = kk0 + kv * VEC;
        const bool kvalid = kk < p.K;
        const int tap = kk / p.Cin;
        const int ci = kk - tap * p.Cin;
        const int rowoff = tap * p.dil - p.pad;
#pragma unroll
        for (int q = 0; q < AP; ++q) {
            const int r = r0 + q * RPP;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < BM) {
                const int row = m0 + r + rowoff;
                if (kvalid && row >= 0 && row < p.T_in)
                    v = *reinterpret_cast<const uint4*>(xb + (long)row * p.x_rstride + ci);
            }
            areg[q] = v;
        }
#pragma unroll
        for (int q = 0; q < BP; ++q) {
            const int r = r0 + q * RPP;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < BN) {
                const int n = n0 + r;
                if (kvalid && n < p.N) v = *reinterpret_cast<const uint4*>(wg + (long)n * p.K + kk);
            }
            breg[q] = v;
        }
    };
    auto store_lds = [&]() {
#pragma unroll
        for (int q = 0; q < AP; ++q) {
            const int r = r0 + q * RPP;
            if (r < BM) {
                if constexpr (sizeof(T) == 4) {
                    float* d = reinterpret_cast<float*>(As) + r * LDA + kv * VEC;
                    d[0] = __uint_as_float(areg[q].x); d[1] = __uint_as_float(areg[q].y);
                    d[2] = __uint_as_float(areg[q].z); d[3] = __uint_as_float(areg[q].w);
                } else {
                    *reinterpret_cast<uint4*>(As + r * LDA + kv * VEC) = areg[q];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < BP; ++q) {
            const int r = r0 + q * RPP;
            if (r < BN) {
                if constexpr (sizeof(T) == 4) {
                    float* d = reinterpret_cast<float*>(Bs) + r * LDA + kv * VEC;
                    d[0] = __uint_as_float(breg[q].x); d[1] = __uint_as_float(breg[q].y);
                    d[2] = __uint_as_float(breg[q].z); d[3] = __uint_as_float(breg[q].w);
                } else {
                    *reinterpret_cast<uint4*>(Bs + r * LDA + kv * VEC) = breg[q];
                }
            }
        }
    };

    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WGN, wn = wave % WGN;
    const int lr = lane & 31, lk = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_regs(0);
    store_lds();
    __syncthreads();
    for (int kk0 = 0; kk0 < p.K; kk0 += KC) {
        const bool more = kk0 + KC < p.K;
        if (more) load_regs(kk0 + KC);
#pragma unroll
        for (int ks = 0; ks < KC / (2 * KP); ++ks) {
            typename MF::Frag a[TM], bb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const typename MF::Frag*>(As + (wm * WM + i * 32 + lr) * LDA + ks * 2 * KP + lk * KP);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bb[j] = *reinterpret_cast<const typename MF::Frag*>(Bs + (wn * WN + j * 32 + lr) * LDA + ks * 2 * KP + lk * KP);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma(a[i], bb[j], acc[i][j]);
        }
        __syncthreads();
        if (more) {
            store_lds();
            __syncthreads();
        }
    }

    gemm_epilogue<TO, TM, TN, WM, WN>(acc, p, m0, n0, b, g, wm, wn, lr, lk);
}

// ---------------------------------------------------------------------------------------------------
// 16-bit main loop v2: HBM -> LDS by direct DMA (global_load_lds, 16 B per lane), two LDS buffers,
// one barrier per 64-deep K chunk, XOR-swizzled 128-byte tile rows.
//
//   * a K chunk is (tap, 64 input channels): A rows are x[m + tap*dil - pad][c0 .. c0+64), B rows are
//     w[n][tap*Cin + c0 .. +64).  Vectors that fall outside the tensor (time padding, Cin tail, N tail) are
//     fetched from a page of zeros, so the DMA needs no predication and the MFMA loop no masking.
//   * LDS image: row r holds its eight 16-byte k-vectors at slot (kv ^ ((r >> 1) & 7)).  The DMA writes lane-linear
//     (lane l -> row R0 + l/8, slot l%8), so lane l simply FETCHES the k-vector its slot must hold; fragment reads apply
//     the same XOR.  Every ds_read_b128 lane group then covers all 64 banks exactly once (conflict-free).
//   * the loads of chunk c+1 are in flight while the 16 MFMAs per wave of chunk c run.
// ---------------------------------------------------------------------------------------------------
// PAIRS (fp32 only): the operands are split into fp16 {hi, lo * 2^11} pairs IN REGISTERS in front of the matrix cores (x3_split.h)
// and every 16-deep k-step is three v_mfma_f32_32x32x16_f16 on two accumulator sets instead of eight v_mfma_f32_32x32x2_f32 —
// the arithmetic of gemm_x3p.hip for shapes its panel planes do not cover (the grouped k = 31 position convolution of the
// DiT, N = 64 per group): 512 matrix-core cycles per chunk become 96, the split costs ~290 VALU cycles.
template <typename T, typename TO, bool LEPI, int NST = 2, int BT = 128, bool PAIRS = false>
__global__ __launch_bounds__(256, 2) void conv_gemm_dma_kernel(const ConvGemmDev p) {
    using MF = Mfma<T>;
    static_assert(!PAIRS || (sizeof(T) == 4 && NST == 2), "PAIRS: the fp32 two-buffer loop");
    // a tile row is always 128 bytes = eight 16-byte k-vectors: 64 halfs / bf16s or 32 floats per K chunk
    constexpr int VEC = 16 / (int)sizeof(T), KC = 8 * VEC;
    // BT = 128: the 128x128 tile (64x64 per wave).  BT = 64: 64x64 tiles (32x32 per wave) for fp32 problems with too few
    // 128-row tiles to balance 256 CUs — fp32 MFMAs are slow enough (64 cycles) that the doubled fragment traffic is free
    constexpr int BM = BT, BN = BT, WM = BT / 2, WN = BT / 2, TM = WM / 32, TN = WN / 32;
    constexpr int DJ = BT / 32;                           // 8-row DMA groups per wave per operand per chunk
    static_assert(BT == 128 || BT == 64, "tile");
    constexpr int TILE = (BM + BN) * KC;                  // elements per buffer
    static_assert(NST == 2 || NST == 4, "two buffers, or the four-stage ring for one-round launches");
    __shared__ __attribute__((aligned(1024))) T smem[NST * TILE];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lk = lane >> 5;
    // XCD-aware tile order.  Workgroup L lands on XCD (L % 8) (observed dispatch order; only speed depends on
    // it).  Each XCD owns a contiguous range of RC row tiles (batch x M) and walks the N tiles in the OUTER loop,
    // so the RC workgroups that run together share one weight panel in their XCD's 4 MiB L2 and the XCD's
    // activation rows stay L2-resident across N tiles, instead of every L2 streaming every panel from MALL.
    const int L = blockIdx.x;
    int nt, rowt;
    if (p.RC > 0) {
        const int xcd = L & 7, jx = L >> 3;
        nt = jx / p.RC; rowt = xcd * p.RC + (jx - nt * p.RC);
        if (rowt >= p.RT) return;                         // padding workgroups (RT rounded up to 8*RC)
    } else {                                              // plain order: row tiles fastest
        nt = L / p.RT; rowt = L - nt * p.RT;
    }
    const int b = rowt / p.Tm, mt = rowt - b * p.Tm;
    const int m0 = mt * BM, n0 = nt * BN;
    const int g = blockIdx.y;
    const T* xb = (const T*)p.x + (long)b * p.x_bstride + (long)g * p.x_goff;
    const T* wg = (const T*)p.w + (long)g * p.N * p.K;
    const T* zero = (const T*)p.zero;

    // logical k-vector this lane fetches: physical slot (lane & 7) of row R0 + lrow holds k-vector slot ^ ((row >> 1) & 7).
    // (row >> 1, not row: the hardware serves a ds_read_b128 in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...;
    // with `row & 7` rows 0 / 24, 1 / 25, ... of a group share a slot: SQ_LDS_BANK_CONFLICT was 47 % of the LDS-active
    // cycles of this kernel (0 % in the 256-row kernels, which always used the `row >> 1` form).)
    const int kvl0 = (lane & 7) ^ ((lane >> 4) & 7);              // even 8-row DMA groups
    const int kvl1 = (lane & 7) ^ ((4 + (lane >> 4)) & 7);        // odd 8-row DMA groups
    const int lrow = lane >> 3;                           // row inside an 8-row DMA group
    const int cpt = (p.Cin + KC - 1) / KC;                // chunks per tap
    const int nchunks = (p.K / p.Cin) * cpt;

    // p.use_buf: DMA through buffer descriptors — out-of-range lanes get zeros from the hardware range check, the per-lane
    // offsets are loop-invariant (see conv_gemm_dma3_kernel); the flat-address form below stays for Cin tails.
#if defined(__HIP_DEVICE_COMPILE__)
    __amdgpu_buffer_rsrc_t rsa, rsb;
    int avo[DJ], bvo[DJ];
    if (p.use_buf) {
        rsa = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)((((long)p.T_in - 1) * p.x_rstride + p.Cin) * (long)sizeof(T)), 0x00020000);
        rsb = __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, (int)((long)p.N * p.K * (long)sizeof(T)), 0x00020000);
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            const int R0 = (wave * DJ + j) * 8;
            const long n = n0 + R0 + lrow;
            const int kvl = (j & 1) ? kvl1 : kvl0;                 // (wave * DJ + j) & 1 == j & 1: DJ is even
            avo[j] = (int)(((long)(m0 + R0 + lrow) * p.x_rstride + kvl * VEC) * (long)sizeof(T));
            bvo[j] = n < p.N ? (int)((n * p.K + kvl * VEC) * (long)sizeof(T)) : 0x7fffff00;
        }
    }
#endif
    auto issue = [&](int buf, int tap, int c0) {
        T* base = smem + buf * TILE;
#if defined(__HIP_DEVICE_COMPILE__)
        if (p.use_buf) {
            const int ca = (int)(((long)(tap * p.dil - p.pad) * p.x_rstride + c0) * (long)sizeof(T));
            const int cb = (tap * p.Cin + c0) * (int)sizeof(T);
#pragma unroll
            for (int j = 0; j < DJ; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_void*)(base + (wave * DJ + j) * 8 * KC), 16, avo[j] + ca, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < DJ; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lds_void*)(base + (BM + (wave * DJ + j) * 8) * KC), 16, bvo[j] + cb, 0, 0, 0);
            return;
        }
#endif
        const int toff = tap * p.dil - p.pad;
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            const int R0 = (wave * DJ + j) * 8;
            const int ci = c0 + ((j & 1) ? kvl1 : kvl0) * VEC;
            const int t = m0 + R0 + lrow + toff;
            const T* src = (ci < p.Cin && t >= 0 && t < p.T_in) ? xb + (long)t * p.x_rstride + ci : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (lds_void*)(base + R0 * KC), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            const int R0 = (wave * DJ + j) * 8;
            const int ci = c0 + ((j & 1) ? kvl1 : kvl0) * VEC;
            const int n = n0 + R0 + lrow;
            const T* src = (ci < p.Cin && n < p.N) ? wg + (long)n * p.K + (long)tap * p.Cin + ci : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (lds_void*)(base + (BM + R0) * KC), 16, 0, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x16 accb[PAIRS ? TM : 1][PAIRS ? TN : 1];          // PAIRS: the 2^11-scaled cross terms
    if constexpr (PAIRS) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accb[i][j][r] = 0.f;
    }
    int tap = 0, c0 = 0;
    const int ntaps = p.K / p.Cin;
    auto compute = [&](int buf) {
        const T* As = smem + buf * TILE;
        const T* Bs = As + BM * KC;
        if constexpr (PAIRS) {
            using MH = Mfma<f16>;
            using FH = typename MH::Frag;
            // k-step s of the 32-float chunk: lane half lk holds k = 16 s + 8 lk .. + 8 = the k-vectors 4 s + 2 lk and + 1 of its row
            // (A and B alike, so the contraction pairs up); finite values beyond the fp16 range saturate
            auto split8 = [&](const float4 u, const float4 w, FH& fh, FH& fl, bool clamp) __attribute__((always_inline)) {
                float v[8] = {u.x, u.y, u.z, u.w, w.x, w.y, w.z, w.w};
                unsigned h[4], l[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float a = v[2 * q], b2 = v[2 * q + 1];
                    if (clamp) { a = __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f); b2 = __builtin_amdgcn_fmed3f(b2, -65504.f, 65504.f); }
                    x2_split_pair_raw(a, b2, h[q], l[q]);
                }
                fh = __builtin_bit_cast(FH, x3_u4{h[0], h[1], h[2], h[3]});
                fl = __builtin_bit_cast(FH, x3_u4{l[0], l[1], l[2], l[3]});
            };
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                FH ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = wm * WM + i * 32 + lr, sw = (row >> 1) & 7, kv = 4 * st + 2 * lk;
                    split8(*reinterpret_cast<const float4*>(As + row * KC + ((kv ^ sw) * VEC)),
                           *reinterpret_cast<const float4*>(As + row * KC + (((kv + 1) ^ sw) * VEC)), ah[i], al[i], true);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = wn * WN + j * 32 + lr, sw = (row >> 1) & 7, kv = 4 * st + 2 * lk;
                    split8(*reinterpret_cast<const float4*>(Bs + row * KC + ((kv ^ sw) * VEC)),
                           *reinterpret_cast<const float4*>(Bs + row * KC + (((kv + 1) ^ sw) * VEC)), bh[j], bl[j], false);     // weights: finite, in range
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        accb[i][j] = MH::mma(al[i], bh[j], accb[i][j]);
                        accb[i][j] = MH::mma(ah[i], bl[j], accb[i][j]);
                        acc[i][j] = MH::mma(ah[i], bh[j], acc[i][j]);
                    }
            }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if constexpr (sizeof(T) == 2) {
                typename MF::Frag a[TM], bb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = wm * WM + i * 32 + lr;
                    a[i] = *reinterpret_cast<const typename MF::Frag*>(As + row * KC + (((ks * 2 + lk) ^ ((row >> 1) & 7)) * VEC));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = wn * WN + j * 32 + lr;
                    bb[j] = *reinterpret_cast<const typename MF::Frag*>(Bs + row * KC + (((ks * 2 + lk) ^ ((row >> 1) & 7)) * VEC));
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma(a[i], bb[j], acc[i][j]);
            } else {
                // fp32 (v_mfma_f32_32x32x2_f32 takes ONE float per lane and contracts k = {lk = 0, lk = 1}): every lane
                // reads the whole 16-byte k-vector (2*ks + lk) of its row and feeds its four floats to four MFMAs, i.e.
                // instruction e contracts k = 8*ks + e and 8*ks + 4 + e.  A and B use the same pairing, so only the
                // fp32 summation order inside the chunk differs from the k-sequential form.
                float4 a[TM], bb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = wm * WM + i * 32 + lr;
                    a[i] = *reinterpret_cast<const float4*>(As + row * KC + (((ks * 2 + lk) ^ ((row >> 1) & 7)) * VEC));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = wn * WN + j * 32 + lr;
                    bb[j] = *reinterpret_cast<const float4*>(Bs + row * KC + (((ks * 2 + lk) ^ ((row >> 1) & 7)) * VEC));
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = MF::mma(a[i].x, bb[j].x, acc[i][j]);
                        acc[i][j] = MF::mma(a[i].y, bb[j].y, acc[i][j]);
                        acc[i][j] = MF::mma(a[i].z, bb[j].z, acc[i][j]);
                        acc[i][j] = MF::mma(a[i].w, bb[j].w, acc[i][j]);
                    }
            }
        }
    };
    if constexpr (NST == 2 && sizeof(T) == 2) {
        // 16-bit, two stages: fragments one k-step ahead in two register sets, and the chunk boundary software-pipelined
        // as in conv_gemm_dma3_kernel — the wait + barrier that publish chunk c+1 sit before the LAST k-step of chunk c,
        // so the first fragments of chunk c+1 and the DMA of chunk c+2 are issued under MFMAs
        typename MF::Frag fa[2][TM], fb[2][TN];
        auto ldfrag = [&](int buf, int ks, int set) {
            const T* As = smem + buf * TILE;
            const T* Bs = As + BM * KC;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * WM + i * 32 + lr;
                fa[set][i] = *reinterpret_cast<const typename MF::Frag*>(As + row * KC + (((ks * 2 + lk) ^ ((row >> 1) & 7)) * VEC));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * WN + j * 32 + lr;
                fb[set][j] = *reinterpret_cast<const typename MF::Frag*>(Bs + row * KC + (((ks * 2 + lk) ^ ((row >> 1) & 7)) * VEC));
            }
        };
        auto mmas = [&](int set) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma(fa[set][i], fb[set][j], acc[i][j]);
        };
        auto advance = [&]() { if (++tap >= ntaps) { tap = 0; c0 += KC; } };
        issue(0, tap, c0); advance();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (nchunks > 1) { issue(1, tap, c0); advance(); }
        ldfrag(0, 0, 0);
        for (int c = 0; c < nchunks; ++c) {
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                ldfrag(c & 1, ks + 1, (ks + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                mmas(ks & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (c + 1 < nchunks) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // chunk c+1 has landed (this wave's share)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // every read of chunk c's stage has retired
                __builtin_amdgcn_s_barrier();
                if (c + 2 < nchunks) { issue(c & 1, tap, c0); advance(); }
                ldfrag((c + 1) & 1, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mmas(1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (LEPI) {                                      // the staging epilogue reuses the stages: all reads must be done
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    } else if constexpr (NST == 2) {
        issue(0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            const int buf = c & 1;
            int ntap = tap + 1, nc0 = c0;                  // K order = (channel chunk, tap): see launch_conv_gemm
            if (ntap >= ntaps) { ntap = 0; nc0 += KC; }
            if (c + 1 < nchunks) issue(buf ^ 1, ntap, nc0);
            compute(buf);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            tap = ntap; c0 = nc0;
        }
    } else {
        // One-round launches (<= one workgroup per CU, e.g. the DiT O / FF2 projections of a single utterance: 144 tiles):
        // nothing else runs on the CU, so the two-buffer loop above exposes a full L2 / HBM round trip per 64-deep chunk
        // (measured ~1700 cycles per chunk against 512 cycles of MFMA).  Here the whole LDS is one ring of four stages:
        // chunk c+3 is issued before chunk c is computed, the wait is the COUNTED s_waitcnt vmcnt(16) (= this wave's 2 x 8
        // DMA instructions of chunks c+2, c+3 may stay in flight) and the barrier is the raw s_barrier, so three chunks
        // are in flight at any time.  Chunks past the end of K fetch the zero page (c0 >= Cin) to keep the count uniform.
        auto advance = [&]() { if (++tap >= ntaps) { tap = 0; c0 += KC; } };
        issue(0, tap, c0); advance();
        issue(1, tap, c0); advance();
        issue(2, tap, c0); advance();
        static_assert(BT == 128, "the ring's vmcnt(16) assumes eight DMA instructions per wave per chunk");
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int c = 0; c < nchunks; ++c) {
            issue((c + 3) & 3, tap, c0); advance();         // overwrites the stage read in iteration c-1 (all waves passed its barrier)
            compute(c & 3);
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // chunk c+1 has landed (this wave's share)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the trailing zero-page chunks must not land on the epilogue's staging
        __builtin_amdgcn_s_barrier();
    }
    if constexpr (PAIRS) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = __builtin_fmaf(accb[i][j][r], 0x1p-11f, acc[i][j][r]);
    }
    if constexpr (LEPI) {
        constexpr int ERT = (WN <= 64 && TM % 2 == 0) ? 2 : 1;
        float* stage = reinterpret_cast<float*>(smem) + wave * (ERT * 32 * WN);
        if constexpr (sizeof(TO) == 2 && BT == 128) {
            if (p.epi == EPI_QKV_ROPE) {
                if (p.ln_stats_in) gemm_epilogue_qkv_lds<TO, 2, true>(acc, p, m0, n0, b, wm, wn, lr, lk, stage);
                else gemm_epilogue_qkv_lds<TO>(acc, p, m0, n0, b, wm, wn, lr, lk, stage);
                return;
            }
        }
        if constexpr (sizeof(T) == 2) {        // DiT linear layers of the 16-bit engines: the AdaLN fold (one batch item, one group: host-checked)
            if constexpr (sizeof(TO) == 2) {
                if (p.ln_stats_in) { gemm_epilogue_ln_in<TO, TM, TN, 2>(acc, p, m0 + wm * WM, n0 + wn * WN, lr, lk, stage); return; }
            } else {
                if (p.ln_stats_out) { gemm_epilogue_resid_ln<T, TM, TN, 2>(acc, p, m0 + wm * WM, n0 + wn * WN, lr, lk, stage); return; }
            }
        }
        gemm_epilogue_lds<TO, TM, TN, WM, WN>(acc, p, m0, n0, b, g, wm, wn, lr, lk, stage);
    } else {
        gemm_epilogue<TO, TM, TN, WM, WN>(acc, p, m0, n0, b, g, wm, wn, lr, lk);
    }
}

static std::atomic<bool> g_use_dma = true, g_xcd_order = false, g_use_dma3 = true, g_big_tiles = true;
static std::atomic<long> g_big_min = 160, g_n192_min = 160, g_mid_min = 160, g_k_min = 2048;
static std::atomic<bool> g_n192 = true, g_f32_dma = true, g_ring4 = true, g_f32_small = true;
static std::atomic<long> g_f32_small_max = 1024, g_small16_max = 256, g_f32_n64_dma = 1, g_n64_dma16 = 0;      // 16-bit: neutral (455 vs 457 ms at 8 utterances), off
static std::atomic<long> g_ring4_max = 256;
// stream-K (gemm_sk.hip): 0 off ; 1 fp32 linear layers ; 2 also 16-bit ; g_sk_stages: ring depth override (0 = automatic) ;
// g_sk_max_tiles: only launches with at most this many 128x128 tiles (beyond that one tile per workgroup balances by itself)
static std::atomic<long> g_sk = 1, g_sk_stages = 0, g_sk_max_tiles = 2048, g_sk_order = -1;
static std::atomic<long> g_sk_min_tiles = 64;      // plain stream-K linear layers: at least this many 128x128 tiles ("gemm_sk_min_tiles")
// whole tiles first, stream-K for the remainder only: measured SLOWER on the fp32 DiT layers (FF1 / FF2, 288 tiles: 87.0 vs
// 83.7 us per launch — the remainder's eight-piece fix-ups cost more than the aligned K walk of the first phase gains): opt-in
// fp32 linear layers as six exact bf16 x bf16 partial products (gemm_x3.hip) when the caller supplies the weight planes
static std::atomic<long> g_x3 = 1;
// ... with both operands as panel planes (gemm_x3p.hip, round 3) when the caller supplies them
static std::atomic<long> g_x3p = 1;
static std::atomic<long> g_f32_gconv = 1;         // ... and, when the shape allows, with each operand split once per workgroup (gconv_pairs.hip)
bool launch_gconv_pairs(const ConvGemm& p, hipStream_t s);
void gconv_pairs_set_option(long v);
bool launch_gconv16(const ConvGemm& p, hipStream_t s);
void gconv16_set_option(long v);
static std::atomic<long> g_f32_n64_pairs = 1;     // fp32 N = 64 convolutions with >= 8 taps: fp16 pairs split in registers (conv_gemm_dma_kernel PAIRS)
// number format of the panel planes built from now on: 3 = three bf16 planes (six products), 2 = fp16 {hi, lo} planes (three products)
static std::atomic<long> g_x3p_np = 0;          // 0: not set by mi_set_option -> MI355TTS_F32_PLANES, else 2
// The engine's arithmetic (ArithScope, common.h) overrides the process-wide options for the calling thread
ArithOverride& arith_tls() { static thread_local ArithOverride a; return a; }
ArithOverride arith_for(int kind) {
    ArithOverride a;
    if (kind == ARITH_NATIVE) { a.gemm_x3 = 0; a.gemm_x3p = 0; a.n64_pairs = 0; a.gconv = 0; a.attn_x3 = 0; }
    else if (kind == ARITH_PAIRS) { a.gemm_x3 = 1; a.gemm_x3p = 1; a.planes = 2; a.n64_pairs = 1; a.gconv = 1; a.attn_x3 = 2; a.attn_np = 2; }
    else if (kind == ARITH_BF16X3) { a.gemm_x3 = 1; a.gemm_x3p = 1; a.planes = 3; a.n64_pairs = 0; a.gconv = 0; a.attn_x3 = 2; a.attn_np = 3; }
    return a;
}
static inline long opt_x3() { const int o = arith_tls().gemm_x3; return o >= 0 ? o : (long)g_x3; }
static inline long opt_x3p() { const int o = arith_tls().gemm_x3p; return o >= 0 ? o : (long)g_x3p; }
static inline long opt_n64_pairs() { const int o = arith_tls().n64_pairs; return o >= 0 ? o : (long)g_f32_n64_pairs; }
static inline long opt_gconv() { const int o = arith_tls().gconv; return o >= 0 ? o : (long)g_f32_gconv; }
bool gemm_x3_enabled() { return opt_x3() != 0; }
int x3p_planes() {
    const int o = arith_tls().planes;
    if (o == 2 || o == 3) return o;
    if (g_x3p_np == 2 || g_x3p_np == 3) return (int)g_x3p_np;
    static const int env = [] { const char* e = std::getenv("MI355TTS_F32_PLANES"); return e ? std::atoi(e) : 0; }();
    return env == 3 ? 3 : 2;
}
bool gemm_x3p_enabled() { return opt_x3() != 0 && opt_x3p() != 0; }
// fp32 QKV + RoPE: its scatter epilogue is slow and in a persistent launch every workgroup runs it at the same time at the
// end (in-model 184 us against 138 us for the 64x64 tiles, whose epilogues overlap other workgroups' main loops): off
static std::atomic<long> g_sk_qkv32 = 0;
// 256x256 eight-phase kernel (gemm_ph8.hip) for 16-bit linear layers with at least g_ph8_min_tiles tiles of 256x256
static std::atomic<long> g_ph8 = 1, g_ph8_min_tiles = 200, g_ph8_order = 1;
static std::atomic<long> g_row_split = 1;          // rows beyond the last whole round of 256x256 tiles as a second launch (launch_conv_gemm)
static DevBuf g_zero_page[16];

// buffer-descriptor DMA (BUF kernels): whole 64-deep chunks only, and every byte offset must fit the 32-bit range check
static std::atomic<bool> g_buf = true;
static bool buf_ok(const ConvGemmDev& d, int esz = 2) {
    const long a_bytes = (((long)d.T_in - 1) * d.x_rstride + d.Cin) * esz, b_bytes = (long)d.N * d.K * esz;
    return g_buf && d.Cin % (128 / esz) == 0 && a_bytes + (long)512 * d.x_rstride * esz < 0x7fff0000L && b_bytes < 0x7fff0000L;
}

template <typename T, typename TO>
static void dispatch_tiles(const ConvGemmDev& d, int B, hipStream_t s) {
    constexpr int KC = sizeof(T) == 4 ? 16 : 64;
    dim3 blk(256);
    // the AdaLN fold exists in the epilogues of linear_x3p / linear_ph8 / the 16-bit conv_gemm_dma_kernel only: anything else refuses
    auto no_fold = [&] { MI_REQUIRE(!d.ln_stats_in && !d.ln_stats_out, "conv_gemm: this launch lands on a kernel without the AdaLN fold epilogues (gemm_ln_fold_ok disagrees with the dispatch)"); };
    if (d.N <= 32) {
        no_fold();
        dim3 grid((d.M + 255) / 256, (d.N + 31) / 32, B * d.G);
        MI_LAUNCH((conv_gemm_kernel<T, TO, 256, 32, 4, 1, KC>), T, TO, grid, blk, 0, s, d);
    } else if (d.N <= 64) {
        no_fold();
        {
            // N = 64 per group (the DiT position convolution: k = 31, 16 groups of 64 channels): 64x64 tiles of the LDS-DMA kernel
            // instead of the register-staged 128x64 one (fp32: 186 -> 120 us per launch)
            constexpr int VEC_ = 16 / (int)sizeof(T);
            const long lim = sizeof(T) == 4 ? g_f32_n64_dma : g_n64_dma16;
            if (lim && d.N == 64 && g_use_dma && (sizeof(T) != 4 || g_f32_dma) && d.Cin % VEC_ == 0 && d.K % d.Cin == 0) {
                ConvGemmDev e = d;
                e.Tm = (d.M + 63) / 64; e.Tn = 1; e.RT = B * e.Tm; e.RC = 0;
                e.use_buf = buf_ok(d, (int)sizeof(T));
                dim3 g2(e.RT * e.Tn, d.G);
                if constexpr (sizeof(T) == 4) {
                    // fp32 with many taps (the DiT's grouped k = 31 position convolution): operands as fp16 pairs split in registers
                    if (opt_x3() != 0 && opt_n64_pairs() != 0 && d.K / d.Cin >= 8 && d.Cin % 32 == 0) {
                        if (e.lds_epi) { prof_set_kernel("conv_gemm_dma_kernel<float, float, true, 2, 64, fp16 pairs>", "", ""); hipLaunchKernelGGL((conv_gemm_dma_kernel<T, TO, true, 2, 64, true>), g2, blk, 0, s, e); }
                        else { prof_set_kernel("conv_gemm_dma_kernel<float, float, false, 2, 64, fp16 pairs>", "", ""); hipLaunchKernelGGL((conv_gemm_dma_kernel<T, TO, false, 2, 64, true>), g2, blk, 0, s, e); }
                        MI_HIP(hipGetLastError());
                        return;
                    }
                }
                if (e.lds_epi) MI_LAUNCH((conv_gemm_dma_kernel<T, TO, true, 2, 64>), T, TO, g2, blk, 0, s, e);
                else MI_LAUNCH((conv_gemm_dma_kernel<T, TO, false, 2, 64>), T, TO, g2, blk, 0, s, e);
                MI_HIP(hipGetLastError());
                return;
            }
        }
        dim3 grid((d.M + 127) / 128, (d.N + 63) / 64, B * d.G);
        MI_LAUNCH((conv_gemm_kernel<T, TO, 128, 64, 2, 2, KC>), T, TO, grid, blk, 0, s, d);
    } else {
        dim3 grid((d.M + 127) / 128, (d.N + 127) / 128, B * d.G);
        // a launch that carries the AdaLN fold (gemm_ln_fold_ok said yes) skips the 16-bit kernels without fold epilogues (dma3<192 |
        // 256 | 128>, 16-bit stream-K) instead of refusing in the middle of an inference: e.g. dim = 1152 at ~4 utterances would
        // otherwise land O / FF2 on dma3<192>; the fold kernels (ph8, conv_gemm_dma) take every shape those take
        const bool fold = d.ln_stats_in || d.ln_stats_out;
        if constexpr (sizeof(T) == 2) {
            // many row tiles (a batch of utterances): the 8-wave 256x256 eight-phase main loop
            const long tiles256 = (long)((d.M + 255) / 256) * ((d.N + 255) / 256);
            if (g_ph8 && B == 1 && d.G == 1 && d.K == d.Cin && d.Cin % 64 == 0 && d.pad == 0 && d.N % 64 == 0 && buf_ok(d, 2) &&
                d.lds_epi && (d.epi == EPI_PLAIN || d.epi == EPI_QKV_ROPE) && tiles256 >= g_ph8_min_tiles) {
                ConvGemmDev e = d;
                e.Tm = (d.M + 255) / 256; e.Tn = (d.N + 255) / 256; e.RT = e.Tm; e.RC = (int)g_ph8_order;
                launch_linear_ph8<T, TO>(e, s);
                return;
            }
        }
        if constexpr (sizeof(T) == 4) {
            const long tiles = (long)((d.M + 127) / 128) * ((d.N + 127) / 128);
            if (d.xp && d.w3p && B == 1) {                      // both kept only when x3p_eligible() said yes (launch_conv_gemm)
                ConvGemmDev e = d;
                e.x = d.xp; e.w3 = d.w3p;
                launch_linear_x3p(e, s);
                return;
            }
            if (d.w3 && B == 1 && tiles >= 64) {                // d.w3 is only kept when x3_eligible() said yes (launch_conv_gemm)
                ConvGemmDev e = d;
                e.Tm = (d.M + 127) / 128; e.Tn = (d.N + 127) / 128; e.RT = e.Tm;
                e.RC = 0;       // row tiles fastest: neighbouring ranges share the weight planes, the heavier operand here (24 of the 40 KB per chunk): 62.9 -> 61.0 us
                if (g_sk_order >= 0) e.RC = (int)g_sk_order;
                no_fold();
                launch_linear_x3(e, s);
                return;
            }
        }
        {
            // plain linear layer (one tap, one group, one M axis, whole K chunks): stream-K over persistent workgroups
            constexpr int KCB = 128 / (int)sizeof(T);
            const long tiles = (long)((d.M + 127) / 128) * ((d.N + 127) / 128);
            if (!(sizeof(T) == 2 && fold) && g_sk >= (sizeof(T) == 4 ? 1 : 2) && d.sk_ws && d.sk_slots >= 256 && B == 1 && d.G == 1 && d.K == d.Cin && d.Cin % KCB == 0 &&
                (d.epi == EPI_PLAIN || (d.epi == EPI_QKV_ROPE && (sizeof(T) == 2 || g_sk_qkv32))) && buf_ok(d, (int)sizeof(T)) && d.M > 128 && tiles >= g_sk_min_tiles && tiles <= g_sk_max_tiles && d.pad == 0) {
                ConvGemmDev e = d;
                e.Tm = (d.M + 127) / 128; e.Tn = (d.N + 127) / 128; e.RT = e.Tm;
                e.RC = d.M <= d.N ? 0 : 1;      // per-XCD groups: whole weight panels (x re-read 8x) when x is the smaller operand, else whole row tiles
                if (g_sk_order >= 0) e.RC = (int)g_sk_order;
                e.tail_tiles = 0;
                no_fold();
                launch_linear_sk<T, TO>(e, (int)g_sk_stages, s);
                return;
            }
        }
        if constexpr (sizeof(T) == 2) {
            // measured on gfx950 (tools/gemm_bench.py): with <= 32 K-chunks the 4-wave 128x128 kernel (two workgroups
            // per CU, short prologue/epilogue) wins; deeper K favours the 8-wave 256-row tiles.
            // N divisible by both 192 and 256 (BigVGAN stage 0, N = 768 at B = 8: 192 tiles of 256x256 leave a quarter of
            // the CUs idle for the whole launch, 256 tiles of 256x192 fill the chip): compare rounds x tile width
            const long rt256 = (long)B * ((d.M + 255) / 256);
            const long rounds192 = (rt256 * (d.N / 192) + 255) / 256, rounds256 = (rt256 * ((d.N + 255) / 256) + 255) / 256;
            const bool n192_wins = d.N % 256 != 0 || (d.K > g_k_min && rounds192 * 192 < rounds256 * 256);
            if (!fold && g_use_dma3 && g_n192 && buf_ok(d) && d.K % d.Cin == 0 && d.M > 128 && d.N % 192 == 0 && n192_wins &&
                d.K >= 576 && (long)B * ((d.M + 255) / 256) * (d.N / 192) >= g_n192_min) {
                // N = 192 / 384 (BigVGAN stages 2 and 1): a 192-wide tile has no padded columns (128-wide tiles waste 25 %
                // of the MFMAs and DMA bytes at N = 192) and the fewest DMA bytes per useful flop after 256x256
                ConvGemmDev e = d;
                e.RC = 0; e.Tm = (d.M + 255) / 256; e.Tn = d.N / 192; e.RT = B * e.Tm;
                no_fold();
                launch_conv_gemm_dma3<T, TO>(e, 192, s);
                MI_HIP(hipGetLastError());
                return;
            }
            if (!fold && g_use_dma3 && d.Cin % 8 == 0 && d.K % d.Cin == 0 && d.M > 128 && d.K > g_k_min) {
                ConvGemmDev e = d;
                e.RC = 0;
                const long blocks_128 = (long)B * ((d.M + 127) / 128) * ((d.N + 127) / 128);
                const long blocks_256x128 = (long)B * ((d.M + 255) / 256) * ((d.N + 127) / 128);
                const long blocks_256x256 = (long)B * ((d.M + 255) / 256) * ((d.N + 255) / 256);
                const int n_waste_256 = ((d.N + 255) / 256) * 256 - d.N;
                if (g_big_tiles && buf_ok(d) && blocks_256x256 >= g_big_min && n_waste_256 * 4 <= d.N) {
                    // every CU busy for >= 2 rounds: the tile with the fewest DMA bytes per flop
                    e.Tm = (d.M + 255) / 256; e.Tn = (d.N + 255) / 256; e.RT = B * e.Tm;
                    no_fold();
                    launch_conv_gemm_dma3<T, TO>(e, 256, s);
                    MI_HIP(hipGetLastError());
                    return;
                }
                if (blocks_256x128 >= g_mid_min || blocks_128 < blocks_256x128 + 32) {
                    e.Tm = (d.M + 255) / 256; e.Tn = (d.N + 127) / 128; e.RT = B * e.Tm;
                    no_fold();
                    launch_conv_gemm_dma3<T, TO>(e, 128, s);
                    MI_HIP(hipGetLastError());
                    return;
                }
                // few tiles: fall through to the 128x128 kernel so that more CUs pull data
            }
            if (g_use_dma && d.Cin % 8 == 0 && d.K % d.Cin == 0) {
                ConvGemmDev e = d;
                e.Tm = (d.M + 127) / 128; e.Tn = (d.N + 127) / 128; e.RT = B * e.Tm;
                e.RC = g_xcd_order ? (e.RT + 7) / 8 : 0;
                e.use_buf = buf_ok(d, 2);
                dim3 g1(e.RC > 0 ? 8 * e.RC * e.Tn : e.RT * e.Tn, d.G);
                const int nchunks = (d.K / d.Cin) * ((d.Cin + 63) / 64);
                if (d.epi == EPI_PLAIN && (long)g1.x * g1.y <= g_small16_max) {
                    // at most one 128x128 tile per CU (the O / FF2 projections of one utterance: 144 tiles): 64x64 tiles put
                    // four times as many, shorter workgroups on ALL CUs (O 15.6 -> 12.1 us, FF2 24.6 -> 21.4 us; with 288
                    // tiles, FF1, the doubled DMA bytes per flop already cost more than the balance gains)
                    e.Tm = (d.M + 63) / 64; e.Tn = (d.N + 63) / 64; e.RT = B * e.Tm; e.RC = 0;
                    dim3 g2(e.RT * e.Tn, d.G);
                    if (e.lds_epi) MI_LAUNCH((conv_gemm_dma_kernel<T, TO, true, 2, 64>), T, TO, g2, blk, 0, s, e);
                    else MI_LAUNCH((conv_gemm_dma_kernel<T, TO, false, 2, 64>), T, TO, g2, blk, 0, s, e);
                    MI_HIP(hipGetLastError());
                    return;
                }
                if (g_ring4 && (long)g1.x * g1.y <= g_ring4_max && nchunks >= 6) {
                    // at most one workgroup per CU: the four-stage ring hides the DMA round trip that the two-buffer
                    // loop exposes when a CU has no second workgroup to switch to
                    if (e.lds_epi) MI_LAUNCH((conv_gemm_dma_kernel<T, TO, true, 4>), T, TO, g1, blk, 0, s, e);
                    else MI_LAUNCH((conv_gemm_dma_kernel<T, TO, false, 4>), T, TO, g1, blk, 0, s, e);
                } else if (e.lds_epi) MI_LAUNCH((conv_gemm_dma_kernel<T, TO, true>), T, TO, g1, blk, 0, s, e);
                else MI_LAUNCH((conv_gemm_dma_kernel<T, TO, false>), T, TO, g1, blk, 0, s, e);
                MI_HIP(hipGetLastError());
                return;
            }
        }
        if constexpr (sizeof(T) == 4) {
            // fp32: the same 128x128 LDS-DMA kernel with 32-float K chunks (MFMA-bound: 8x the MFMA cycles per DMA byte)
            if (g_use_dma && g_f32_dma && d.Cin % 4 == 0 && d.K % d.Cin == 0) {
                no_fold();
                ConvGemmDev e = d;
                e.Tm = (d.M + 127) / 128; e.Tn = (d.N + 127) / 128; e.RT = B * e.Tm;
                e.RC = g_xcd_order ? (e.RT + 7) / 8 : 0;
                e.use_buf = buf_ok(d, 4);
                if (g_f32_small && (long)e.RT * e.Tn * d.G < g_f32_small_max) {
                    // fewer than a few 128x128 tiles per CU (one utterance: 144 / 288 / 432 tiles on 256 CUs): 64x64 tiles
                    // balance the chip (O projection 144 -> 576 workgroups: makespan 3 quarter-tiles instead of 4)
                    e.Tm = (d.M + 63) / 64; e.Tn = (d.N + 63) / 64; e.RT = B * e.Tm; e.RC = 0;
                    dim3 g2(e.RT * e.Tn, d.G);
                    if (e.lds_epi) MI_LAUNCH((conv_gemm_dma_kernel<T, TO, true, 2, 64>), T, TO, g2, blk, 0, s, e);
                    else MI_LAUNCH((conv_gemm_dma_kernel<T, TO, false, 2, 64>), T, TO, g2, blk, 0, s, e);
                    MI_HIP(hipGetLastError());
                    return;
                }
                dim3 g1(e.RC > 0 ? 8 * e.RC * e.Tn : e.RT * e.Tn, d.G);
                if (e.lds_epi) MI_LAUNCH((conv_gemm_dma_kernel<T, TO, true>), T, TO, g1, blk, 0, s, e);
                else MI_LAUNCH((conv_gemm_dma_kernel<T, TO, false>), T, TO, g1, blk, 0, s, e);
                MI_HIP(hipGetLastError());
                return;
            }
        }
        no_fold();
        MI_LAUNCH((conv_gemm_kernel<T, TO, 128, 128, 2, 2, KC>), T, TO, grid, blk, 0, s, d);
    }
    MI_HIP(hipGetLastError());
}

// test / tuning hook: force a tile configuration regardless of problem size
bool gemm_set_option(const char* key, long v) {
    const std::string k(key);
    if (k == "gemm_big_tile_min") g_big_min = v;
    else if (k == "gemm_n192_min") g_n192_min = v;
    else if (k == "gemm_mid_tile_min") g_mid_min = v;
    else if (k == "gemm_dma3_k_min") g_k_min = v;
    else if (k == "gemm_use_dma3") g_use_dma3 = v != 0;
    else if (k == "gemm_use_dma") g_use_dma = v != 0;
    else if (k == "gemm_big_tiles") g_big_tiles = v != 0;
    else if (k == "gemm_n192") g_n192 = v != 0;
    else if (k == "gconv_two_taps") gconv_pairs_set_option(v);
    else if (k == "gconv16") gconv16_set_option(v);
    else if (k == "gemm_f32_dma") g_f32_dma = v != 0;
    else if (k == "gemm_ring4") g_ring4 = v != 0;
    else if (k == "gemm_buf") g_buf = v != 0;
    else if (k == "gemm_f32_small") g_f32_small = v != 0;
    else if (k == "gemm_f32_small_max") g_f32_small_max = v;
    else if (k == "gemm_small16_max") g_small16_max = v;
    else if (k == "gemm_ring4_max") g_ring4_max = v;
    else if (k == "gemm_sk") g_sk = v;
    else if (k == "gemm_sk_stages") g_sk_stages = v;
    else if (k == "gemm_f32_x3") g_x3 = v;
    else if (k == "gemm_f32_x3p") g_x3p = v;
    else if (k == "gemm_f32_n64_pairs") g_f32_n64_pairs = v;
    else if (k == "gemm_f32_gconv") g_f32_gconv = v;
    else if (k == "gemm_f32_planes") { if (v != 2 && v != 3) return false; g_x3p_np = v; }
    else if (k == "gemm_x3p_noalign") x3p_set_option(0, v);
    else if (k == "gemm_x3p_grid") x3p_set_option(1, v);
    else if (k == "gemm_x3d") x3d_set_option(0, v);
    else if (k == "gemm_x3d_min_eff") x3d_set_option(1, v);
    else if (k == "gemm_ph8") g_ph8 = v;
    else if (k == "gemm_ph8_min_tiles") g_ph8_min_tiles = v;
    else if (k == "gemm_row_split") g_row_split = v;
    else if (k == "gemm_ph8_order") g_ph8_order = v;
    else if (k == "gemm_ph8_split_max") ph8_set_split_max(v);               // (test hooks of the split-tail instantiation)
    else if (k == "gemm_ph8_split_min_nk") ph8_set_split_min_nk(v);
    else return false;
    return true;
}

// the bf16x3 kernel (gemm_x3.hip) takes this launch: decided ONCE here, because the epilogue kind depends on it
static bool x3_eligible(const ConvGemm& p) {
    if (!opt_x3() || !p.w3 || p.dtype != MI_F32 || (p.out_dtype >= 0 && p.out_dtype != MI_F32)) return false;
    if (!p.sk_ws || !p.sk_flags || p.sk_slots < 256 || p.B != 1 || p.G != 1 || p.taps != 1 || p.pad != 0 || p.Cin % 32 != 0 || p.M <= 128) return false;
    if (p.epi != EPI_PLAIN && p.epi != EPI_QKV_ROPE) return false;
    if (p.epi == EPI_QKV_ROPE && !(p.head_dim == 64 && p.rope_pack && (p.rows_per_item == 0 ? p.M : p.rows_per_item) >= 64 &&
                                   ((uintptr_t)p.out % 16) == 0 && ((uintptr_t)p.out2 % 16) == 0 && ((uintptr_t)p.out3 % 16) == 0)) return false;
    const long tiles = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (tiles < 64 || tiles > g_sk_max_tiles) return false;
    const long a_bytes = (((long)p.T_in - 1) * p.x_rstride + p.Cin) * 4, b_bytes = (long)3 * p.N * p.Cin * 2;
    return g_buf && a_bytes + (long)512 * p.x_rstride * 4 < 0x7fff0000L && b_bytes < 0x7fff0000L;
}

// ... and the panel-plane form of it (gemm_x3p.hip): whole 128-column weight panels, whole 32-deep chunks
static bool x3p_eligible(const ConvGemm& p) {
    if (!opt_x3p() || !p.xp || !p.w3p) return false;
    ConvGemm q = p;
    q.w3 = p.w3p;                                     // same conditions as the round-2 kernel (q.w3 only has to be non-null)
    if (!x3_eligible(q)) return false;
    const long nch = p.Cin / 32;
    const long chb = x3p_chunk_bytes(p.np);
    return (p.np == 2 || p.np == 3) && p.N % 128 == 0 && ((long)(p.M + 127) / 128) * nch * chb < 0x7fff0000L && ((long)p.N / 128) * nch * chb < 0x7fff0000L;
}

bool gemm_x3p_would_run(const ConvGemm& p) { return x3p_eligible(p) && p.B == 1 && p.G == 1; }

// outputs leave through the LDS-staged, 16-byte-store epilogues (decided once per launch: the kernels instantiate either kind)
static bool lds_epi_for(const ConvGemm& p, int odt, bool use_x3) {
    static int no_lds_epi = -1;
    if (no_lds_epi < 0) { const char* q = std::getenv("MI355TTS_NO_LDS_EPI"); no_lds_epi = (q && q[0] == '1') ? 1 : 0; }
    if (no_lds_epi) return false;
    const int ch = 16 / (int)dtype_size(odt);
    // measured: +8-14 % on the K = 1024 DiT linears in the 2-blocks-per-CU 128x128 kernel, a loss on the conv shapes
    // (N <= 768) and in the one-block-per-CU 8-wave kernels, so it is used for wide linear layers only
    // (fp32 outputs: only the bf16x3 kernel has the staged QKV epilogue; rows leave as 32-byte stores, V either way)
    const bool qkv_lds = p.epi == EPI_QKV_ROPE && p.head_dim == 64 && (dtype_size(odt) == 2 || use_x3) && p.G == 1 &&
                         (p.rows_per_item == 0 ? p.M : p.rows_per_item) >= 64 && ((uintptr_t)p.out % 16) == 0 &&
                         ((uintptr_t)p.out2 % 16) == 0;
    if (qkv_lds) return true;
    return p.epi == EPI_PLAIN && p.taps == 1 && p.N >= 1024 && p.N % ch == 0 && p.out_rstride % ch == 0 &&
           p.out_bstride % ch == 0 && ((uintptr_t)p.out % 16) == 0 && (!p.res || ((uintptr_t)p.res % 16) == 0);
}

// ... and can it leave its output as panel planes (ConvGemm::out_planes)?  Only the LDS-staged epilogue writes them, and that one is
// chosen for wide outputs only (N >= 1024): a narrow model (dim 256: FF1 has N = 512) at a large M is eligible for the kernel but
// not for plane output — found by the N = 4096 limit tests of round 4 (the launch refused loudly; the caller now asks first)
bool gemm_x3p_can_write_planes(const ConvGemm& p) {
    return gemm_x3p_would_run(p) && p.epi == EPI_PLAIN && !p.res && !p.gate && !p.accumulate && p.alpha == 1.f && p.N % 32 == 0 &&
           lds_epi_for(p, MI_F32, true);
}

// The AdaLN fold (ConvGemm::ln_*) lives in the LDS-staged epilogues of three kernels: linear_x3p (fp32 engines), linear_ph8 and
// conv_gemm_dma_kernel (16-bit engines).  This answers, for the caller that has to choose between the fold and a row-norm launch,
// whether launch_conv_gemm(p) ends up on one of them; dispatch_tiles() refuses (loudly) if the two ever disagree.
bool gemm_ln_fold_ok(const ConvGemm& p) {
    const int odt = p.out_dtype < 0 ? p.dtype : p.out_dtype;
    if (p.B != 1 || p.G != 1 || p.taps != 1 || p.pad != 0 || p.alpha != 1.f || p.accumulate || p.gate_bstride != 0 || p.N % 64 != 0 || p.M <= 128) return false;
    if (p.epi != EPI_PLAIN && p.epi != EPI_QKV_ROPE) return false;
    if (p.dtype == MI_F32) return gemm_x3p_would_run(p) && lds_epi_for(p, odt, true);
    if (!lds_epi_for(p, odt, false) || !g_use_dma || p.Cin % 64 != 0 || p.N <= 64) return false;
    if (g_use_dma3 && p.Cin > g_k_min) return false;                    // K > 2048: the 256-row dma3 kernels are the faster choice (a preference, not a constraint: dispatch_tiles keeps fold launches off the kernels without fold epilogues)
    return true;
}

void launch_conv_gemm(const ConvGemm& p_in, hipStream_t s) {
    ConvGemm p = p_in;
    if (p.B > 1 && p.taps == 1 && p.G == 1 && p.pad == 0 && p.epi == EPI_PLAIN && p.M == p.T_in && p.gate_bstride == 0 &&
        p.x_bstride == (long)p.M * p.x_rstride && p.out_bstride == (long)p.M * p.out_rstride) {
        p.T_in = p.M = p.B * p.M; p.B = 1;          // one-tap GEMM over contiguous batch items: a single M axis
    }
    const int odt = p.out_dtype < 0 ? p.dtype : p.out_dtype;
    // Row split of a 16-bit linear layer whose 256x256 tiles spill into a nearly empty last round.  A batch of 8 utterances is
    // M = 18016 rows = 70.4 row tiles: N = 1024 gives 284 tiles on 256 CUs — a full round and then 28 tiles with 228 CUs idle for
    // as long again (O / FF2 ran at 312 TFLOP/s against ~600 for QKV / FF1); N = 2048: 568 = two rounds + 56.  The rows of whole
    // rounds (M1 = a multiple of 256 * CUs / column tiles) go to the eight-phase kernel as before, the remaining rows (1632
    // here) are a second launch that the dispatch below hands to the 128x128 kernels: 104 / 208 small tiles instead of a round
    // of big ones.  Rows are independent in every EPI_PLAIN epilogue (bias, activation, gate per column, residual per element).
    const bool split_plain = p.epi == EPI_PLAIN && p.gate_bstride == 0 && !p.out_planes && !p.accumulate;
    const bool split_qkv = p.epi == EPI_QKV_ROPE && p.rows_per_item >= 32 && !p.kv_planes && p.m_off == 0;   // the epilogue indexes
                                                      // tokens from the flattened row: the second launch carries its row offset (m_off)
    if (g_row_split != 0 && g_ph8 != 0 && dtype_size(p.dtype) == 2 && p.B == 1 && p.G == 1 && p.taps == 1 && p.pad == 0 &&
        (split_plain || split_qkv) && p.M == p.T_in && p.N % 256 == 0 && !p.xp) {
        int dev = 0, cus = 256;
        MI_HIP(hipGetDevice(&dev));
        {
            static int cu_count[16] = {0};
            if (!cu_count[dev & 15]) { hipDeviceProp_t pr; MI_HIP(hipGetDeviceProperties(&pr, dev)); cu_count[dev & 15] = pr.multiProcessorCount; }
            cus = cu_count[dev & 15];
        }
        const long ntn = p.N / 256;
        long gc = cus, t = ntn;
        while (t) { const long u = gc % t; gc = t; t = u; }       // gcd(cus, ntn)
        const long rstep = cus / gc;                              // row tiles that make a whole number of rounds
        const long r = (p.M / 256) / rstep * rstep;               // N = 1024 / 2048 / 3072 on 256 CUs: multiples of 64 / 32 / 64
        const long rem = p.M - 256 * r;
        const long rem_tiles = (rem + 255) / 256 * ntn;
        if (r >= 1 && rem > 0 && rem_tiles * 2 < cus && r * ntn >= g_ph8_min_tiles) {
            ConvGemm a = p, b = p;
            a.M = a.T_in = (int)(256 * r);
            b.M = b.T_in = (int)rem;
            b.x = (const char*)p.x + (size_t)(256 * r) * p.x_rstride * dtype_size(p.dtype);
            if (split_qkv) b.m_off = (int)(256 * r);       // (the fold's statistics are indexed by the flattened row there too)
            else {
                b.out = (char*)p.out + (size_t)(256 * r) * p.out_rstride * dtype_size(odt);
                if (p.res) b.res = (const char*)p.res + (size_t)(256 * r) * p.out_rstride * dtype_size(odt);
                // AdaLN fold: the per-row operands of the second launch start at its first row (16-bit engines: ln_out is rows [M][N])
                if (p.ln_stats_in) b.ln_stats_in = p.ln_stats_in + (size_t)(256 * r) * (p.ln_final ? 1 : p.ln_dim / LN_BLK) * 2;
                if (p.ln_stats_out) {
                    b.ln_stats_out = p.ln_stats_out + (size_t)(256 * r) * (p.N / LN_BLK) * 2;
                    b.ln_out = (char*)p.ln_out + (size_t)(256 * r) * p.N * dtype_size(p.dtype);
                }
            }
            launch_conv_gemm(a, s);
            launch_conv_gemm(b, s);
            return;
        }
    }
    const int vec = 16 / (int)dtype_size(p.dtype);
    MI_REQUIRE(p.Cin % vec == 0, "conv_gemm: Cin must be a multiple of the 16-byte vector");
    MI_REQUIRE(p.x_rstride % vec == 0 && p.x_bstride % vec == 0 && p.x_goff % vec == 0, "conv_gemm: x strides");
    MI_REQUIRE(((uintptr_t)p.x % 16) == 0 && ((uintptr_t)p.w % 16) == 0, "conv_gemm: 16-byte alignment");
    MI_REQUIRE(odt == p.dtype || odt == MI_F32, "conv_gemm: out dtype");
    MI_REQUIRE(p.M > 0 && p.N > 0 && p.B > 0 && p.G > 0, "conv_gemm: empty problem");
    ConvGemmDev d;
    d.x = p.x; d.w = p.w; d.bias = p.bias; d.out = p.out; d.res = p.res; d.gate = p.gate;
    d.gate_bstride = p.gate_bstride;
    d.G = p.G; d.T_in = p.T_in; d.M = p.M; d.N = p.N; d.Cin = p.Cin; d.K = p.taps * p.Cin; d.dil = p.dil; d.pad = p.pad;
    d.x_bstride = p.x_bstride; d.x_rstride = p.x_rstride; d.out_bstride = p.out_bstride; d.out_rstride = p.out_rstride;
    d.x_goff = p.x_goff;
    d.act = p.act; d.alpha = p.alpha; d.accumulate = p.accumulate; d.epi = p.epi;
    d.u = p.u; d.Cout = p.Cout; d.padT = p.padT; d.T_out = p.T_out;
    d.rope_cos = p.rope_cos; d.rope_sin = p.rope_sin; d.rope_pack = p.rope_pack; d.heads = p.heads; d.head_dim = p.head_dim;
    d.out2 = p.out2; d.out3 = p.out3; d.v_ld = p.v_ld; d.Mb = p.rows_per_item; d.m_off = p.m_off;
    MI_REQUIRE(p.m_off == 0 || (p.epi == EPI_QKV_ROPE && p.rows_per_item >= 32), "conv_gemm: a row offset needs the flattened QKV epilogue");
    d.kv_planes = p.kv_planes; d.k_ld = p.k_ld;
    d.sk_ws = p.sk_ws; d.sk_flags = p.sk_flags; d.sk_slots = p.sk_slots;
    const bool use_x3p = x3p_eligible(p);
    const bool use_x3 = use_x3p || x3_eligible(p);
    d.w3 = use_x3 ? p.w3 : nullptr;
    d.xp = use_x3p ? p.xp : nullptr; d.w3p = use_x3p ? p.w3p : nullptr; d.np = p.np;
    d.out_planes = nullptr;
    if (p.out_planes) {
        MI_REQUIRE(use_x3p && p.epi == EPI_PLAIN && !p.res && !p.gate && !p.accumulate && p.alpha == 1.f && p.N % 32 == 0,
                   "conv_gemm: out_planes needs the panel-plane kernel and a plain bias + activation epilogue");
        d.out_planes = p.out_planes;
    }
    d.ln_scale = p.ln_scale; d.ln_out = p.ln_out; d.ln_stats_out = p.ln_stats_out; d.ln_out_np = p.ln_out_np;
    d.ln_stats_in = p.ln_stats_in; d.ln_p = p.ln_p; d.ln_c = p.ln_c; d.ln_dim = p.ln_dim; d.ln_eps = p.ln_eps; d.ln_final = p.ln_final;
    d.sat = p.sat;
    if (p.ln_stats_in || p.ln_stats_out) {
        MI_REQUIRE(!(p.ln_stats_in && p.ln_stats_out), "conv_gemm: a launch is the producer OR the consumer of the AdaLN fold");
        MI_REQUIRE(p.B == 1 && p.G == 1 && p.taps == 1 && p.alpha == 1.f && !p.accumulate && p.gate_bstride == 0 && p.N % 64 == 0,
                   "conv_gemm: AdaLN fold needs a plain linear layer over one M axis");
        if (p.ln_stats_out)
            MI_REQUIRE(p.epi == EPI_PLAIN && odt == MI_F32 && p.res && p.ln_scale && p.ln_out && p.act == ACT_NONE && !p.out_planes &&
                           ((uintptr_t)p.ln_out % 16) == 0 && ((uintptr_t)p.ln_stats_out % 8) == 0 && ((uintptr_t)p.ln_scale % 16) == 0 &&
                           (p.dtype != MI_F32 || p.ln_out_np == p.np),
                       "conv_gemm: AdaLN fold producer: fp32 residual rows out, (1 + scale) and the next operand's buffer");
        else
            MI_REQUIRE(p.ln_p && p.ln_c && p.ln_dim >= 128 && p.ln_dim % 128 == 0 && !p.res && !p.gate && ((uintptr_t)p.ln_stats_in % 8) == 0 && (p.ln_final || ((uintptr_t)p.ln_stats_in % 16) == 0) &&
                           ((uintptr_t)p.ln_p % 16) == 0 && ((uintptr_t)p.ln_c % 16) == 0 && (odt != MI_F32 || p.epi == EPI_QKV_ROPE || p.out_planes),
                       "conv_gemm: AdaLN fold consumer: statistics, W(1 + scale) and W shift + b vectors");
    }
    d.tail_tiles = 0; d.tail_split = 1;
    d.use_buf = 0;
    {
        static std::once_flag env_once;      // handles on different threads may launch concurrently
        std::call_once(env_once, [] { const char* e = std::getenv("MI355TTS_NO_DMA_GEMM"); g_use_dma = !(e && e[0] == '1');
            const char* x = std::getenv("MI355TTS_XCD_ORDER"); g_xcd_order = x && x[0] == '1';
            const char* y = std::getenv("MI355TTS_NO_DMA3_GEMM"); g_use_dma3 = !(y && y[0] == '1');
            const char* z = std::getenv("MI355TTS_NO_BIG_TILES"); g_big_tiles = !(z && z[0] == '1');
            if (const char* m = std::getenv("MI355TTS_BIG_TILE_MIN")) g_big_min = std::atol(m);
            if (const char* m = std::getenv("MI355TTS_DMA3_K_MIN")) g_k_min = std::atol(m);
            if (const char* n = std::getenv("MI355TTS_NO_N192")) g_n192 = !(n[0] == '1');
            if (const char* n = std::getenv("MI355TTS_NO_RING4")) g_ring4 = !(n[0] == '1');
            if (const char* n = std::getenv("MI355TTS_NO_BUF")) g_buf = !(n[0] == '1');
            if (const char* n = std::getenv("MI355TTS_NO_F32_SMALL")) g_f32_small = !(n[0] == '1');
            if (const char* n = std::getenv("MI355TTS_F32_SMALL_MAX")) g_f32_small_max = std::atol(n);
            if (const char* n = std::getenv("MI355TTS_F32_N64_DMA")) g_f32_n64_dma = std::atol(n);
            if (const char* n = std::getenv("MI355TTS_N64_DMA16")) g_n64_dma16 = std::atol(n);
            if (const char* n = std::getenv("MI355TTS_SMALL16_MAX")) g_small16_max = std::atol(n);
            if (const char* n = std::getenv("MI355TTS_RING4_MAX")) g_ring4_max = std::atol(n);
            if (const char* n = std::getenv("MI355TTS_SK")) g_sk = std::atol(n);
            if (const char* n = std::getenv("MI355TTS_SK_STAGES")) g_sk_stages = std::atol(n);
            if (const char* n = std::getenv("MI355TTS_SK_ORDER")) g_sk_order = std::atol(n);
            if (const char* n = std::getenv("MI355TTS_F32_X3")) g_x3 = std::atol(n);
            if (const char* n = std::getenv("MI355TTS_F32_X3P")) g_x3p = std::atol(n);
            if (const char* n = std::getenv("MI355TTS_X3P_NOALIGN")) x3p_set_option(0, std::atol(n));
            if (const char* n = std::getenv("MI355TTS_X3P_GRID")) x3p_set_option(1, std::atol(n));
            if (const char* n = std::getenv("MI355TTS_SK_QKV32")) g_sk_qkv32 = std::atol(n);
            if (const char* n = std::getenv("MI355TTS_PH8")) g_ph8 = std::atol(n);
            if (const char* n = std::getenv("MI355TTS_PH8_MIN")) g_ph8_min_tiles = std::atol(n);
            if (const char* n = std::getenv("MI355TTS_PH8_ORDER")) g_ph8_order = std::atol(n);
            if (const char* n = std::getenv("MI355TTS_PH8_SPLIT")) ph8_set_split_max(std::atol(n));
        });
        int dev = 0;
        MI_HIP(hipGetDevice(&dev));
        {
            static std::mutex zero_mu;
            std::lock_guard<std::mutex> lk(zero_mu);
            DevBuf& z = g_zero_page[dev & 15];
            if (!z.p) { z.ensure(4096); MI_HIP(hipMemset(z.p, 0, 4096)); MI_HIP(hipDeviceSynchronize()); }
            d.zero = z.p;
        }
        const char* dm = std::getenv("MI355TTS_GEMM_DBG");
        d.dbg = dm ? std::atoi(dm) : 0;
        // K-loop order of the DMA kernels is (channel chunk, tap), not (tap, channel chunk): sweeping all Cin channels
        // of a block's rows once per tap overflowed the 4 MiB XCD L2 between taps at C = 384 (32 resident blocks x 196 KB
        // of rows) and every tap re-fetched its rows from the fabric (PMC: 341 MB per launch against 100 MB algorithmic);
        // consecutive taps now re-read the same rows x 64 channels.  fp32 accumulation: only the summation order changes.
        // (A run-time switch between the two orders cost 12 % by itself, so the order is fixed.)
        d.lds_epi = lds_epi_for(p, odt, use_x3) ? 1 : 0;
    }
    if (p.ln_stats_in || p.ln_stats_out)
        MI_REQUIRE(d.lds_epi && (p.dtype != MI_F32 || use_x3p), "conv_gemm: the AdaLN fold lives in the LDS-staged epilogues of linear_x3p / linear_ph8 / conv_gemm_dma (gemm_ln_fold_ok)");
    if (p.kv_planes) MI_REQUIRE(p.epi == EPI_QKV_ROPE && p.dtype == MI_F32 && d.lds_epi && p.v_ld > 0 && p.k_ld > 0 && p.v_ld % 8 == 0,
                                "conv_gemm: kv_planes needs the fp32 LDS-staged QKV epilogue with transposed V");
    if (p.epi == EPI_CONVT) MI_REQUIRE(p.Cout > 0 && p.N == p.u * p.Cout, "conv_gemm: convT shape");
    if (p.epi == EPI_QKV_ROPE)
        MI_REQUIRE((p.rows_per_item == 0 || p.rows_per_item >= 32) && p.G == 1 && p.heads > 0 && p.head_dim % 2 == 0 && p.N == 3 * p.heads * p.head_dim && p.N % 32 == 0 &&
                       p.rope_cos && p.rope_sin && p.out2 && p.out3, "conv_gemm: qkv-rope epilogue arguments");

    const double esz = (double)dtype_size(p.dtype), osz = (double)dtype_size(odt);
    // algorithmic traffic: read x once, weights once, write out once (+ residual / accumulate reads)
    const double rows_out = p.epi == EPI_CONVT ? (double)p.T_out : (double)p.M;
    const double ncols = p.epi == EPI_CONVT ? (double)p.Cout : (double)p.N * p.G;
    double bytes = (double)p.B * p.T_in * (double)p.Cin * p.G * esz + (double)p.G * p.N * d.K * esz +
                   (double)p.B * rows_out * ncols * osz * (1.0 + (p.res ? 1.0 : 0.0) + (p.accumulate ? 1.0 : 0.0));
    double flops = 2.0 * p.B * p.G * (double)p.M * p.N * d.K;
    ProfScope ps(FAM_CONV_GEMM, s, bytes, flops);

    // fp32 grouped convolutions with 64 channels per group and >= 8 taps (the DiT's position convolution): gconv_pairs.hip
    if (p.dtype == MI_F32 && opt_x3() != 0 && opt_n64_pairs() != 0 && opt_gconv() != 0 && g_use_dma && launch_gconv_pairs(p, s)) return;
    // ... and the same shape on the 16-bit engines, weights as LDS images built at load: gconv16.hip
    if (p.dtype != MI_F32 && p.gcp_w && g_use_dma && launch_gconv16(p, s)) return;
    if (p.dtype == MI_F32) {
        dispatch_tiles<float, float>(d, p.B, s);
    } else if (p.dtype == MI_F16) {
        if (odt == MI_F32) dispatch_tiles<f16, float>(d, p.B, s);
        else dispatch_tiles<f16, f16>(d, p.B, s);
    } else {
        if (odt == MI_F32) dispatch_tiles<bf16, float>(d, p.B, s);
        else dispatch_tiles<bf16, bf16>(d, p.B, s);
    }
}

void SkWorkspace::ensure(int n_slots, hipStream_t s) {
    if (n_slots <= slots) return;
    ws.ensure((size_t)n_slots * 128 * 128 * 4);
    flags.ensure((size_t)n_slots * 4);
    MI_HIP(hipMemsetAsync(flags.p, 0, (size_t)n_slots * 4, s));
    MI_HIP(hipStreamSynchronize(s));
    slots = n_slots;
}

// The stream-K / split-tail hand-offs (gemm_sk.hip, gemm_x3.hip, gemm_ph8.hip) leave every flag at zero when a launch
// completes; a launch that faulted or was aborted (or a failed graph capture) may not have.  Consumers spin on these
// flags, so whoever observes an error on the stream re-zeroes them before the next launch.  Errors are swallowed: this
// runs on the error path.
bool SkWorkspace::tripped() const {
    if (!flags.p || slots <= 0) return false;
    int v = 0;
    if (hipMemcpy(&v, (const int*)flags.p + slots - 1, 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
    return v != 0;
}

void SkWorkspace::reset(hipStream_t s) {
    if (!flags.p || slots <= 0) return;
    (void)hipMemsetAsync(flags.p, 0, (size_t)slots * 4, s);
    (void)hipStreamSynchronize(s);
}

}  // namespace mi
