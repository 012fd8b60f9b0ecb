// gemm_x1d.hip — the 16-bit DiT linear layers of a batch of utterances as an EXACT-FIT DATA-PARALLEL tiling (round 6).
//
// gemm_ph8.hip tiles M x N into 256 x 256: at eight utterances (M = 18016 rows) that is 70.4 row tiles — N = 1024 gives 284 tiles
// on 256 CUs (a round and a ninth), so launch_conv_gemm splits the rows: whole rounds on the eight-phase kernel and the last 9 % of
// the rows as a second launch of small tiles that costs 13 % of the step.  The same shapes divide EXACTLY into
// 288 x 256 tiles: 18016 = 62.6 x 288 -> 63 row groups, x 4 | 8 | 12 column groups = 252 | 504 | 756 tiles = 1 | 2 | 3 rounds of 256
// CUs at 98.4 % of the slots and 97.7 % useful area — one launch, every workgroup one whole tile over the whole K, no second launch,
// no partial tiles.  x1d_plan takes this kernel when the useful area over whole rounds is >= 92 %; else the launch goes on as before.
//
// 288 = 18 x 16 rows: v_mfma_f32_16x16x32_{bf16,f16}.  Twelve waves (three per SIMD, <= 168 VGPRs) as 3 (M) x 4 (N): 96 x 64 per
// wave = 6 x 4 blocks, 96 accumulator registers; operands row-major [M][K] / [N][K] as the engines keep them, K chunks of 64 (whole
// 128-byte lines per row), LDS-DMA pieces of 8 rows x 128 bytes whose 16-byte slots are XOR-swizzled by (row >> 1) & 7 on the GLOBAL
// side (the LDS image of a piece is lane-linear), so that the 16-row fragments are read conflict-free.  Two chunk slots of 68 KB: every
// fragment read of a chunk is ISSUED before the chunk's barrier (A fragments are re-read for the next k-step right behind the row of
// MFMAs that used them last, B fragments are double-buffered), so behind the barrier the slot is dead and takes chunk c + 2 while
// the last row of MFMAs of chunk c covers the first reads of chunk c + 1.  Fragment reads are inline asm with counted lgkmcnt waits
// (see gemm_x3d.hip).  Epilogue: each wave parks 32 x 64 groups of its tile in a private LDS block and runs the shared epilogues of
// gemm_epilogue.h (QKV + RoPE, FF1 with the AdaLN fold, gated residual + fold producer) on them.
#include <atomic>
#include "common.h"
#include "mfma.h"
#include "gemm_epilogue.h"
#include <cstdlib>
#include <type_traits>
#include <algorithm>

namespace mi {

typedef float x1_f4 __attribute__((ext_vector_type(4)));

template <typename RSRC>
__device__ __forceinline__ void x1d_dma(RSRC rsrc, int voff, unsigned lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
#endif
}
template <int N> __device__ __forceinline__ void x1d_wait_lgkm() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory");
#endif
}
template <typename FR, int OFF> __device__ __forceinline__ void x1d_rd128(FR& d, unsigned a) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF) : "memory");
#endif
}
template <int N, int I = 0, typename F> __device__ __forceinline__ void x1d_unroll(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); x1d_unroll<N, I + 1>(f); }
}
template <typename T> struct X1dMfma;
template <> struct X1dMfma<bf16> {
    using Frag = bf16x8;
    static __device__ __forceinline__ x1_f4 mma(Frag a, Frag b, x1_f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct X1dMfma<f16> {
    using Frag = f16x8;
    static __device__ __forceinline__ x1_f4 mma(Frag a, Frag b, x1_f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

// p.x rows [M][x_rstride] of T, p.w rows [N][K] of T; p.Tm x p.Tn = row groups x column groups, p.RT = band height of the tile order
template <typename T, typename TO>
__global__ __launch_bounds__(768) void linear_x1d_kernel(const ConvGemmDev p) {
    using MF = X1dMfma<T>;
    using Frag = typename MF::Frag;
    constexpr int TM = 288, TN = 256, NA = 6, NB = 4;                    // wave tile 96 x 64 = 6 x 4 blocks of 16 x 16
    constexpr int A_BYTES = TM * 128, B_BYTES = TN * 128, CH = A_BYTES + B_BYTES;      // one K chunk of 64: 36 + 32 KB
    constexpr int PA = TM / 8, PB = TN / 8, PC = PA + PB;                 // LDS-DMA pieces (8 rows x 128 B) per chunk: 36 + 32
    constexpr int NHI = (PC + 11) / 12, NHIW = PC - 12 * (NHI - 1);       // 6 pieces for waves < 8, 5 for the rest
    constexpr int BLK = 2176;                                             // floats of a wave's private 32 x 64 epilogue block
    static_assert(2 * CH <= 163840 && 12 * BLK * 4 <= 2 * CH, "x1d: LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * CH];
    (void)smem;
#if defined(__HIP_DEVICE_COMPILE__)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per = (int)gridDim.x >> 3;
    const int u = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    const int RGn = p.Tm, CGn = p.Tn, band = p.RT;
    if (u >= RGn * CGn) return;
    const int bi = u / (band * CGn);
    const int rem = u - bi * band * CGn;
    const int bh = min(band, RGn - bi * band);
    const int cg = rem / bh, rg = bi * band + rem - cg * bh;
    const int m0 = rg * TM, n0 = cg * TN;
    const int nk = p.K >> 6;
    const int wm = wave >> 2, wn = wave & 3;

    // ---- LDS-DMA pieces: q = wave + 12 i; lane -> row lane / 8 of the piece, LDS slot lane % 8 = global slot ^ ((row >> 1) & 7) ----
    const unsigned smem_lds = (unsigned)(unsigned long)(const __attribute__((address_space(3))) void*)smem;
    const int bytesA = (int)((((long)p.T_in - 1) * p.x_rstride + p.Cin) * 2L), bytesB = (int)((long)p.N * p.K * 2L);
    constexpr int OOB = 0x7fffff00;
    __amdgpu_buffer_rsrc_t pc_rs[NHI];
    int pc_voff[NHI];                                                     // per lane: byte offset of its 16 bytes at K chunk 0
    unsigned pc_lofs[NHI];
#pragma unroll
    for (int i = 0; i < NHI; ++i) {
        const int q = wave + 12 * i;
        const bool isA = q < PA;
        const int pa = isA ? q : q - PA;                                  // piece of its operand: rows pa * 8 .. + 7 of the tile
        const int lrow = lane >> 3;
        const int gslot = (lane & 7) ^ ((4 * (pa & 1) + (lane >> 4)) & 7);     // ((pa * 8 + lrow) >> 1) & 7
        const long row = (long)(isA ? m0 : n0) + pa * 8 + lrow;
        const long ld = isA ? p.x_rstride : (long)p.K;
        const bool ok = row < (isA ? (long)p.M : (long)p.N);
        pc_voff[i] = ok ? (int)((row * ld + gslot * 8) * 2L) : OOB;
        pc_rs[i] = __builtin_amdgcn_make_buffer_rsrc(isA ? (void*)p.x : (void*)p.w, 0, isA ? bytesA : bytesB, 0x00020000);
        pc_lofs[i] = (unsigned)__builtin_amdgcn_readfirstlane(isA ? pa * 1024 : A_BYTES + pa * 1024);
    }
    const bool hi_wave = wave < NHIW;
    auto issue_chunk = [&](int c) __attribute__((always_inline)) {       // chunk c -> slot c & 1 (past the end: zero fill, nothing fetched)
        const int koff = __builtin_amdgcn_readfirstlane(c < nk ? c * 128 : OOB);
        const unsigned base = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)((c & 1) * CH));
#pragma unroll
        for (int i = 0; i < NHI; ++i) {
            if (i == NHI - 1 && !hi_wave) continue;
            x1d_dma(pc_rs[i], (int)((unsigned)pc_voff[i] + (unsigned)koff), base + pc_lofs[i]);
        }
    };

    // ---- fragment addresses: lane -> row lane & 15 of a 16-row block, k-slot 4 s + (lane >> 4) of k-step s, swizzled -----------
    const int fr = lane & 15, fq = lane >> 4, ff = (fr >> 1) & 7;
    const unsigned fl0 = (unsigned)(fr * 128 + (((0 + fq) ^ ff) << 4)), fl1 = (unsigned)(fr * 128 + (((4 + fq) ^ ff) << 4));
    const unsigned a_w = (unsigned)(wm * 96 * 128), b_w = (unsigned)(A_BYTES + wn * 64 * 128);

    Frag a[NA], b[2][NB];
    x1_f4 acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = x1_f4{0.f, 0.f, 0.f, 0.f};
    auto rd_a1 = [&](unsigned s, auto I_) __attribute__((always_inline)) { constexpr int i = decltype(I_)::value; x1d_rd128<Frag, i * 2048>(a[i], s); };
    auto rd_b = [&](unsigned s, Frag (&dst)[NB]) __attribute__((always_inline)) {
        x1d_unroll<NB>([&](auto J_) __attribute__((always_inline)) { constexpr int j = decltype(J_)::value; x1d_rd128<Frag, j * 2048>(dst[j], s); });
    };
    auto row_mma = [&](auto I_, const Frag (&bb)[NB]) __attribute__((always_inline)) {
        constexpr int i = decltype(I_)::value;
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = MF::mma(a[i], bb[j], acc[i][j]);
    };
#define X1D_SB() __builtin_amdgcn_sched_barrier(0)

    // ---- prologue ----------------------------------------------------------------------------------------------------------------
    issue_chunk(0);
    issue_chunk(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        const unsigned s0 = smem_lds;
        x1d_unroll<NA - 1>([&](auto I_) __attribute__((always_inline)) { rd_a1(s0 + a_w + fl0, I_); });
        rd_b(s0 + b_w + fl0, b[0]);
        rd_a1(s0 + a_w + fl0, std::integral_constant<int, NA - 1>{});
    }
    for (int c = 0; c < nk; ++c) {
        const unsigned sc = smem_lds + (unsigned)((c & 1) * CH), sn = smem_lds + (unsigned)(((c + 1) & 1) * CH);
        // ---- k-step 0: rows 0 .. 5 on b[0]; b[1] (k-step 1) requested first, every A fragment re-read for k-step 1 behind its row -------
        rd_b(sc + b_w + fl1, b[1]);
        x1d_wait_lgkm<1 + NB>();                                          // a[0 .. 4], b[0] are in (a[5] and b[1] may be on their way)
        x1d_unroll<NA>([&](auto I_) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value;
            if constexpr (i == NA - 1) x1d_wait_lgkm<NB + NA - 1>();      // a[5]: behind it b[1] and the re-read a[0 .. 4]
            X1D_SB(); row_mma(I_, b[0]); X1D_SB();
            rd_a1(sc + a_w + fl1, I_);
        });
        // ---- k-step 1: rows 0 .. 4 on b[1] --------------------------------------------------------------------------------------------
        x1d_unroll<NA - 1>([&](auto I_) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value;
            x1d_wait_lgkm<NA - 1 - i>();                                  // b[1] and a[0 .. i] of k-step 1 are in
            X1D_SB(); row_mma(I_, b[1]); X1D_SB();
        });
        // ---- chunk boundary: every read of chunk c has been issued; behind the barrier its slot takes chunk c + 2 ----------------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this wave's pieces of chunk c + 1 have landed
        x1d_wait_lgkm<0>();
        __builtin_amdgcn_s_barrier();
        issue_chunk(c + 2);
        x1d_unroll<NA - 1>([&](auto I_) __attribute__((always_inline)) { rd_a1(sn + a_w + fl0, I_); });
        rd_b(sn + b_w + fl0, b[0]);
        X1D_SB(); row_mma(std::integral_constant<int, NA - 1>{}, b[1]); X1D_SB();       // the last row of chunk c covers those reads
        rd_a1(sn + a_w + fl0, std::integral_constant<int, NA - 1>{});
    }
#undef X1D_SB
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    x1d_wait_lgkm<0>();
    __builtin_amdgcn_s_barrier();
    if (p.dbg & 4) return;

    // ---- epilogue: 32 x 64 groups of the wave tile through a private LDS block, then the shared per-role epilogues --------------------
    float* blk = reinterpret_cast<float*>(smem) + wave * BLK;
    const int lr = lane & 31, lk = lane >> 5;
    float lrs[3] = {0.f, 0.f, 0.f}, lmr[3] = {0.f, 0.f, 0.f};
    if constexpr (sizeof(TO) == 2) {
        if (p.ln_stats_in) {
#pragma unroll
            for (int g = 0; g < 3; ++g) ln_rows32(p, (long)p.m_off + m0 + wm * 96 + g * 32, (long)p.m_off + p.M - 1, lr, lk, lrs[g], lmr[g]);
        }
    }
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int mrow = m0 + wm * 96 + g * 32;
        if (mrow >= p.M) continue;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) blk[(ii * 16 + 4 * (lane >> 4) + r) * 64 + j * 16 + (lane & 15)] = acc[2 * g + ii][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x16 h[1][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[0][j][r] = blk[((r & 3) + 8 * (r >> 2) + 4 * lk) * 64 + j * 32 + lr];
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int ncol = n0 + wn * 64;
        if constexpr (sizeof(TO) == 2) {
            if (p.epi == EPI_QKV_ROPE) {
                if (p.ln_stats_in) gemm_epilogue_qkv_lds<TO, 1, true, true>(h, p, mrow, n0, 0, 0, wn, lr, lk, blk, lrs + g, lmr + g);
                else gemm_epilogue_qkv_lds<TO, 1>(h, p, mrow, n0, 0, 0, wn, lr, lk, blk);
            } else if (p.ln_stats_in) gemm_epilogue_ln_in<TO, 1, 2, 2, true>(h, p, mrow, ncol, lr, lk, blk, lrs + g, lmr + g);
            else gemm_epilogue_lds<TO, 1, 2, 32, 64>(h, p, mrow, n0, 0, 0, 0, wn, lr, lk, blk);
        } else {
            if (p.ln_stats_out) gemm_epilogue_resid_ln<T, 1, 2, 2>(h, p, mrow, ncol, lr, lk, blk);
            else gemm_epilogue_lds<TO, 1, 2, 32, 64>(h, p, mrow, n0, 0, 0, 0, wn, lr, lk, blk);
        }
    }
#endif
}

static std::atomic<long> g_x1d = 1, g_x1d_min_eff = 92;       // options gemm_x1d (0 off, 1 automatic), gemm_x1d_min_eff (per cent)
void x1d_set_option(int which, long v) { if (which == 0) g_x1d = v; else g_x1d_min_eff = v; }

// Does the exact-fit tiling pay for this 16-bit linear layer?  (M, N, K of the launch; cus = CUs of the device)
bool x1d_plan(int M, int N, int K, int cus, int& rgn, int& cgn, int& band) {
    static int env_read = 0;
    if (!env_read) {
        env_read = 1;
        if (const char* s = std::getenv("MI355TTS_X1D")) g_x1d = std::atol(s);
        if (const char* s = std::getenv("MI355TTS_X1D_MIN_EFF")) g_x1d_min_eff = std::atol(s);
    }
    if (!g_x1d || N % 256 != 0 || K % 64 != 0 || K < 128 || cus < 8) return false;
    rgn = (M + 287) / 288; cgn = N / 256;
    const long tiles = (long)rgn * cgn, rounds = (tiles + cus - 1) / cus;
    const double eff = (double)M * N / ((double)rounds * cus * 288.0 * 256.0);
    if (eff * 100.0 < (double)g_x1d_min_eff) return false;
    // band height a: an XCD's 32 concurrent tiles span a row groups x 32 / a column groups; fabric bytes ~ a * 288 + (32 / a) * 256 rows
    int ba = 1; double bc = 1e300;
    for (int a = 1; a <= 32; a *= 2) {
        if (a > rgn && a > 1) break;
        const double c = (double)std::min(a, rgn) * 288.0 + (double)std::min(32 / a, cgn) * 256.0;
        if (c < bc) { bc = c; ba = a; }
    }
    band = std::min(ba, rgn);
    return true;
}

template <typename T, typename TO>
void launch_linear_x1d(const ConvGemmDev& e_in, int rgn, int cgn, int band, hipStream_t s) {
    ConvGemmDev e = e_in;
    e.Tm = rgn; e.Tn = cgn; e.RT = band;
    const long tiles = (long)rgn * cgn;
    const dim3 grid((unsigned)(((tiles + 7) / 8) * 8));
    prof_set_kernel("linear_x1d_kernel<T, TO>", type_label<T>(), type_label<TO>());
    hipLaunchKernelGGL((linear_x1d_kernel<T, TO>), grid, dim3(768), 0, s, e);
    MI_HIP(hipGetLastError());
}
template void launch_linear_x1d<f16, f16>(const ConvGemmDev&, int, int, int, hipStream_t);
template void launch_linear_x1d<f16, float>(const ConvGemmDev&, int, int, int, hipStream_t);
template void launch_linear_x1d<bf16, bf16>(const ConvGemmDev&, int, int, int, hipStream_t);
template void launch_linear_x1d<bf16, float>(const ConvGemmDev&, int, int, int, hipStream_t);

}  // namespace mi
