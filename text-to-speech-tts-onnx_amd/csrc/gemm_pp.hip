// gemm_pp.hip — 16-bit implicit-GEMM main loop v4 for gfx950: 256-row tiles, 8 waves in two groups that run one
// slot apart ("ping-pong"), 64-deep K chunks (128-byte rows) in two LDS stages that are refilled PIECEWISE by LDS-DMA as
// soon as a piece has been read for the last time.
//
// Same contraction, operands, LDS image, K order and epilogue as conv_gemm_dma3_kernel (gemm_conv.hip); what changes is
// the schedule.  In the v3 kernel the two waves of a SIMD run the same code in lockstep (both wait at the chunk barrier,
// both fetch fragments, both run their MFMAs) and only one chunk's DMA is ever in flight.  Here a K chunk is four slots
//     M0 = ds_read B (all 64 k) + the LOW half of this wave's A rows | C0 = MFMAs on them |
//     M1 = ds_read the HIGH half of the A rows                       | C1 = MFMAs (B fragments are kept)
// each closed by a workgroup barrier, and waves 4-7 execute ONE extra barrier before the loop (waves 0-3 one after it):
// in every slot one wave of each SIMD is in a C slot (matrix pipe) while the other is in an M slot (LDS + DMA issue).
//
// Slots are numbered globally; group 0 runs {M0, C0, M1, C1} of chunk q in slots 4q..4q+3, group 1 in 4q+1..4q+4.
//   last reads of chunk q:  B, A-low: slot 4q+1 (group 1's M0)      A-high: slot 4q+3 (group 1's M1)
//   refill with chunk q+2:  B, A-low: issued in slot 4q+2           A-high: issued in slot 4q+4      (after the barrier
//                           that closes the last reading slot, whose reads were retired by lgkmcnt(0)  -> no WAR)
//   first reads of q+2:     B, A-low: slot 4q+8                     A-high: slot 4q+10
//   waits:                  every wave executes s_waitcnt vmcnt(NB+4) at the end of every ODD slot: the pieces issued
//                           after the one needed next are exactly {A-high (2), B + A-low (NB+2)} in either order, so the
//                           needed piece has landed before the barrier that precedes its first read  -> no RAW,
// and every piece has six slots (1.5 chunk periods) to land while the next pieces are already queued behind it: the
// DMA engine never idles (the two-stage whole-chunk version of this kernel had one chunk in flight and was bound by the
// DMA round trip).  Pieces past the end of K fetch the zero page so the vmcnt arithmetic stays uniform.
//
// Measured while building this (tools/ubench/, gpurun_out/): the L2 -> LDS rate of this access pattern is bound per
// cache LINE touched: 128-byte row pieces reach 78-103 GB/s per CU, 64-byte pieces (32-deep chunks) only 51-68 — a
// 32-deep four-stage ring was DMA-bound and slower than v3.  MFMA throughput is data-dependent (power): the same
// kernel runs 1.24 PF on zero operands and 0.97 PF on random ones.
#include "common.h"
#include "mfma.h"
#include "gemm_epilogue.h"

namespace mi {

template <typename T, typename TO, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(512) void conv_gemm_pp_kernel(const ConvGemmDev p) {
    using MF = Mfma<T>;
    constexpr int KC = 64, TM = WM / 32, TN = WN / 32, WGN = BN / WN, TH = TM / 2;
    static_assert((BM / WM) * WGN == 8 && TM % 2 == 0 && BM == 256, "eight waves, even row tiles");
    constexpr int TILE = (BM + BN) * KC;                  // elements per stage
    constexpr int NB = BN / 64;                           // B 8-row DMA groups per wave per chunk (A: 2 low + 2 high)
    constexpr int GPB = WM / 16;                          // low (or high) 8-row groups per WM-row block
    __shared__ __attribute__((aligned(1024))) T smem[2 * TILE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN, lr = lane & 31, lk = lane >> 5;
    const int L = blockIdx.x;
    const int nt = L / p.RT, rowt = L - nt * p.RT;        // row tiles fastest: neighbours share the weight panel
    const int b = rowt / p.Tm, mt = rowt - b * p.Tm;
    const int m0 = mt * BM, n0 = nt * BN;
    const int g = blockIdx.y;
    const T* xb = (const T*)p.x + (long)b * p.x_bstride + (long)g * p.x_goff;
    const T* wg = (const T*)p.w + (long)g * p.N * p.K;
    const T* zero = (const T*)p.zero;

    const int lrow = lane >> 3;
    const int kv0 = (lane & 7) ^ ((lane >> 4) & 7);               // even 8-row groups   (slot = kv ^ ((row >> 1) & 7))
    const int kv1 = (lane & 7) ^ ((4 + (lane >> 4)) & 7);         // odd 8-row groups
    const int ntaps = p.K / p.Cin;
    const int nchunks = ntaps * ((p.Cin + KC - 1) / KC);

    int itap = 0, ic0 = 0, iq = 0;                                // cursor: the chunk whose pieces are issued next
    auto dma_a = [&](T* base, int R0) {                           // 8 A rows R0..R0+7 of the tile
        const int ci = ic0 + (((R0 >> 3) & 1) ? kv1 : kv0) * 8;
        const int t = m0 + R0 + lrow + itap * p.dil - p.pad;
        const T* src = (iq < nchunks && ci < p.Cin && t >= 0 && t < p.T_in) ? xb + (long)t * p.x_rstride + ci : zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (lds_void*)(base + R0 * KC), 16, 0, 0);
    };
    auto issue_blo = [&](int st) {                                // B (all rows) + the low halves of the A row blocks
        if (p.dbg & 1) return;
        T* base = smem + st * TILE;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int R0 = (wave * NB + j) * 8;
            const int ci = ic0 + (((R0 >> 3) & 1) ? kv1 : kv0) * 8;
            const int n = n0 + R0 + lrow;
            const T* src = (iq < nchunks && ci < p.Cin && n < p.N) ? wg + (long)n * p.K + (long)itap * p.Cin + ci : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (lds_void*)(base + (BM + R0) * KC), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) { const int a = wave * 2 + j; dma_a(base, (a / GPB) * WM + (a % GPB) * 8); }
    };
    auto issue_hi = [&](int st) {                                 // the high halves of the A row blocks; advances the cursor
        if (!(p.dbg & 1)) {
            T* base = smem + st * TILE;
#pragma unroll
            for (int j = 0; j < 2; ++j) { const int a = wave * 2 + j; dma_a(base, (a / GPB) * WM + WM / 2 + (a % GPB) * 8); }
        }
        ++iq;
        if (++itap >= ntaps) { itap = 0; ic0 += KC; }             // K order = (channel chunk, tap)
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses (elements, relative to the stage base) for k-step 0: row * 64 + ((lk ^ ((row >> 1) & 7)) << 3);
    // k-step ks toggles bits 1-2 of the slot: offset ^ (ks << 4)
    int aoff[TM], boff[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int row = wm * WM + i * 32 + lr; aoff[i] = row * KC + ((lk ^ ((row >> 1) & 7)) << 3); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int row = BM + wn * WN + j * 32 + lr; boff[j] = row * KC + ((lk ^ ((row >> 1) & 7)) << 3); }

    typename MF::Frag fa[4][TH] = {}, fb[4][TN] = {};
    auto reads = [&](int st, int h) {                             // h = 0: B + low A rows, h = 1: high A rows
        if (p.dbg & 4) return;
        const T* sb = smem + st * TILE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (h == 0) {
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[ks][j] = *reinterpret_cast<const typename MF::Frag*>(sb + (boff[j] ^ (ks << 4)));
            }
#pragma unroll
            for (int i = 0; i < TH; ++i) fa[ks][i] = *reinterpret_cast<const typename MF::Frag*>(sb + (aoff[h * TH + i] ^ (ks << 4)));
        }
    };
    auto mmas = [&](int h) {
        if (p.dbg & 2) return;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < TH; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[h * TH + i][j] = MF::mma(fa[ks][i], fb[ks][j], acc[h * TH + i][j]);
        __builtin_amdgcn_s_setprio(0);
    };
#define PP_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define PP_VMW() do { if constexpr (NB == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); } while (0)
    static_assert(NB == 4 || NB == 3, "vmcnt immediates");

    issue_blo(0); issue_hi(0);                                    // chunk 0
    issue_blo(1);                                                 // chunk 1: B + A-low
    PP_VMW();                                                     // B + A-low of chunk 0 (this wave's share)
    __builtin_amdgcn_s_barrier();
    if (wave < 4) {
        for (int q = 0; q < nchunks; ++q) {
            const int st = q & 1;
            issue_hi(st ^ 1);                                     // slot 4q   : A-high of chunk q+1
            reads(st, 0); PP_LGKM0(); PP_BAR();
            mmas(0); PP_VMW(); PP_BAR();                          // slot 4q+1
            issue_blo(st);                                        // slot 4q+2 : B + A-low of chunk q+2
            reads(st, 1); PP_LGKM0(); PP_BAR();
            mmas(1); PP_VMW(); PP_BAR();                          // slot 4q+3
        }
        PP_BAR();                                                 // barrier counts of the two groups match again
    } else {
        issue_hi(1);                                              // slot 0    : A-high of chunk 1
        PP_BAR();                                                 // group 1 runs one slot behind
        for (int q = 0; q < nchunks; ++q) {
            const int st = q & 1;
            reads(st, 0); PP_LGKM0(); PP_VMW(); PP_BAR();         // slot 4q+1
            issue_blo(st);                                        // slot 4q+2 : B + A-low of chunk q+2
            mmas(0); PP_BAR();
            reads(st, 1); PP_LGKM0(); PP_VMW(); PP_BAR();         // slot 4q+3
            issue_hi(st);                                         // slot 4q+4 : A-high of chunk q+2
            mmas(1); PP_BAR();
        }
    }
#undef PP_BAR
#undef PP_LGKM0
#undef PP_VMW
    gemm_epilogue<TO, TM, TN, WM, WN>(acc, p, m0, n0, b, g, wm, wn, lr, lk);
}

// bn = 256: 256x256 tile (wave tile 128x64) ; bn = 192: 256x192 tile (wave tile 64x96)
template <typename T, typename TO>
void launch_conv_gemm_pp(const ConvGemmDev& d, int B, int bn, hipStream_t s) {
    ConvGemmDev e = d;
    e.RC = 0;
    e.Tm = (d.M + 255) / 256; e.Tn = (d.N + bn - 1) / bn; e.RT = B * e.Tm;
    const dim3 grid(e.RT * e.Tn, d.G);
    if (bn == 192) hipLaunchKernelGGL((conv_gemm_pp_kernel<T, TO, 256, 192, 64, 96>), grid, dim3(512), 0, s, e);
    else hipLaunchKernelGGL((conv_gemm_pp_kernel<T, TO, 256, 256, 128, 64>), grid, dim3(512), 0, s, e);
    MI_HIP(hipGetLastError());
}

template void launch_conv_gemm_pp<f16, f16>(const ConvGemmDev&, int, int, hipStream_t);
template void launch_conv_gemm_pp<f16, float>(const ConvGemmDev&, int, int, hipStream_t);
template void launch_conv_gemm_pp<bf16, bf16>(const ConvGemmDev&, int, int, hipStream_t);
template void launch_conv_gemm_pp<bf16, float>(const ConvGemmDev&, int, int, hipStream_t);

}  // namespace mi
