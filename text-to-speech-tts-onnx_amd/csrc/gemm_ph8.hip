// gemm_ph8.hip — 256x256 tile, 8 waves, 8 phases per two K tiles: the 16-bit main loop for the DiT linear layers of a
// batch of utterances (M = 2 x U x N_frames rows: 18016 at U = 8), where the 128x128 two-buffer kernel is LDS-read and
// DMA-latency bound (64x64 per wave: one ds_read byte per MFMA cycle; one 64-deep chunk of look-ahead).
//
// Geometry.  512 threads = two GROUPS of four waves; group wr owns tile rows [wr*128, +128), wave wc of a group owns
// columns [wc*64, +64): 128x64 per wave = 4x2 accumulator blocks of 32x32 (v_mfma_f32_32x32x16), 32 MFMAs per K tile
// (BK = 64).  The four SIMDs of the CU each hold one wave of either group.
//
// LDS: two stages x four half-tiles of 128 rows x 128 bytes (16 KB): A_0 A_1 B_0 B_1 = 128 KB.  Half-tile A_a holds, for
// EVERY wave, the a-th 64-row half of its rows (local row wr*64 + r <-> tile row wr*128 + a*64 + r); B_b the b-th 32-column
// half of every wave's columns (local row wc*32 + c <-> tile column wc*64 + b*32 + c).  A wave therefore reads sub-tile
// A_0 + B_0 in phase 0, B_1 in phase 1, A_1 in phase 2 and nothing in phase 3, and a half-tile is dead for ALL waves one
// phase after it was read — it is re-staged (LDS-DMA, two 16-byte pieces per lane) while the rest of the K tile computes:
//
//   phase (K tile k)    ds_read            MFMA block (8 each)        DMA issued (stage)
//   0                   B_0(k), A_0(k)     A_0 x B_0 -> acc[0..1][0]  A_1(k+1) -> stage (k+1)&1
//   1                   B_1(k)             A_0 x B_1 -> acc[0..1][1]  B_0(k+2) -> stage k&1
//   2                   A_1(k)             A_1 x B_1 -> acc[2..3][1]  A_0(k+2)
//   3                   -                  A_1 x B_0 -> acc[2..3][0]  B_1(k+2), then s_waitcnt vmcnt(6)
//
// Every phase is  [ds_reads | DMA | barrier | MFMAs | barrier].  Group 1 runs ONE barrier behind group 0, so while one
// group's four waves issue LDS reads and DMA the other group's four waves keep the four MFMA pipes busy, and they swap at
// every barrier.  The only DMA wait is the counted vmcnt(6) of phase 3 (three half-tiles = 6 instructions per wave stay in
// flight): it retires K tile k+1 completely; the barriers that follow publish it to both groups before phase 0 of k+1.
// Hazards (slots = barrier intervals; group 0 reads in slot 2P, group 1 in 2P+1, P = 4k + phase):
//   WAR  B_0: its reads are retired BEFORE phase 0's first barrier (s_waitcnt lgkmcnt(8): the four B reads are issued first),
//        so both groups are done with it when group 0 issues the DMA in phase 1.  A_0 / B_1 / A_1: read in phase p, retired
//        by the lgkmcnt(0) after that phase's first barrier, re-staged two phases later.
//   RAW  a wave's vmcnt wait precedes a barrier, and every reader passes at least one later barrier before it reads.
// K tiles past the end are fetched with an out-of-range offset (the buffer range check writes zeros, nothing is fetched) so
// the counted wait stays uniform; everything is drained before the epilogue reuses the LDS.
//
// The DMA instructions are issued from inline asm (see gemm_sk.hip: hipcc drains vmcnt in front of a ds_read that follows
// a builtin LDS-DMA).  Epilogues: the shared LDS-staged ones (gemm_epilogue.h) on the wave's contiguous 128x64 tile.
#include "common.h"
#include "mfma.h"
#include "gemm_epilogue.h"

namespace mi {

template <typename RSRC>
__device__ __forceinline__ void ph8_bufds16(RSRC rsrc, int voff, unsigned lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
#endif
}

template <typename T, typename TO>
__global__ __launch_bounds__(512, 1) void linear_ph8_kernel(const ConvGemmDev p) {
    using MF = Mfma<T>;
    using Frag = typename MF::Frag;
    static_assert(sizeof(T) == 2, "16-bit operands");
    constexpr int KC = 64;                                      // K tile: 128 bytes per row
    constexpr int HALF = 128 * KC;                              // elements per half-tile (16 KB)
    constexpr int STAGE = 4 * HALF;                             // A_0 A_1 B_0 B_1
    __shared__ __attribute__((aligned(1024))) T smem[2 * STAGE];
    (void)smem;
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int HALFB = HALF * 2, STAGEB = STAGE * 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3, lr = lane & 31, lk = lane >> 5;

    // descriptors first: values defined after the (uniform, but not provably so) early exit below would be placed in VGPRs
    const T* xb = (const T*)p.x;
    const T* wg = (const T*)p.w;
    __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)((((long)p.T_in - 1) * p.x_rstride + p.Cin) * 2L), 0x00020000);
    __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, (int)((long)p.N * p.K * 2L), 0x00020000);
    // XCD-aware order: workgroup b lands on XCD b % 8; each XCD walks a contiguous range of the tile list (panels fastest),
    // so the 32 tiles an XCD runs together share a few row panels and the weight panels in its own L2
    const int T_all = p.Tm * p.Tn;
    const int xg = (int)blockIdx.x & 7, j_in = (int)blockIdx.x >> 3;
    const int q8 = T_all >> 3, r8 = T_all & 7;
    const int tile = (xg < r8 ? xg * (q8 + 1) : r8 * (q8 + 1) + (xg - r8) * q8) + j_in;
    if (j_in >= q8 + (xg < r8 ? 1 : 0)) return;
    int mt, nt;
    if (p.RC == 0) { nt = tile / p.Tm; mt = tile - nt * p.Tm; }
    else { mt = tile / p.Tn; nt = tile - mt * p.Tn; }
    const int m0 = mt * 256, n0 = nt * 256;
    const int nk = p.K / KC;

    constexpr int OOB = 0x7fffff00;
    const unsigned smem_lds = (unsigned)(unsigned long)(const __attribute__((address_space(3))) void*)smem;

    // DMA: one instruction moves 8 rows x 128 bytes (lane -> row lane/8, 16-byte slot lane%8 holding k-vector
    // slot ^ ((row >> 1) & 7)); a half-tile is 16 row groups = 2 instructions for each of the 8 waves (row group j*8 + wave)
    int avo[2][2], bvo[2][2];                                   // [half][instruction]: byte offset at K tile 0
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rg = j * 8 + wave;
        const int r = rg * 8 + (lane >> 3);                     // local row inside the half-tile
        const int kvl = (lane & 7) ^ (((rg & 1) * 4 + (lane >> 4)) & 7);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int trow = (r >> 6) * 128 + h * 64 + (r & 63);
            const int tcol = (r >> 5) * 64 + h * 32 + (r & 31);
            const long m = (long)m0 + trow, n = (long)n0 + tcol;
            avo[h][j] = m < p.M ? (int)((m * p.x_rstride + kvl * 8) * 2L) : OOB;
            bvo[h][j] = n < p.N ? (int)((n * p.K + kvl * 8) * 2L) : OOB;
        }
    }
    // half-tile ids: 0 = A_0, 1 = A_1, 2 = B_0, 3 = B_1
    auto dma_half = [&](int st, int h, int k) __attribute__((always_inline)) {
        if (p.dbg & 1) return;
        const int kb = k < nk ? k * (KC * 2) : OOB;            // wave-uniform
        const unsigned base = smem_lds + (unsigned)(st * STAGEB + h * HALFB + wave * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int vo = h < 2 ? avo[h][j] : bvo[h - 2][j];
            const unsigned dst = __builtin_amdgcn_readfirstlane(base + (unsigned)(j * 8192));
            if (h < 2) ph8_bufds16(rsa, (int)((unsigned)vo + (unsigned)kb), dst);
            else ph8_bufds16(rsb, (int)((unsigned)vo + (unsigned)kb), dst);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses: row-local byte offset + swizzled k-vector ((ks*2 + lk) ^ ((row >> 1) & 7)) * 16
    const int sw = (lr >> 1) & 7;
    int kofs[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kofs[ks] = ((ks * 2 + lk) ^ sw) * 8;       // in elements
    const int arow = (wr * 64 + lr) * KC, brow = (wc * 32 + lr) * KC;

    Frag fa[2][4], fb0[4], fb1[4];
    auto rdA = [&](int st, int a) __attribute__((always_inline)) {
        if (p.dbg & 2) return;
        const T* base = smem + st * STAGE + a * HALF + arow;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fa[ii][ks] = *reinterpret_cast<const Frag*>(base + ii * 32 * KC + kofs[ks]);
    };
    auto rdB = [&](int st, int b, Frag (&fb)[4]) __attribute__((always_inline)) {
        if (p.dbg & 2) return;
        const T* base = smem + st * STAGE + (2 + b) * HALF + brow;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb[ks] = *reinterpret_cast<const Frag*>(base + kofs[ks]);
    };
    auto mm = [&](int a, int b, Frag (&fb)[4]) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) acc[a * 2 + ii][b] = MF::mma(fa[ii][ks], fb[ks], acc[a * 2 + ii][b]);
        __builtin_amdgcn_s_setprio(0);
    };
#define PH8_SB() __builtin_amdgcn_sched_barrier(0)
#define PH8_BAR() do { PH8_SB(); __builtin_amdgcn_s_barrier(); PH8_SB(); } while (0)

    if (p.dbg & 2) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { fa[0][ks] = Frag{}; fa[1][ks] = Frag{}; fb0[ks] = Frag{}; fb1[ks] = Frag{}; }
    }
    // prologue: K tile 0 complete, and the three half-tiles of K tile 1 that phases 1-3 of "tile -1" would have staged
    dma_half(0, 2, 0); dma_half(0, 0, 0); dma_half(0, 3, 0); dma_half(0, 1, 0);
    dma_half(1, 2, 1); dma_half(1, 0, 1); dma_half(1, 3, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    PH8_BAR();
    if (wr == 1) PH8_BAR();                                     // group 1 runs one barrier behind

    for (int k = 0; k < nk; ++k) {
        const int st = k & 1;
        // phase 0
        rdB(st, 0, fb0); PH8_SB();
        rdA(st, 0); PH8_SB();
        dma_half(st ^ 1, 1, k + 1); PH8_SB();
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");       // the four B_0 reads have retired: B_0 may be re-staged in phase 1
        PH8_BAR();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PH8_SB(); mm(0, 0, fb0);
        PH8_BAR();
        // phase 1
        rdB(st, 1, fb1); PH8_SB();
        dma_half(st, 2, k + 2);
        PH8_BAR();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PH8_SB(); mm(0, 1, fb1);
        PH8_BAR();
        // phase 2
        rdA(st, 1); PH8_SB();
        dma_half(st, 0, k + 2);
        PH8_BAR();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PH8_SB(); mm(1, 1, fb1);
        PH8_BAR();
        // phase 3
        dma_half(st, 3, k + 2); PH8_SB();
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");         // K tile k+1 has landed (this wave's pieces)
        PH8_BAR();
        mm(1, 0, fb0);
        PH8_BAR();
    }
    if (wr == 0) PH8_BAR();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the trailing out-of-range pieces must not land on the epilogue's staging
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PH8_BAR();
#undef PH8_SB
#undef PH8_BAR
    if (p.dbg & 4) return;

    {   // LDS-staged epilogues only (the host sends nothing else here: the direct epilogue on a 128x64 wave tile spills)
        // the staged epilogues take a 64x64 wave tile (two 32-row blocks per pass): the wave's two 64-row halves in turn,
        // written out twice rather than looped (the unroller gives up on a loop around bodies this large, and a rolled loop
        // would index the accumulators dynamically, i.e. through scratch)
        float* stage = reinterpret_cast<float*>(smem) + wave * (2 * 32 * 64);
        auto half_out = [&](f32x16 (&q)[2][2], int mh) __attribute__((always_inline)) {
            if constexpr (sizeof(TO) == 2) {
                if (p.epi == EPI_QKV_ROPE) { gemm_epilogue_qkv_lds<TO>(q, p, mh, n0, 0, 0, wc, lr, lk, stage); return; }
            }
            gemm_epilogue_lds<TO, 2, 2, 64, 64>(q, p, mh, n0, 0, 0, 0, wc, lr, lk, stage);
        };
        {
            f32x16 q[2][2] = {{acc[0][0], acc[0][1]}, {acc[1][0], acc[1][1]}};
            half_out(q, m0 + wr * 128);
        }
        {
            f32x16 q[2][2] = {{acc[2][0], acc[2][1]}, {acc[3][0], acc[3][1]}};
            half_out(q, m0 + wr * 128 + 64);
        }
    }
#endif
}

template <typename T, typename TO>
void launch_linear_ph8(const ConvGemmDev& e, hipStream_t s) {
    const int T_all = e.Tm * e.Tn;
    const dim3 grid(8 * ((T_all + 7) / 8));
    auto kfn = linear_ph8_kernel<T, TO>;
    prof_set_kernel("linear_ph8_kernel<T, TO>", type_label<T>(), type_label<TO>());
    hipLaunchKernelGGL(kfn, grid, dim3(512), 0, s, e);
    MI_HIP(hipGetLastError());
}

template void launch_linear_ph8<f16, f16>(const ConvGemmDev&, hipStream_t);
template void launch_linear_ph8<f16, float>(const ConvGemmDev&, hipStream_t);
template void launch_linear_ph8<bf16, bf16>(const ConvGemmDev&, hipStream_t);
template void launch_linear_ph8<bf16, float>(const ConvGemmDev&, hipStream_t);

}  // namespace mi
