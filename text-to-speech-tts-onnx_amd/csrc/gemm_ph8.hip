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
// half of every wave's columns (local row wc*32 + c <-> tile column wc*64 + b*32 + c).  So one half-tile is read by all
// eight waves in the SAME phase, is dead two phases later, and is re-staged then (LDS-DMA, two 16-byte pieces per lane).
// A wave reads sub-tile A_0 + B_0 in phase 0, B_1 in phase 1, A_1 in phase 2 and nothing in phase 3 (64 fragment registers
// next to the 128 accumulators; keeping both A halves live, which would even the reads out to 8/8/4/4, needs 96 and spills):
//
//   phase (K tile k)   ds_read            MFMA block (8 x 32x32x16)   DMA issued (stage)          vmcnt(10) retires
//   0                  B_0(k), A_0(k)     A_0 x B_0 -> acc[0..1][0]   A_1(k+1) -> (k+1)&1         B_1(k)   (read in phase 1)
//   1                  B_1(k)             A_0 x B_1 -> acc[0..1][1]   B_0(k+2) -> k&1             A_1(k)   (read in phase 2)
//   2                  A_1(k)             A_1 x B_1 -> acc[2..3][1]   A_0(k+2)                    -
//   3                  -                  A_1 x B_0 -> acc[2..3][0]   B_1(k+2)                    B_0, A_0(k+1) (phase 0)
//
// Every phase is  [ds_reads | DMA | counted wait | barrier | lgkmcnt(0) | MFMAs | barrier].  Group 1 runs ONE barrier
// behind group 0, so while one group's four waves issue LDS reads and DMA the other group's four waves keep the four MFMA
// pipes busy, and they swap at every barrier.  One half-tile is issued per phase and every wait is vmcnt(10): the five
// youngest half-tiles (two instructions per wave each) stay in flight, i.e. a half-tile has five phases = 1.25 K tiles to
// land, and no wait in the loop ever drains the queue.
// Hazards (slots = barrier intervals; group 0 reads in slot 2P, group 1 in 2P+1, P = 4k + phase):
//   RAW  the wait of phase P-1 precedes a barrier in both groups (slots 2P-2 and 2P-1); the reads come in slots 2P, 2P+1.
//   WAR  A_0 / B_1 / A_1: read in phase P, retired by the lgkmcnt(0) after that phase's first barrier (slots 2P+1, 2P+2),
//        re-staged in phase P+2 (slots 2P+4, 2P+5).  B_0 is re-staged ONE phase after its reads: they are retired BEFORE
//        phase 0's first barrier (s_waitcnt lgkmcnt(8): the four B reads are issued first).
// K tiles past the end are fetched with an out-of-range offset (the buffer range check writes zeros, nothing is fetched) so
// the counted wait stays uniform; everything is drained before the epilogue reuses the LDS.
//
// The DMA instructions are issued from inline asm (see gemm_sk.hip: hipcc drains vmcnt in front of a ds_read that follows
// a builtin LDS-DMA).  Epilogues: the shared LDS-staged ones (gemm_epilogue.h) on the wave's contiguous 128x64 tile.
#include <atomic>
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

template <typename T, typename TO, bool SPLIT>
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
    // partial-tile workspace of the split tail (see below): 256 KB slabs
    __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.sk_ws, 0, (int)((long)p.sk_slots * 65536L), 0x00020000);
    // Tile list.  The first T_full tiles run one per workgroup over the whole K; XCD-aware order: workgroup b lands on XCD
    // b % 8 and each XCD walks a contiguous range of the list (panels fastest), so the 32 tiles an XCD runs together share
    // a few row panels and the weight panels in its own L2.
    // Split tail: with T tiles on C CUs the last T mod C tiles would occupy a whole extra round at T mod C / C of the chip
    // (284 tiles on 256 CUs: 2 rounds for 1.11 rounds of work).  The host cuts each of those `rem` tiles into S slices of K,
    // S * rem <= C, so the tail costs 1 / S of a round.  Slices 1 .. S-1 publish their accumulators (write-through stores,
    // vmcnt(0), barrier, relaxed agent-scope flag: the gemm_sk.hip hand-off) and the slice-0 workgroup adds them IN SLICE
    // ORDER, resets the flags and runs the epilogue.  Publishers get the lower workgroup ids, and all tail workgroups fit on
    // the chip together, so the wait cannot deadlock.
    const int nk_all = p.K / KC;
    const int T_all = p.Tm * p.Tn, rem = (SPLIT && p.tail_split > 1) ? p.tail_tiles : 0, S = rem > 0 ? p.tail_split : 1;
    const int T_full = T_all - rem, G_full = 8 * ((T_full + 7) >> 3);
    int tile, slice = 0, kbeg = 0, kend = nk_all;
    if ((int)blockIdx.x < G_full) {
        const int xg = (int)blockIdx.x & 7, j_in = (int)blockIdx.x >> 3;
        const int q8 = T_full >> 3, r8 = T_full & 7;
        tile = (xg < r8 ? xg * (q8 + 1) : r8 * (q8 + 1) + (xg - r8) * q8) + j_in;
        if (j_in >= q8 + (xg < r8 ? 1 : 0)) return;
    } else if constexpr (SPLIT) {
        const int u = (int)blockIdx.x - G_full;
        slice = S - 1 - u / rem;
        tile = T_full + u % rem;
        kbeg = slice * nk_all / S; kend = (slice + 1) * nk_all / S;
    } else {
        return;
    }
    int mt, nt;
    if (p.RC == 0) { nt = tile / p.Tm; mt = tile - nt * p.Tm; }
    else { mt = tile / p.Tn; nt = tile - mt * p.Tn; }
    const int m0 = mt * 256, n0 = nt * 256;
    const int nk = kend;                                        // K tiles >= kend are out-of-range dummies

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
        const int kb = (k < nk && !(p.dbg & 8)) ? k * (KC * 2) : OOB;      // wave-uniform ; dbg 8: issue cost only
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
#define PH8_WAIT5() asm volatile("s_waitcnt vmcnt(10)" ::: "memory")

    if (p.dbg & 2) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { fa[0][ks] = Frag{}; fa[1][ks] = Frag{}; fb0[ks] = Frag{}; fb1[ks] = Frag{}; }
    }
    // prologue: the seven half-tiles the phases before the first K tile would have issued, in pipeline order
    dma_half(0, 2, kbeg); dma_half(0, 0, kbeg); dma_half(0, 3, kbeg); dma_half(0, 1, kbeg);
    dma_half(1, 2, kbeg + 1); dma_half(1, 0, kbeg + 1); dma_half(1, 3, kbeg + 1);
    PH8_WAIT5();                                                 // B_0, A_0 of the first K tile
    PH8_BAR();
    if (wr == 1) PH8_BAR();                                     // group 1 runs one barrier behind

    for (int k = kbeg; k < kend; ++k) {
        const int st = (k - kbeg) & 1;
        // phase 0
        rdB(st, 0, fb0); PH8_SB();
        rdA(st, 0); PH8_SB();
        dma_half(st ^ 1, 1, k + 1); PH8_SB();
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");       // the four B_0 reads have retired: B_0 may be re-staged in phase 1
        PH8_WAIT5();                                             // B_1(k)
        PH8_BAR();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PH8_SB(); mm(0, 0, fb0);
        PH8_BAR();
        // phase 1
        rdB(st, 1, fb1); PH8_SB();
        dma_half(st, 2, k + 2); PH8_SB();
        PH8_WAIT5();                                             // A_1(k)
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
        PH8_WAIT5();                                             // B_0, A_0 of K tile k+1
        PH8_BAR();
        mm(1, 0, fb0);
        PH8_BAR();
    }
    if (wr == 0) PH8_BAR();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the trailing out-of-range pieces must not land on the epilogue's staging
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PH8_BAR();
#undef PH8_WAIT5
#undef PH8_SB
#undef PH8_BAR
    if (p.dbg & 4) return;

    if constexpr (SPLIT) if (tile >= T_full && S > 1) {
        const int ti = tile - T_full;
        int tidx = tid;
        asm volatile("" : "+v"(tidx));                          // keeps the 32 slab offsets from being hoisted above the K loop (they spilled an accumulator block there)
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        int* flags = p.sk_flags;
        if (slice > 0) {
            const int q = ti * (S - 1) + slice - 1;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        u4 v;
                        v.x = __float_as_uint(acc[i][j][4 * e]); v.y = __float_as_uint(acc[i][j][4 * e + 1]);
                        v.z = __float_as_uint(acc[i][j][4 * e + 2]); v.w = __float_as_uint(acc[i][j][4 * e + 3]);
                        const int unit = ((i * 2 + j) * 4 + e) * 512 + tidx;
                        __builtin_amdgcn_raw_buffer_store_b128(v, rsw, (q * 16384 + unit) * 16, 0, 16 /* sc1: write-through */);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flags + q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
#pragma unroll
        for (int sl = 1; sl < 4; ++sl) {                        // S <= 4 (host); unrolled: a rolled loop carries the 128 accumulators through phis and spills
            if (sl >= S) break;
            const int q = ti * (S - 1) + sl - 1;
            if (tid == 0) {
                while (__hip_atomic_load(flags + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(2);
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsw, (q * 16384 + (i * 8 + u) * 512 + tidx) * 16, 0, 16 /* sc1 */);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const u4 w = v[j * 4 + e];
                        acc[i][j][4 * e] += __uint_as_float(w.x); acc[i][j][4 * e + 1] += __uint_as_float(w.y);
                        acc[i][j][4 * e + 2] += __uint_as_float(w.z); acc[i][j][4 * e + 3] += __uint_as_float(w.w);
                    }
                __builtin_amdgcn_sched_barrier(0);              // eight loads (32 registers) in flight at a time: all 32 at once spill
            }
            __syncthreads();                                    // every lane holds its share: the slab may be recycled
            if (tid == 0) __hip_atomic_store(flags + q, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    {   // LDS-staged epilogues only (the host sends nothing else here: the direct epilogue on a 128x64 wave tile spills)
        // the staged epilogues take a 64x64 wave tile (two 32-row blocks per pass): the wave's two 64-row halves in turn,
        // written out twice rather than looped (the unroller gives up on a loop around bodies this large, and a rolled loop
        // would index the accumulators dynamically, i.e. through scratch)
        float* stage = reinterpret_cast<float*>(smem) + wave * (2 * 32 * 64);
        // AdaLN fold, consumer side: the (rstd, mean * rstd) of the wave's four 32-row blocks, all requested up front
        float lrs[4] = {0.f, 0.f, 0.f, 0.f}, lmr[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (sizeof(TO) == 2) {
            if (p.ln_stats_in) {
#pragma unroll
                for (int i = 0; i < 4; ++i) ln_rows32(p, (long)p.m_off + m0 + wr * 128 + i * 32, (long)p.m_off + p.M - 1, lr, lk, lrs[i], lmr[i]);
            }
        }
        auto half_out = [&](f32x16 (&q)[2][2], int mh, const float* prs, const float* pmr) __attribute__((always_inline)) {
            if constexpr (sizeof(TO) == 2) {
                if (p.epi == EPI_QKV_ROPE) {
                    if (p.ln_stats_in) gemm_epilogue_qkv_lds<TO, 2, true, true>(q, p, mh, n0, 0, 0, wc, lr, lk, stage, prs, pmr);
                    else gemm_epilogue_qkv_lds<TO>(q, p, mh, n0, 0, 0, wc, lr, lk, stage);
                    return;
                }
                if (p.ln_stats_in) { gemm_epilogue_ln_in<TO, 2, 2, 2, true>(q, p, mh, n0 + wc * 64, lr, lk, stage, prs, pmr); return; }      // AdaLN fold: FF1
            } else {
                if (p.ln_stats_out) { gemm_epilogue_resid_ln<T, 2, 2, 2>(q, p, mh, n0 + wc * 64, lr, lk, stage); return; }  // AdaLN fold: O / FF2
            }
            gemm_epilogue_lds<TO, 2, 2, 64, 64>(q, p, mh, n0, 0, 0, 0, wc, lr, lk, stage);
        };
        {
            f32x16 q[2][2] = {{acc[0][0], acc[0][1]}, {acc[1][0], acc[1][1]}};
            half_out(q, m0 + wr * 128, lrs, lmr);
        }
        {
            f32x16 q[2][2] = {{acc[2][0], acc[2][1]}, {acc[3][0], acc[3][1]}};
            half_out(q, m0 + wr * 128 + 64, lrs + 2, lmr + 2);
        }
    }
#endif
}

static std::atomic<long> g_ph8_split_max = 2, g_ph8_split_min_nk = 24;      // measured: a gain only for the K = 2048 layer (FF2, 32 K tiles), two slices
void ph8_set_split_min_nk(long v) { g_ph8_split_min_nk = v; }
void ph8_set_split_max(long v) { g_ph8_split_max = v < 1 ? 1 : v > 4 ? 4 : v; }       // the kernel's fix-up is unrolled for at most 4 slices

template <typename T, typename TO>
void launch_linear_ph8(const ConvGemmDev& e_in, hipStream_t s) {
    ConvGemmDev e = e_in;
    int dev = 0, cus = 256;
    MI_HIP(hipGetDevice(&dev));
    {
        static int cu_count[16] = {0};
        if (!cu_count[dev & 15]) { hipDeviceProp_t pr; MI_HIP(hipGetDeviceProperties(&pr, dev)); cu_count[dev & 15] = pr.multiProcessorCount; }
        cus = cu_count[dev & 15];
    }
    const int T_all = e.Tm * e.Tn, nk = e.K / 64;
    // split tail: the largest S (<= g_ph8_split_max) with S * rem workgroups on the chip at once, >= 4 K tiles per slice and
    // (S - 1) * rem slabs of 256 KB in the workspace; worth it only when the tail round is mostly empty
    const int rem = T_all % cus;
    int S = 1;
    if (T_all > cus && rem > 0 && rem * 2 <= cus && e.sk_ws && e.sk_flags && nk >= g_ph8_split_min_nk) {
        for (int c = 2; c <= (int)g_ph8_split_max; ++c)
            if (rem * c <= cus && nk / c >= 4 && (long)rem * (c - 1) * 4 <= e.sk_slots) S = c;
    }
    e.tail_tiles = S > 1 ? rem : 0;
    e.tail_split = S;
    const int T_full = T_all - e.tail_tiles;
    const dim3 grid(8 * ((T_full + 7) / 8) + e.tail_tiles * S);
    // the split-tail fix-up is a separate instantiation: its code costs the plain main loop ~7 % (register allocation)
    if (S > 1) {
        auto kfn = linear_ph8_kernel<T, TO, true>;
        prof_set_kernel("linear_ph8_kernel<T, TO, true>", type_label<T>(), type_label<TO>());
        hipLaunchKernelGGL(kfn, grid, dim3(512), 0, s, e);
    } else {
        auto kfn = linear_ph8_kernel<T, TO, false>;
        prof_set_kernel("linear_ph8_kernel<T, TO, false>", type_label<T>(), type_label<TO>());
        hipLaunchKernelGGL(kfn, grid, dim3(512), 0, s, e);
    }
    MI_HIP(hipGetLastError());
}

template void launch_linear_ph8<f16, f16>(const ConvGemmDev&, hipStream_t);
template void launch_linear_ph8<f16, float>(const ConvGemmDev&, hipStream_t);
template void launch_linear_ph8<bf16, bf16>(const ConvGemmDev&, hipStream_t);
template void launch_linear_ph8<bf16, float>(const ConvGemmDev&, hipStream_t);

}  // namespace mi
