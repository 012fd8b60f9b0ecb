// gemm_sk.hip — stream-K main loop for the DiT linear layers (plain GEMM: one tap, one group, batch flattened into M).
//
// Why: one utterance gives M = 2252 rows, i.e. 144 / 288 / 432 tiles of 128x128 for N = 1024 / 2048 / 3072 on 256 CUs.
// One tile per workgroup leaves 44 % of the chip idle (144 tiles) or runs a ragged second round; 64x64 tiles balance
// better (576 ... 1728 workgroups) but each wave then owns only 16 MFMAs per K chunk, too few to cover the L2 / MALL
// round trip of the next chunk's DMA (PMC, fp32: SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) = 0.63 - 0.69).
// Here the launch is `P` persistent workgroups (one per CU with the three-stage ring, two per CU with two stages); the
// (tile, K chunk) iteration space is cut into P equal contiguous ranges, so every workgroup issues the same number of
// MFMAs whatever the tile count, with 64 MFMAs per wave per chunk (128x128 tile, 64x64 per wave).
//
// A range that starts or ends inside a tile produces a partial accumulator.  The workgroup that owns a tile's chunk 0
// ("head") finishes the tile: it adds the partials of the following ranges IN RANGE ORDER (bit-reproducible) and runs
// the epilogue.  A range that starts mid-tile computes that piece FIRST and publishes it at once, so by the time the
// head owner gets to its last piece the partials it needs were written long ago.  Ranges are numbered against the
// workgroup ids (see the XCD note in the kernel): the ranges a finisher waits for belong to LOWER workgroup ids, which the
// dispatcher started earlier — the wait cannot deadlock even if the grid is not fully resident.  Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility):
// write-through (sc1) 16-byte slab stores -> s_waitcnt vmcnt(0) in every wave -> barrier -> relaxed agent-scope flag
// store; the finisher polls the flag with relaxed agent loads, then reads the slab with sc1 loads (they bypass L1 and
// are served by L2, which the write-through stores did not leave a stale line in).  The finisher resets the flag, so a
// hipGraph replay of the same launch finds the workspace clean.
#include "common.h"
#include "mfma.h"
#include "gemm_epilogue.h"
#include <type_traits>

namespace mi {

typedef unsigned int sk_u4 __attribute__((ext_vector_type(4)));

// LDS-DMA of 16 bytes per lane, issued from inline asm so that hipcc does not see an LDS write: for the builtin forms
// (global_load_lds and buffer_load ... lds alike) it puts s_waitcnt vmcnt(0) in front of the next ds_read, which drains the
// ring every chunk (seen in the ISA of this kernel; it knows nothing of the counted inline-asm waits).  M0 carries the
// wave-uniform LDS byte address and is saved / restored around the instruction (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void sk_glds16(const void* gsrc, unsigned lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
#endif
}
// the same through a buffer descriptor (buffer_load_dwordx4 ... offen lds): one 32-bit per-lane offset, and the hardware
// range check writes ZEROS for offsets outside [0, num_records) — row tails and dummy chunks need no select
template <typename RSRC>
__device__ __forceinline__ void sk_bufds16(RSRC rsrc, int voff, unsigned lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
#endif
}
__device__ __forceinline__ unsigned sk_lds_addr(const void* p) {
    return (unsigned)(unsigned long)(const __attribute__((address_space(3))) void*)p;
}

// PROD: two extra waves do nothing but issue the LDS-DMA (wave 4: the 16 x-row pieces of a chunk, wave 5: the 16 weight-row
// pieces) and wait for it; the four MFMA waves
// carry no VMEM instruction and no vmcnt wait.  Why: a DMA instruction costs its wave ~100 cycles of issue (M0 write, address
// add, the instruction itself) during which that wave — alone on its SIMD — issues no MFMA: eight per 4096-cycle fp32 chunk
// were the 18 % the "no DMA" ablation recovered.  The producer shares SIMD 0 with wave 0 and is paced by the same barriers.
template <typename T, typename TO, bool LEPI, int NST, bool PROD = false>
__global__ __launch_bounds__(PROD ? 384 : 256, NST <= 2 ? 2 : 1) void linear_sk_kernel(const ConvGemmDev p) {
    using MF = Mfma<T>;
    constexpr int VEC = 16 / (int)sizeof(T), KC = 8 * VEC;      // a tile row is 128 bytes: 64 halfs or 32 floats per K chunk
    constexpr int BM = 128, BN = 128, WM = 64, WN = 64, TM = 2, TN = 2, DJ = 4, PER = 2 * DJ;
    constexpr int TILE = (BM + BN) * KC;
    constexpr int AHEAD = NST - 1;                              // chunks in flight beyond the one being computed
    static_assert(NST >= 2 && NST <= 4, "ring depth");
    static_assert(!PROD || NST == 3, "the producer wave runs two chunks ahead in a three-stage ring");
    // ONE static array (a dynamic `extern __shared__` block makes hipcc put s_waitcnt vmcnt(0) in front of every ds_read
    // that follows an LDS-DMA: the ring then drains every chunk — seen in the ISA, 56 % MFMA duty instead of ~90 %)
    __shared__ __attribute__((aligned(1024))) T smem[NST * TILE];
#if defined(__HIP_DEVICE_COMPILE__)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lk = lane >> 5;
    // XCD-aware ranges.  Workgroup b lands on XCD b % 8 (observed dispatch order; only speed depends on it) and every XCD
    // has its own 4 MiB L2, so the tile list is cut into 8 contiguous groups, one per XCD, and each group is stream-K'd over
    // that XCD's P/8 workgroups: with row tiles fastest a group is a few whole weight panels (x is re-read per XCD), with
    // panels fastest (p.RC = 1) a few whole row tiles (the weights are re-read per XCD) — the host picks the cheaper one.
    // Before, consecutive ranges sat on different XCDs and every L2 streamed every operand from the fabric: the fp32 QKV
    // layer fetched 432 tile-pairs x 1 MB per launch and ran 22 % slower than with the DMA switched off.
    // Inside a group range r sits on workgroup (R-1-r)*8 + y: the pieces a finisher waits for belong to LOWER workgroup
    // ids (dispatched earlier), so the wait cannot deadlock even when the grid is not fully resident; groups hold whole
    // tiles, so nothing is ever awaited across groups.
    const int P = (int)gridDim.x, R = P >> 3;
    const int xg = (int)blockIdx.x & 7;
    const int l = R - 1 - ((int)blockIdx.x >> 3);               // range index inside the group
    const int nch = p.K / KC;
    const int T_all = p.Tm * p.Tn;
    const int tile_lo_g = (int)((long)xg * T_all / 8), tile_hi = (int)((long)(xg + 1) * T_all / 8);
    // Hybrid (p.tail_tiles != 0): when a group holds at least one tile per workgroup, every workgroup first computes
    // `full` WHOLE tiles (tile_lo_g + j*R + l), all of them starting at chunk 0 together — the 32 workgroups of an XCD then
    // walk K in step and share their row / weight panels in L2 (pure stream-K ranges start at unrelated chunks: TCC hit
    // rate 38 % against 82 % for one tile per workgroup) — and only the group's remaining tiles are stream-K'd.
    const int full = p.tail_tiles ? (tile_hi - tile_lo_g) / R : 0;
    const int tile_lo = tile_lo_g + full * R;                   // first stream-K'd tile of the group
    const long I = (long)(tile_hi - tile_lo) * nch;
    long it = (long)l * I / R;
    const long it1 = (long)(l + 1) * I / R;
    const int slot0 = xg * R;                                   // workspace slots / flags of this group
    int dp_done = 0;

    const int kvl0 = (lane & 7) ^ ((lane >> 4) & 7);            // even 8-row DMA groups (swizzle: slot = kv ^ ((row >> 1) & 7))
    const int kvl1 = (lane & 7) ^ ((4 + (lane >> 4)) & 7);      // odd 8-row DMA groups
    const int lrow = lane >> 3;
    const T* xb = (const T*)p.x;
    const T* wg = (const T*)p.w;
    __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)((((long)p.T_in - 1) * p.x_rstride + p.Cin) * (long)sizeof(T)), 0x00020000);
    __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, (int)((long)p.N * p.K * (long)sizeof(T)), 0x00020000);
    constexpr int OOB = 0x7fffff00;                             // beyond every num_records (and OOB + OOB wraps to a value >= 2^31 - 512: still out of range as unsigned)
    const unsigned smem_lds = sk_lds_addr(smem);
    __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.sk_ws, 0, (int)((long)P * BM * BN * 4), 0x00020000);
    int* flags = p.sk_flags;

    while (dp_done < full || it < it1) {
        const bool dp = p.tail_tiles == 2 ? !(it < it1) : dp_done < full;   // a whole tile of the data-parallel phase (mode 2: after the stream-K'd pieces, mode 1: before)
        const int tile_g = dp ? 0 : (int)(it / nch);             // tile inside the stream-K'd part of the group
        const int cb = dp ? 0 : (int)(it - (long)tile_g * nch);
        const int tile = dp ? tile_lo_g + dp_done * R + l : tile_lo + tile_g;
        const int n = dp ? nch : (int)((it1 - it) < (long)(nch - cb) ? (it1 - it) : (long)(nch - cb));
        const int ce = cb + n;
        int nt, mt;
        if (p.RC == 0) { nt = tile / p.Tm; mt = tile - nt * p.Tm; }   // row tiles fastest: a group = whole weight panels
        else { mt = tile / p.Tn; nt = tile - mt * p.Tn; }             // panels fastest: a group = whole row tiles
        const int m0 = mt * BM, n0 = nt * BN;
        // Out-of-range rows and the dummy chunks past the range fetch a page of zeros.
        // per-lane byte offsets of this tile's rows at chunk 0 (loop-invariant); rows past N / T_in are out of range already
        int avo[DJ], bvo[DJ];
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            const int R0 = (wave * DJ + j) * 8;
            const long nn = n0 + R0 + lrow;
            const int kvl = (j & 1) ? kvl1 : kvl0;
            avo[j] = (int)(((long)(m0 + R0 + lrow) * p.x_rstride + kvl * VEC) * (long)sizeof(T));
            bvo[j] = nn < p.N ? (int)((nn * p.K + kvl * VEC) * (long)sizeof(T)) : OOB;
        }
        // one of the 8 DMA instructions of a chunk (j < 4: A row groups, j >= 4: B row groups); chunk >= ce is a dummy that
        // keeps the vmcnt arithmetic uniform (offset out of range: the range check writes zeros, nothing is fetched)
        auto dma_one = [&](int st, int chunk, int j) {
            if (p.dbg & 1) return;                                // tuning: no DMA (stale LDS)
            const unsigned base = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)(st * TILE * (int)sizeof(T)));
            const int cbytes = chunk < ce ? chunk * KC * (int)sizeof(T) : OOB;      // wave-uniform
            const int jj = j & 3;
            const int R0 = (wave * DJ + jj) * 8;
            if (j < 4) sk_bufds16(rsa, (int)((unsigned)avo[jj] + (unsigned)cbytes), base + (unsigned)(R0 * KC * (int)sizeof(T)));
            else sk_bufds16(rsb, (int)((unsigned)bvo[jj] + (unsigned)cbytes), base + (unsigned)((BM + R0) * KC * (int)sizeof(T)));
        };
        auto issue = [&](int st, int chunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) dma_one(st, chunk, j);
        };
        // at a chunk boundary: chunk c+1 has landed; younger DMAs may stay in flight = the later prefetched chunks plus the
        // 6 instructions of the newest one that were already interleaved into this chunk (the prologue passes 8 instead)
        auto wait_landed = [&](bool prologue) {
            if (prologue) {
                if constexpr (AHEAD == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if constexpr (AHEAD == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            } else {
                if constexpr (AHEAD == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // see the AHEAD == 1 note in the loop
                else if constexpr (AHEAD == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
            }
            static_assert(PER == 8, "counted vmcnt immediates");
        };

        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // fragments run one k-vector step ahead of the MFMAs in two register sets
        using FragT = typename std::conditional<sizeof(T) == 4, float4, typename MF::Frag>::type;
        FragT fa[2][TM], fb[2][TN];
        auto ldfrag = [&](int st, int ks, int set) {
            if (p.dbg & 2) return;                                // tuning: no fragment reads
            const T* As = smem + st * TILE;
            const T* Bs = As + BM * KC;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * WM + i * 32 + lr;
                fa[set][i] = *reinterpret_cast<const FragT*>(As + row * KC + (((ks * 2 + lk) ^ ((row >> 1) & 7)) * VEC));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * WN + j * 32 + lr;
                fb[set][j] = *reinterpret_cast<const FragT*>(Bs + row * KC + (((ks * 2 + lk) ^ ((row >> 1) & 7)) * VEC));
            }
        };
        // a quarter of one k-vector step: fp32 = element e of the lanes' float4 (four MFMAs, four independent accumulators),
        // 16-bit = one (i, j) MFMA
        auto mma_q = [&](int set, int q) {
            if constexpr (sizeof(T) == 4) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float a = q == 0 ? fa[set][i].x : q == 1 ? fa[set][i].y : q == 2 ? fa[set][i].z : fa[set][i].w;
                        const float bq = q == 0 ? fb[set][j].x : q == 1 ? fb[set][j].y : q == 2 ? fb[set][j].z : fb[set][j].w;
                        acc[i][j] = MF::mma(a, bq, acc[i][j]);
                    }
            } else {
                acc[q >> 1][q & 1] = MF::mma(fa[set][q >> 1], fb[set][q & 1], acc[q >> 1][q & 1]);
            }
        };
#define SK_SB() __builtin_amdgcn_sched_barrier(0)

        // ---- K loop over chunks [cb, ce): ring of NST stages.  An in-order wave overlaps its own VALU / DS / DMA-issue
        //      instructions with MFMA execution only if they sit BETWEEN MFMAs in program order, so every k-vector step is
        //      Q0 | fragment reads of the next step | Q1 | DMA | Q2 | DMA | Q3  (PMC before: 26 % of the busy cycles had no
        //      MFMA in flight, the eight address-computation + DMA blocks ran in front of the MFMA group) ---------------------
        if constexpr (PROD) {
            if (wave >= 4) {
                const bool pa = wave == 4;                       // wave 4 stages the x rows, wave 5 the weight rows
                // ---- producer: chunk c+2 goes out right after the barrier that ended the reads of chunk c-1 (same stage);
                //      s_waitcnt vmcnt(16) = everything but this wave's newest chunk has landed; then the barrier that publishes
                //      chunk c+1.  Rows past M / N are beyond num_records: the range check writes zeros.
                const int a_ev = (int)(((long)(m0 + lrow) * p.x_rstride + kvl0 * VEC) * (long)sizeof(T));
                const int a_od = (int)(((long)(m0 + 8 + lrow) * p.x_rstride + kvl1 * VEC) * (long)sizeof(T));
                const int b_ev = (int)(((long)(n0 + lrow) * p.K + kvl0 * VEC) * (long)sizeof(T));
                const int b_od = (int)(((long)(n0 + 8 + lrow) * p.K + kvl1 * VEC) * (long)sizeof(T));
                const int a_step = (int)((long)16 * p.x_rstride * (long)sizeof(T)), b_step = (int)((long)16 * p.K * (long)sizeof(T));
                auto issue_all = [&](int stg, int chunk) __attribute__((always_inline)) {
                    if (p.dbg & 1) return;
                    const unsigned base = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)(stg * TILE * (int)sizeof(T)));
                    const int cbytes = chunk < ce ? chunk * KC * (int)sizeof(T) : OOB;
                    if (pa) {
#pragma unroll
                        for (int g = 0; g < 8; ++g) {
                            sk_bufds16(rsa, (int)((unsigned)a_ev + (unsigned)(g * a_step) + (unsigned)cbytes), base + (unsigned)((2 * g) * 8 * KC * (int)sizeof(T)));
                            sk_bufds16(rsa, (int)((unsigned)a_od + (unsigned)(g * a_step) + (unsigned)cbytes), base + (unsigned)((2 * g + 1) * 8 * KC * (int)sizeof(T)));
                        }
                    } else {
#pragma unroll
                        for (int g = 0; g < 8; ++g) {
                            sk_bufds16(rsb, (int)((unsigned)b_ev + (unsigned)(g * b_step) + (unsigned)cbytes), base + (unsigned)((BM + (2 * g) * 8) * KC * (int)sizeof(T)));
                            sk_bufds16(rsb, (int)((unsigned)b_od + (unsigned)(g * b_step) + (unsigned)cbytes), base + (unsigned)((BM + (2 * g + 1) * 8) * KC * (int)sizeof(T)));
                        }
                    }
                };
                issue_all(0, cb); issue_all(1, cb + 1);
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                int stp = 2;
                for (int c = 0; c + 1 < n; ++c) {
                    issue_all(stp, cb + c + 2);
                    if (++stp == NST) stp = 0;
                    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
            } else {
                __builtin_amdgcn_s_barrier();                   // chunk cb has landed
                ldfrag(0, 0, 0);
                int st = 0;
                for (int c = 0; c < n; ++c) {
                    int stn = st + 1; if (stn == NST) stn = 0;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const int set = ks & 1;
                        if (ks == 3 && c + 1 < n) {
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            __builtin_amdgcn_s_barrier();       // chunk c+1 published, stage of chunk c released
                        }
                        SK_SB(); mma_q(set, 0); SK_SB();
                        if (ks < 3) ldfrag(st, ks + 1, set ^ 1);
                        else if (c + 1 < n) ldfrag(stn, 0, 0);
                        SK_SB(); mma_q(set, 1); SK_SB();
                        SK_SB(); mma_q(set, 2); SK_SB();
                        SK_SB(); mma_q(set, 3); SK_SB();
                    }
                    st = stn;
                }
            }
        } else {
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) issue(a, cb + a);
        wait_landed(true);
        __builtin_amdgcn_s_barrier();
        ldfrag(0, 0, 0);
        int st = 0, st_issue = AHEAD % NST;                     // stage of chunk c ; stage chunk c+AHEAD goes to (= stage of chunk c-1)
        for (int c = 0; c < n; ++c) {
            int stn = st + 1; if (stn == NST) stn = 0;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int set = ks & 1;
                if (ks == 3 && c + 1 < n) {
                    // publish chunk c+1 before the LAST step of chunk c (its fragments are in registers): the barrier
                    // latency and the first fragment reads of chunk c+1 run under those MFMAs
                    wait_landed(false);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                // DMA slots of this step: ring (AHEAD >= 2): two per step, the last two of a chunk after the boundary wait
                // (the counted wait allows for them); two stages (AHEAD == 1): chunk c+1 must be complete at the boundary
                // before step 3, so its eight instructions go into steps 0-2
                constexpr bool RING = AHEAD > 1;
                SK_SB(); mma_q(set, 0); SK_SB();
                if (ks < 3) ldfrag(st, ks + 1, set ^ 1);
                else if (c + 1 < n) ldfrag(stn, 0, 0);
                if (!RING && ks < 2) dma_one(st_issue, cb + c + AHEAD, 3 * ks);
                SK_SB(); mma_q(set, 1); SK_SB();
                if (RING) dma_one(st_issue, cb + c + AHEAD, 2 * ks);
                else if (ks < 2) dma_one(st_issue, cb + c + AHEAD, 3 * ks + 1);
                else if (ks == 2) dma_one(st_issue, cb + c + AHEAD, 6);
                SK_SB(); mma_q(set, 2); SK_SB();
                if (RING) dma_one(st_issue, cb + c + AHEAD, 2 * ks + 1);
                else if (ks < 2) dma_one(st_issue, cb + c + AHEAD, 3 * ks + 2);
                else if (ks == 2) dma_one(st_issue, cb + c + AHEAD, 7);
                SK_SB(); mma_q(set, 3); SK_SB();
            }
            if (++st_issue == NST) st_issue = 0;
            st = stn;
        }
        }
#undef SK_SB
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // trailing dummies have landed: the ring may be reused
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        // ---- partial tile: publish (tail / middle piece) or collect (head piece) -------------------------------------------
        const int slot_lane = (wave * 16) * 64 + lane;          // + (i*2+j)*4*64 + q*64 : 16-byte units inside a 64 KB slot
        if (p.dbg & 4) { if (dp) ++dp_done; else it += n; continue; }                    // tuning: no fix-up, no epilogue
        const bool worker = !PROD || wave < 4;                   // the producer wave only takes part in the barriers below
        if (cb > 0) {
            if (worker)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        sk_u4 v;
                        v.x = __float_as_uint(acc[i][j][4 * q]); v.y = __float_as_uint(acc[i][j][4 * q + 1]);
                        v.z = __float_as_uint(acc[i][j][4 * q + 2]); v.w = __float_as_uint(acc[i][j][4 * q + 3]);
                        const int unit = slot_lane + ((i * 2 + j) * 4 + q) * 64;
                        __builtin_amdgcn_raw_buffer_store_b128(v, rsw, ((slot0 + l) * (BM * BN / 4) + unit) * 16, 0, 16 /* sc1: write-through */);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flags + slot0 + l, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (ce < nch) {
                int cov = ce;
                for (int q_l = l + 1; cov < nch; ++q_l) {       // the pieces that follow, in range order
                    const long q0 = (long)q_l * I / R, q1 = (long)(q_l + 1) * I / R;
                    if (q1 == q0) continue;                     // an empty range publishes nothing
                    if (tid == 0) {
                        while (__hip_atomic_load(flags + slot0 + q_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(2);
                    }
                    __syncthreads();
                    sk_u4 v[TM * TN * 4];
                    if (worker) {
#pragma unroll
                    for (int u = 0; u < TM * TN * 4; ++u)
                        v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsw, ((slot0 + q_l) * (BM * BN / 4) + slot_lane + u * 64) * 16, 0, 16 /* sc1 */);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const sk_u4 w = v[(i * 2 + j) * 4 + q];
                                acc[i][j][4 * q] += __uint_as_float(w.x); acc[i][j][4 * q + 1] += __uint_as_float(w.y);
                                acc[i][j][4 * q + 2] += __uint_as_float(w.z); acc[i][j][4 * q + 3] += __uint_as_float(w.w);
                            }
                    }
                    __syncthreads();                            // every lane holds its share: the slot may be recycled
                    if (tid == 0) __hip_atomic_store(flags + slot0 + q_l, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    cov += (int)((q1 - q0) < (long)(nch - cov) ? (q1 - q0) : (long)(nch - cov));
                }
            }
            if (worker) {
            if constexpr (LEPI) {
                constexpr int ERT = 2;
                float* stage = reinterpret_cast<float*>(smem) + wave * (ERT * 32 * WN);
                bool done = false;
                if constexpr (sizeof(TO) == 2) {
                    if (p.epi == EPI_QKV_ROPE) { gemm_epilogue_qkv_lds<TO>(acc, p, m0, n0, 0, wm, wn, lr, lk, stage); done = true; }
                }
                if (!done) gemm_epilogue_lds<TO, TM, TN, WM, WN>(acc, p, m0, n0, 0, 0, wm, wn, lr, lk, stage);
            } else {
                gemm_epilogue<TO, TM, TN, WM, WN>(acc, p, m0, n0, 0, 0, wm, wn, lr, lk);
            }
            }
            __syncthreads();                                    // the staging epilogue read the ring's LDS
        }
        if (dp) ++dp_done; else it += n;
    }
#endif
}

// stages: 0 = automatic (fp32: three-stage ring, one workgroup per CU; 16-bit: two stages, two workgroups per CU).
// Round 3 pruned the A/B losers: the producer-wave form (PROD = true: 93.0 / 88.3 vs 83.9 us) is no longer
// instantiated, the whole-tiles-first hybrid (87.0 / 87.2 vs 83.7 us) is no longer reachable (tail_tiles stays 0).
template <typename T, typename TO>
void launch_linear_sk(const ConvGemmDev& e, int stages, hipStream_t s) {
    constexpr int KC = 128 / (int)sizeof(T);
    const int nst = stages ? stages : (sizeof(T) == 4 ? 3 : 2);
    (void)KC;
    int dev = 0, cus = 256;
    MI_HIP(hipGetDevice(&dev));
    {
        static int cu_count[16] = {0};
        if (!cu_count[dev & 15]) { hipDeviceProp_t pr; MI_HIP(hipGetDeviceProperties(&pr, dev)); cu_count[dev & 15] = pr.multiProcessorCount; }
        cus = cu_count[dev & 15];
    }
    const int P = std::min(cus * (nst <= 2 ? 2 : 1), e.sk_slots) & ~7;         // a multiple of the 8 XCD groups
    const dim3 grid(P);
#define SK_LAUNCH(LE, NS)                                                                                              \
    do {                                                                                                               \
        auto kfn = linear_sk_kernel<T, TO, LE, NS>;                                                                     \
        prof_set_kernel("linear_sk_kernel<T, TO, " #LE ", " #NS ">", type_label<T>(), type_label<TO>());               \
        hipLaunchKernelGGL(kfn, grid, dim3(256), 0, s, e);                                                             \
    } while (0)
    if (e.lds_epi) { if (nst == 2) SK_LAUNCH(true, 2); else if (nst == 3) SK_LAUNCH(true, 3); else SK_LAUNCH(true, 4); }
    else { if (nst == 2) SK_LAUNCH(false, 2); else if (nst == 3) SK_LAUNCH(false, 3); else SK_LAUNCH(false, 4); }
#undef SK_LAUNCH
    MI_HIP(hipGetLastError());
}

template void launch_linear_sk<float, float>(const ConvGemmDev&, int, hipStream_t);
template void launch_linear_sk<f16, f16>(const ConvGemmDev&, int, hipStream_t);
template void launch_linear_sk<f16, float>(const ConvGemmDev&, int, hipStream_t);
template void launch_linear_sk<bf16, bf16>(const ConvGemmDev&, int, hipStream_t);
template void launch_linear_sk<bf16, float>(const ConvGemmDev&, int, hipStream_t);

}  // namespace mi
