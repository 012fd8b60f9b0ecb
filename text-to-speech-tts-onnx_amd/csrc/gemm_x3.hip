// gemm_x3.hip — fp32 linear layers on the 16-bit matrix cores by an exact three-way split ("bf16x3").
//
// Every fp32 operand value is written as the sum of three bf16 values  a = a1 + a2 + a3  (a1 = bf16(a), a2 = bf16(a - a1),
// a3 = bf16(a - a1 - a2); the subtractions are exact in fp32, so the three pieces carry 24 significant bits: the whole
// fp32 mantissa).  A product of two bf16 numbers is exact in fp32, and
//     a * b = a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1) + O(2^-24 |a b|),
// so six v_mfma_f32_32x32x16_bf16 per 32x32x16 block, accumulated in fp32 exactly like v_mfma_f32_32x32x2_f32 accumulates,
// reproduce the fp32 product to within the rounding of the fp32 accumulation itself: this is fp32 arithmetic carried by
// the bf16 pipes, not a reduced-precision mode.  (The dropped terms a2 b3, a3 b2, a3 b3 are below 2^-24 of the product.)
// Cost: 6 x 32 cycles per 32x32x16 block against 8 x 64 cycles for the native fp32 MFMA — 2.7x fewer matrix-core cycles,
// and far fewer joules, which is what bounds the native fp32 kernels on this chip.
//
// The weights are split once at load ([3][N][K] bf16 planes, `split3_planes`); the activations arrive as fp32 rows,
// are staged by LDS-DMA as fp32 (128-byte rows of 32 floats, the swizzle of the other kernels) and are split in
// registers on their way from LDS to the MFMA (11 VALU instructions per pair of values, v_cvt_pk_bf16_f32 rounding to
// nearest even), interleaved with the MFMAs of the previous fragment.
//
// Structure: the stream-K frame of gemm_sk.hip (one persistent workgroup per CU, three-stage LDS ring, equal (tile, K
// chunk) ranges per XCD group, range-ordered fix-up through write-through slabs and relaxed flags), 128x128 tiles, 64x64
// per wave, 32-deep K chunks.  LDS per stage: x rows 16 KB + 3 weight planes x 8 KB (64-byte rows, 16-byte slot
// kv ^ ((row >> 2) & 3): a ds_read_b128 lane group covers the 64 banks once).
#include "common.h"
#include "mfma.h"
#include "gemm_epilogue.h"
#include "x3_split.h"

namespace mi {

template <typename RSRC>
__device__ __forceinline__ void x3_bufds16(RSRC rsrc, int voff, unsigned lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
#endif
}

// weights: fp32 [rows][K] -> planes [3][rows][K] bf16
__global__ __launch_bounds__(256) void split3_planes_kernel(const float* __restrict__ w, __bf16* __restrict__ out, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i >= n) return;
    const float x = w[i], y = i + 1 < n ? w[i + 1] : 0.f;
    unsigned p1, p2, p3;
    x3_split_pair(x, y, p1, p2, p3);
    unsigned short* o = reinterpret_cast<unsigned short*>(out);
    o[i] = (unsigned short)(p1 & 0xffff); o[n + i] = (unsigned short)(p2 & 0xffff); o[2 * n + i] = (unsigned short)(p3 & 0xffff);
    if (i + 1 < n) { o[i + 1] = (unsigned short)(p1 >> 16); o[n + i + 1] = (unsigned short)(p2 >> 16); o[2 * n + i + 1] = (unsigned short)(p3 >> 16); }
}

void split3_planes(const float* w, void* planes, long n, hipStream_t s) {
    hipLaunchKernelGGL(split3_planes_kernel, dim3((unsigned)((n / 2 + 255) / 256 + 1)), dim3(256), 0, s, w, (__bf16*)planes, n);
    MI_HIP(hipGetLastError());
}

// WIDE: wave tile 32 x 128 (one row block, four column blocks) instead of 64 x 64 — the split of an x fragment then feeds
// 24 MFMAs instead of 12 (half the VALU work per MFMA); plain epilogues only (the QKV epilogue wants 64 x 64 per wave).
// SHAPE 2: eight waves, 32 x 64 each — two waves per SIMD, so that one wave's DMA issue / waits / barriers leave the other
// one feeding the matrix core (each DMA instruction costs its wave ~100 cycles of issue; with one wave per SIMD the ten per
// chunk were a quarter of the kernel: 79.6 -> 58.6 us for FF1 with the DMA switched off)
template <typename TO, bool LEPI, int SHAPE, int NST>
__global__ __launch_bounds__(SHAPE >= 2 ? 512 : 256, 1) void linear_x3_kernel(const ConvGemmDev p) {
    // SHAPE 3: 256 x 128 tile, eight waves of 64 x 64, TWO stages (56 KB each): 28 KB of operands per 128x128x32 block of work
    // instead of 40 — this kernel is bound by the fabric-side fill
    constexpr bool WIDE = SHAPE == 1, W8 = SHAPE == 2, BIG = SHAPE == 3;
    constexpr int NW = (W8 || BIG) ? 8 : 4;
    using MF = Mfma<bf16>;
    using Frag = typename MF::Frag;
    constexpr int KC = 32;                                      // K chunk: 32 floats = 128 bytes of an x row
    constexpr int BM = BIG ? 256 : 128, BN = 128, WM = (WIDE || W8) ? 32 : 64, WN = WIDE ? 128 : 64, TM = (WIDE || W8) ? 1 : 2, TN = WIDE ? 4 : 2;
    constexpr int AHEAD = NST - 1;                              // chunks in flight beyond the one being computed (2 or 3)
    static_assert((BIG && NST == 2) || (!BIG && (NST == 3 || NST == 4)), "ring depth");
    constexpr int STEPS = 2 * TM, GRP = 2 * TN;                 // steps (k16 step, row block) per chunk ; groups of three MFMAs per step
    constexpr int A_BYTES = BM * KC * 4, BP_BYTES = BN * KC * 2, STAGE_BYTES = A_BYTES + 3 * BP_BYTES;     // 16 KB + 3 x 8 KB
    constexpr int DA = BM / 8 / NW, DB = 8 / NW;                    // DMA instructions per wave per chunk: DA x-row groups + 3 x DB weight-plane groups
    constexpr int PER = DA + 3 * DB;                            // 10 (four waves) or 5 (eight waves)
    constexpr int EPI_BYTES = NW * 16384 + 8 * 512;              // LDS-staged epilogues: 16 KB per wave (+ the 32 x 65 slack of the transposed-V path)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE_BYTES > EPI_BYTES || !BIG ? NST * STAGE_BYTES : EPI_BYTES];
    (void)smem;
#if defined(__HIP_DEVICE_COMPILE__)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = WIDE ? wave : wave >> 1, wn = WIDE ? 0 : wave & 1, lr = lane & 31, lk = lane >> 5;     // W8 / BIG: wm 0..3, wn 0..1
    // ranges: see gemm_sk.hip (XCD groups of whole tiles; range r of a group on workgroup (R-1-r)*8 + xg)
    const int P = (int)gridDim.x, R = P >> 3;
    const int xg = (int)blockIdx.x & 7;
    const int l = R - 1 - ((int)blockIdx.x >> 3);
    const int nch = p.K / KC;
    const int T_all = p.Tm * p.Tn;
    const int tile_lo_g = (int)((long)xg * T_all / 8), tile_hi = (int)((long)(xg + 1) * T_all / 8);
    // hybrid (p.tail_tiles != 0): `full` whole tiles per workgroup first, all starting at chunk 0 together, so that the 32
    // workgroups of an XCD walk K in step and share their panels in L2; only the group's remaining tiles are stream-K'd.
    // (This kernel is bound by the fabric-side fill — PMC: 313 MB per launch against 40 MB algorithmic with pure stream-K
    // ranges, which start at unrelated chunks — where the native fp32 kernel, MFMA-bound, lost 4 % to the same change.)
    const int full = p.tail_tiles ? (tile_hi - tile_lo_g) / R : 0;
    const int tile_lo = tile_lo_g + full * R;
    const long I = (long)(tile_hi - tile_lo) * nch;
    long it = (long)l * I / R;
    const long it1 = (long)(l + 1) * I / R;
    const int slot0 = xg * R;
    int dp_done = 0;

    const float* xb = (const float*)p.x;
    __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)((((long)p.T_in - 1) * p.x_rstride + p.Cin) * 4L), 0x00020000);
    __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, (int)((long)3 * p.N * p.K * 2L), 0x00020000);
    constexpr int OOB = 0x7fffff00;
    const unsigned smem_lds = (unsigned)(unsigned long)(const __attribute__((address_space(3))) void*)smem;
    __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.sk_ws, 0, (int)((long)P * BM * BN * 4), 0x00020000);
    int* flags = p.sk_flags;

    // x rows: lane -> row R0 + lane/8, 16-byte slot lane%8 holding k-vector slot ^ ((row >> 1) & 7)   (as gemm_sk.hip)
    const int kvl0 = (lane & 7) ^ ((lane >> 4) & 7), kvl1 = (lane & 7) ^ ((4 + (lane >> 4)) & 7);
    const int lrow = lane >> 3;
    // weight planes: lane -> row R0 + lane/4 (16 rows per instruction), slot lane%4 holding k-vector slot ^ ((row >> 2) & 3)
    const int kvb = (lane & 3) ^ ((lane >> 4) & 3);
    const int brow = lane >> 2;

    while (dp_done < full || it < it1) {
        const bool dp = p.tail_tiles == 2 ? !(it < it1) : dp_done < full;     // whole tile (mode 1: before the stream-K'd pieces, mode 2: after)
        const int tile_g = dp ? 0 : (int)(it / nch);
        const int cb = dp ? 0 : (int)(it - (long)tile_g * nch);
        const int tile = dp ? tile_lo_g + dp_done * R + l : tile_lo + tile_g;
        const int n = dp ? nch : (int)((it1 - it) < (long)(nch - cb) ? (it1 - it) : (long)(nch - cb));
        const int ce = cb + n;
        int nt, mt;
        if (p.RC == 0) { nt = tile / p.Tm; mt = tile - nt * p.Tm; }
        else { mt = tile / p.Tn; nt = tile - mt * p.Tn; }
        const int m0 = mt * BM, n0 = nt * BN;
        int avo[DA], bvo[DB];
#pragma unroll
        for (int j = 0; j < DA; ++j) {
            const int R0 = (wave * DA + j) * 8;
            avo[j] = (int)(((long)(m0 + R0 + lrow) * p.x_rstride + ((j & 1) ? kvl1 : kvl0) * 4) * 4L);
        }
#pragma unroll
        for (int h = 0; h < DB; ++h) {
            const long nn = n0 + (wave * DB + h) * 16 + brow;
            bvo[h] = nn < p.N ? (int)((nn * p.K + kvb * 8) * 2L) : OOB;
        }
        const int plane_bytes = (int)((long)p.N * p.K * 2L);
        // j < DA: x row groups ; DA <= j < PER: plane (j-DA)/DB, 16-row group (j-DA)%DB of this wave
        auto dma_one = [&](int st, int chunk, int j) __attribute__((always_inline)) {
            if (p.dbg & 1) return;
            const unsigned base = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)(st * STAGE_BYTES));
            const bool live = chunk < ce;
            if (j < DA) {
                const int cbytes = live ? chunk * KC * 4 : OOB;
                x3_bufds16(rsa, (int)((unsigned)avo[j] + (unsigned)cbytes), base + (unsigned)((wave * DA + j) * 8 * KC * 4));
            } else {
                const int pl = (j - DA) / DB, h = (j - DA) % DB;
                const int cbytes = live ? chunk * KC * 2 + pl * plane_bytes : OOB;
                x3_bufds16(rsb, (int)((unsigned)bvo[h] + (unsigned)cbytes), base + (unsigned)(A_BYTES + pl * BP_BYTES + (wave * DB + h) * 1024));
            }
        };
        auto issue = [&](int st, int chunk) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < PER; ++j) dma_one(st, chunk, j);
        };

        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // ---- fragments.  quarter q of a chunk = (k16 step q >> 1, row block q & 1): 12 MFMAs (2 column blocks x 6 terms) ----
        float4 araw[2];
        Frag a3[2][3], b3[2][TN][3];                            // a3[set][piece] ; b3[k16 parity][j][piece]
        const int sw_a = (lr >> 1) & 7, sw_b = (lr >> 2) & 3;
        auto ldA = [&](int st, int q) __attribute__((always_inline)) {
            if (p.dbg & 2) return;
            const int ks = q / TM, i = q % TM;
            const unsigned char* As = smem + st * STAGE_BYTES + (wm * WM + i * 32 + lr) * (KC * 4);
            const int kv = ks * 4 + lk * 2;
            araw[0] = *reinterpret_cast<const float4*>(As + ((kv ^ sw_a) << 4));
            araw[1] = *reinterpret_cast<const float4*>(As + (((kv + 1) ^ sw_a) << 4));
        };
        // pair pr (0..3) of the split: two of the eight values -> one 32-bit word of each of the three pieces (11 VALU
        // instructions: placed behind ONE MFMA, whose 32 matrix-core cycles cover most of it — a wave alone on its SIMD only
        // overlaps what sits in the shadow of an MFMA it has just issued)
        unsigned u1[4], u2[4], u3[4];
        auto splitA_pair = [&](int pr) __attribute__((always_inline)) {
            const float4 v = araw[pr >> 1];
            if (pr & 1) x3_split_pair(v.z, v.w, u1[pr], u2[pr], u3[pr]);
            else x3_split_pair(v.x, v.y, u1[pr], u2[pr], u3[pr]);
        };
        auto packA = [&](int set) __attribute__((always_inline)) {
            const x3_u4 w1 = {u1[0], u1[1], u1[2], u1[3]}, w2 = {u2[0], u2[1], u2[2], u2[3]}, w3 = {u3[0], u3[1], u3[2], u3[3]};
            a3[set][0] = __builtin_bit_cast(Frag, w1); a3[set][1] = __builtin_bit_cast(Frag, w2); a3[set][2] = __builtin_bit_cast(Frag, w3);
        };
        auto ldB = [&](int st, int ks) __attribute__((always_inline)) {
            if (p.dbg & 2) return;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const unsigned char* Bs = smem + st * STAGE_BYTES + A_BYTES + pl * BP_BYTES + (wn * WN + j * 32 + lr) * (KC * 2);
                    b3[ks & 1][j][pl] = *reinterpret_cast<const Frag*>(Bs + (((ks * 2 + lk) ^ sw_b) << 4));
                }
        };
        // MFMA k of step q: terms ordered small to large, the column blocks alternating
        auto mma1 = [&](int q, int set, int k) __attribute__((always_inline)) {
            const int ks = q / TM, i = q % TM;
            constexpr int TA[6] = {0, 1, 2, 0, 1, 0}, TB[6] = {2, 1, 0, 1, 0, 0};
            const int t = k / TN, j = k % TN;
            acc[i][j] = MF::mma(a3[set][TA[t]], b3[ks & 1][j][TB[t]], acc[i][j]);
        };
#define X3_SB() __builtin_amdgcn_sched_barrier(0)
        if (p.dbg & 2) {
            araw[0] = float4{0.f, 0.f, 0.f, 0.f}; araw[1] = araw[0];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int c = 0; c < 3; ++c) b3[a][b][c] = Frag{};
        }

        // ---- K loop: ring of three stages, two chunks ahead.  The 10 DMA instructions of chunk c+2 sit in the quarters of
        //      chunk c (3 + 3 + 2 before the boundary wait, 2 after it); the boundary (chunk c+1 landed, stage of chunk c
        //      released) comes before the LAST quarter, whose operands are already in registers.
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) issue(a, cb + a);
        // chunk cb has landed: the AHEAD-1 younger chunks (PER instructions each) may stay in flight
        if constexpr (AHEAD == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr ((AHEAD - 1) * PER == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if constexpr ((AHEAD - 1) * PER == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        static_assert(AHEAD == 1 || (AHEAD - 1) * PER == 10 || (AHEAD - 1) * PER == 20 || (AHEAD - 1) * PER == 5, "counted prologue wait");
        __builtin_amdgcn_s_barrier();
        ldA(0, 0); ldB(0, 0);
        splitA_pair(0); splitA_pair(1); splitA_pair(2); splitA_pair(3); packA(0);
        int st = 0, st_issue = AHEAD % NST;
        for (int c = 0; c < n; ++c) {
            int stn = st + 1; if (stn == NST) stn = 0;
            const int cn = cb + c + AHEAD;
#pragma unroll
            for (int q = 0; q < STEPS; ++q) {
                const int set = q & 1;
                if (q == STEPS - 1 && c + 1 < n) {
                    // in flight: the pieces of the newest chunk issued so far (PER - 2 with four waves, PER - 1 with eight) plus
                    // the AHEAD - 2 whole chunks before it
                    constexpr int INFL = AHEAD == 1 ? 0 : (AHEAD - 2) * PER + (W8 ? PER - 1 : PER - 2);      // two stages: everything has landed
                    if constexpr (INFL == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if constexpr (INFL == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else if constexpr (INFL == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
                    else if constexpr (INFL == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                    static_assert(INFL == 0 || INFL == 8 || INFL == 18 || INFL == 4 || INFL == 9, "counted boundary wait");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                const bool more = q < STEPS - 1 || c + 1 < n;     // another step follows: its operands are fetched and split under this one's MFMAs
                constexpr int NM = 6 * TN, SP = NM / 6;          // MFMAs per step ; spacing of the four split pairs
#pragma unroll
                for (int k = 0; k < NM; ++k) {
                    X3_SB(); mma1(q, set, k); X3_SB();
                    const int f = q * NM + k;                    // 0..47 inside the chunk
                    if (k == 0) {
                        if (q < STEPS - 1) {
                            ldA(st, q + 1);
                            if (q == 0) ldB(st, 1);              // weight fragments of the second k16 step
                        } else if (c + 1 < n) { ldA(stn, 0); ldB(stn, 0); }
                    }
                    if (more && k >= SP && k % SP == 0 && k / SP <= 4) {
                        splitA_pair(k / SP - 1);
                        if (k / SP == 4) packA(set ^ 1);
                    }
                    // the ten DMA instructions of chunk c+2: eight before the boundary step, two after it
                    if constexpr (SHAPE == 0) {
                        if (f == 1) dma_one(st_issue, cn, 0); else if (f == 5) dma_one(st_issue, cn, 1);
                        else if (f == 9) dma_one(st_issue, cn, 2); else if (f == 13) dma_one(st_issue, cn, 3);
                        else if (f == 17) dma_one(st_issue, cn, 4); else if (f == 21) dma_one(st_issue, cn, 5);
                        else if (f == 25) dma_one(st_issue, cn, 6); else if (f == 29) dma_one(st_issue, cn, 7);
                        else if (f == 37) dma_one(st_issue, cn, 8); else if (f == 41) dma_one(st_issue, cn, 9);
                    } else if constexpr (WIDE) {
                        if (f < 24 && (f % 3) == 1) dma_one(st_issue, cn, f / 3);             // f = 1, 4, ..., 22: eight
                        else if (f == 25) dma_one(st_issue, cn, 8); else if (f == 29) dma_one(st_issue, cn, 9);
                    } else if constexpr (BIG) {                                               // 48 MFMAs per chunk, all 7 pieces before the boundary
                        if (f < 28 && (f & 3) == 1) dma_one(st_issue, cn, f >> 2);             // f = 1, 5, ..., 25
                    } else {                                                                  // eight waves: 24 MFMAs per chunk, 4 + 1 pieces
                        if (f == 1) dma_one(st_issue, cn, 0); else if (f == 4) dma_one(st_issue, cn, 1);
                        else if (f == 7) dma_one(st_issue, cn, 2); else if (f == 10) dma_one(st_issue, cn, 3);
                        else if (f == 14) dma_one(st_issue, cn, 4);
                    }
                }
            }
            if (++st_issue == NST) st_issue = 0;
            st = stn;
        }
#undef X3_SB
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        // ---- partial tile: publish or collect (gemm_sk.hip) ----------------------------------------------------------------
        const int slot_lane = (wave * (TM * TN * 4)) * 64 + lane;
        if (p.dbg & 4) { if (dp) ++dp_done; else it += n; continue; }
        if (cb > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        x3_u4 v;
                        v.x = __float_as_uint(acc[i][j][4 * q]); v.y = __float_as_uint(acc[i][j][4 * q + 1]);
                        v.z = __float_as_uint(acc[i][j][4 * q + 2]); v.w = __float_as_uint(acc[i][j][4 * q + 3]);
                        const int unit = slot_lane + ((i * TN + j) * 4 + q) * 64;
                        __builtin_amdgcn_raw_buffer_store_b128(v, rsw, ((slot0 + l) * (BM * BN / 4) + unit) * 16, 0, 16);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flags + slot0 + l, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (ce < nch) {
                int cov = ce;
                for (int q_l = l + 1; cov < nch; ++q_l) {
                    const long q0 = (long)q_l * I / R, q1 = (long)(q_l + 1) * I / R;
                    if (q1 == q0) continue;
                    if (tid == 0) {
                        while (__hip_atomic_load(flags + slot0 + q_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(2);
                    }
                    __syncthreads();
                    x3_u4 v[TM * TN * 4];
#pragma unroll
                    for (int u = 0; u < TM * TN * 4; ++u)
                        v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsw, ((slot0 + q_l) * (BM * BN / 4) + slot_lane + u * 64) * 16, 0, 16);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const x3_u4 w = v[(i * TN + j) * 4 + q];
                                acc[i][j][4 * q] += __uint_as_float(w.x); acc[i][j][4 * q + 1] += __uint_as_float(w.y);
                                acc[i][j][4 * q + 2] += __uint_as_float(w.z); acc[i][j][4 * q + 3] += __uint_as_float(w.w);
                            }
                    __syncthreads();
                    if (tid == 0) __hip_atomic_store(flags + slot0 + q_l, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    cov += (int)((q1 - q0) < (long)(nch - cov) ? (q1 - q0) : (long)(nch - cov));
                }
            }
            if constexpr (LEPI) {
                float* stage = reinterpret_cast<float*>(smem) + wave * (W8 ? 2176 : 4096);      // 16 KB per wave (2 x 32 x 64 or 1 x 32 x 128 floats); eight waves: 32 x 64, or 32 x 65 for the transposed-V path of the QKV epilogue
                bool done = false;
                if constexpr (SHAPE == 0 || SHAPE == 3) {
                    if (p.epi == EPI_QKV_ROPE) { gemm_epilogue_qkv_lds<TO>(acc, p, m0, n0, 0, wm, wn, lr, lk, stage); done = true; }
                } else if constexpr (SHAPE == 2) {
                    if (p.epi == EPI_QKV_ROPE) { gemm_epilogue_qkv_lds<TO, 1>(acc, p, m0, n0, 0, wm, wn, lr, lk, stage); done = true; }
                }
                if (!done) gemm_epilogue_lds<TO, TM, TN, WM, WN>(acc, p, m0, n0, 0, 0, wm, wn, lr, lk, stage);
            } else {
                gemm_epilogue<TO, TM, TN, WM, WN>(acc, p, m0, n0, 0, 0, wm, wn, lr, lk);
            }
            __syncthreads();
        }
        if (dp) ++dp_done; else it += n;
    }
#endif
}

// Round 3 pruned the A/B losers of this kernel: the launcher instantiates ONE layout — eight waves of 32x64, four-stage ring,
// pure stream-K ranges — the round-2 default.  Measured and dropped: 64x64 x four waves 80.6 us, 32x128 x
// four waves 70.2 us (eight waves: 64.0 us in the model); three stages +2 %; whole-tiles-first hybrids 67.4 / 66.0 vs 63.3 us;
// 256x128 tiles with two stages QKV 113 vs 99 us.  (The template keeps those shapes' code paths; nothing instantiates them.)
// This kernel is the fallback of gemm_x3p.hip (N not a multiple of 128, or the caller has no panel planes).
void launch_linear_x3(const ConvGemmDev& e_in, hipStream_t s) {
    ConvGemmDev e = e_in;
    e.tail_tiles = 0;
    int dev = 0, cus = 256;
    MI_HIP(hipGetDevice(&dev));
    {
        static int cu_count[16] = {0};
        if (!cu_count[dev & 15]) { hipDeviceProp_t pr; MI_HIP(hipGetDeviceProperties(&pr, dev)); cu_count[dev & 15] = pr.multiProcessorCount; }
        cus = cu_count[dev & 15];
    }
    const int P = std::min(cus, e.sk_slots) & ~7;
    const dim3 grid(P);
    if (e.lds_epi) {
        prof_set_kernel("linear_x3_kernel<float, true, 8 waves, 4>", "", "");
        hipLaunchKernelGGL((linear_x3_kernel<float, true, 2, 4>), grid, dim3(512), 0, s, e);
    } else {
        prof_set_kernel("linear_x3_kernel<float, false, 8 waves, 4>", "", "");
        hipLaunchKernelGGL((linear_x3_kernel<float, false, 2, 4>), grid, dim3(512), 0, s, e);
    }
    MI_HIP(hipGetLastError());
}

}  // namespace mi
