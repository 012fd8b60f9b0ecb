// gemm_x3p.hip — the fp32 DiT linear layers (QKV, O, FF1, FF2) from 16-bit partial products (NP = 2: fp16 {hi, lo} pairs,
// three MFMAs per block, the default; NP = 3: exact three-way bf16 splits, six) with BOTH operands pre-split and pre-tiled
// ("panel planes"), so that the main loop is nothing but LDS-DMA, ds_read_b128 and MFMA.
//
// Arithmetic (x3_split.h).  NP = 2: a = hi + lo with hi = fp16(a), lo = fp16((a - hi) * 2^11) — 22 significant bits; per
// block hi*hi into one accumulator set and hi*lo + lo*hi into a second, joined as acc0 + 2^-11 acc1 at the end of the tile;
// 2500 / 3 = 833 TFLOP/s ceiling; operands beyond fp16's range are counted (range watch) and the engine falls back to
// NP = 3: that of gemm_x3.hip (a = a1 + a2 + a3, b likewise in bf16, six partial products per block, fp32 accumulate,
// small terms first), 2500 / 6 = 416.7 TFLOP/s ceiling.
//
// What round 2's kernel (gemm_x3.hip) paid for, measured (profiles/r2): it staged the activations as
// fp32 and split them in registers in front of the MFMAs (11 VALU per pair, repeated by each of the 8-24 column tiles that
// read the same rows), fetched the weight planes as 64-byte row pieces (the per-CU L2->LDS fill rate is bound per cache
// line touched: 64-byte pieces run at half the rate of whole lines), and its stream-K ranges walked K out of step, so an
// XCD's workgroups did not share panels in L2 (TCC hit 38 %, 214 MB of fabric traffic per launch against ~38 MB).
//
// Here:
//  * Panel-plane layout, for the weights (once at load) AND for the activations (written by the producer of the rows:
//    x3p_split_rows, or the fused epilogues of rownorm / attention / FF1):
//        [panel = row / 128][chunk = k / 32][plane 0..NP-1][row % 128][32 x 16 bit]   = 8 NP KB per (panel, chunk)
//    with the 16-byte k-slots of a 64-byte row XOR-swizzled by (row >> 2) & 3 IN MEMORY.  One K chunk of a tile operand
//    is 8 NP KB contiguous: every LDS-DMA instruction moves 1 KB of consecutive bytes (the best fill pattern, fillrate.hip),
//    needs no per-lane address arithmetic, and lands in LDS already in the conflict-free fragment layout.
//  * 128x128 tile, eight waves = two groups of four: group g multiplies k16 step g of every chunk on 64x64 per wave
//    (24 MFMAs per wave per chunk from 12 ds_read_b128: 62 B/clk of LDS reads at full MFMA rate, against 85-94 for the
//    32x64 wave tiles of round 2); the two groups' accumulators meet once per tile in LDS.  Two waves per SIMD.
//  * LDS-DMA ring.  NP = 3: three stages of 48 KB, one barrier per chunk in front of the last six MFMAs.  NP = 2 (fp16
//    {hi, lo * 2^11} pairs, the default — three MFMAs per block on two accumulator sets): FIVE stages of 32 KB (all 160 KB)
//    and ONE barrier per TWO chunks (X3P_BAR2): the DMA of chunk c + 4 is issued first in the body into the stage chunk
//    c - 1 left, the fragments of chunk c + 1 are read into a second register set under the MFMAs of chunk c, every wait
//    is a counted vmcnt.
//  * One epilogue per instantiation (EPK, round 4): the kernel sits at the 256-VGPR limit and each epilogue compiled into
//    it (QKV + RoPE + V^T | plane output with GELU | gated residual + AdaLN statistics) cost spills that slowed every
//    layer; the launcher picks the instantiation by the layer's role.
//  * Stream-K frame of gemm_sk.hip (range-ordered fix-up, flags) with two changes for L2: the 8 XCD groups own 2-D blocks
//    of tiles (GR x GC bands: A is fetched by GC XCDs and B by GR, instead of A by all 8), and every tile walks K
//    CYCLICALLY from a per-tile start chunk chosen so that all workgroups of the launch are at the same physical chunk at
//    the same time (ranges still differ by (range length mod chunks-per-tile) for their tail pieces): the workgroups of an
//    XCD that share a row or weight panel request the same 24 KB within a few chunk times of each other.
#include <atomic>
#include "common.h"
#include "mfma.h"
#include "gemm_epilogue.h"
#include "x3_split.h"
#include <type_traits>
#include <cstdlib>

namespace mi {

long x3p_bytes(long rows, long K, int np) { return ((rows + 127) / 128) * (K / 32) * (long)x3p_chunk_bytes(np); }

// fp32 rows [rows][ld] (K columns used) -> panel planes.  One thread = one 16-byte k-slot (8 values) of one row; rows in
// [rows, rows_pad) are written as zeros (the last row panel is always whole).
template <int NP>
__global__ __launch_bounds__(256) void x3p_split_rows_kernel(const float* __restrict__ x, long ld, unsigned char* __restrict__ out,
                                                             int rows, int K, long total, int* __restrict__ satp) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int s8n = K >> 3;
    const int row = (int)(idx / s8n), s8 = (int)(idx - (long)row * s8n);
    float4 v0 = float4{0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (row < rows) {
        const float4* src = reinterpret_cast<const float4*>(x + (long)row * ld + s8 * 8);
        v0 = src[0]; v1 = src[1];
    }
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    x3_u4 pl[NP];
    unsigned sat = 0;
    xnp_split8_sat<NP>(v, pl, sat);
    unsigned char* dst = out + x3p_slot_offset(row, s8, K >> 5, NP);
#pragma unroll
    for (int q = 0; q < NP; ++q) *reinterpret_cast<x3_u4*>(dst + q * X3P_PLANE) = pl[q];
    if constexpr (NP == 2) sat_publish(satp, sat);
}

void x3p_split_rows(const float* x, long ld, void* planes, int rows, int K, hipStream_t s, int np, int* sat) {
    MI_REQUIRE(K % 32 == 0 && ld % 4 == 0 && ((uintptr_t)x % 16) == 0, "x3p_split_rows: K must be whole 32-deep chunks, rows 16-byte aligned");
    MI_REQUIRE(np == 2 || np == 3, "x3p_split_rows: 2 or 3 planes");
    const int rows_pad = (rows + 127) / 128 * 128;
    const long total = (long)rows_pad * (K / 8);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (np == 3) {
        prof_set_kernel("x3p_split_rows_kernel<3>", "", "");
        hipLaunchKernelGGL(x3p_split_rows_kernel<3>, grid, dim3(256), 0, s, x, ld, (unsigned char*)planes, rows, K, total, sat);
    } else {
        prof_set_kernel("x3p_split_rows_kernel<2>", "", "");
        hipLaunchKernelGGL(x3p_split_rows_kernel<2>, grid, dim3(256), 0, s, x, ld, (unsigned char*)planes, rows, K, total, sat);
    }
    MI_HIP(hipGetLastError());
}

template <typename RSRC>
__device__ __forceinline__ void x3p_dma16(RSRC rsrc, int voff, unsigned lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
#endif
}

// the same with an instruction offset OFF (0 / 1024 / 2048 / 3072): it advances the global address AND the LDS address, and the
// operand chunk is laid out in LDS exactly as in memory, so four consecutive 1 KB pieces share one M0 value
template <int OFF, typename RSRC>
__device__ __forceinline__ void x3p_dma16_off(RSRC rsrc, int voff) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (OFF == 0) asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(rsrc) : "memory");
    else if constexpr (OFF == 1024) asm volatile("buffer_load_dwordx4 %0, %1, 0 offen offset:1024 lds" :: "v"(voff), "s"(rsrc) : "memory");
    else if constexpr (OFF == 2048) asm volatile("buffer_load_dwordx4 %0, %1, 0 offen offset:2048 lds" :: "v"(voff), "s"(rsrc) : "memory");
    else asm volatile("buffer_load_dwordx4 %0, %1, 0 offen offset:3072 lds" :: "v"(voff), "s"(rsrc) : "memory");
#endif
}
template <int N> __device__ __forceinline__ void x3p_wait_vm() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
#endif
}
__device__ __forceinline__ void x3p_set_m0(unsigned v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(v) : "memory", "m0");
#endif
}

// p.x  = A panel planes (x3p layout, Tm panels x nch chunks) ; p.w3 = B panel planes (Tn panels x nch chunks)
// p.Tm, p.Tn tiles ; p.RT = GR, p.RC = GC (XCD bands) ; p.tail_tiles bit0 = no cyclic K alignment (A/B switch)
// DBG (tuning builds of the same kernel, MI355TTS_GEMM_DBG): bit 0 no LDS-DMA, bit 1 no fragment reads, bit 3 no MFMA; p.dbg bit 2
// (run time): no fix-up / epilogue
// NP = 3: bf16 planes, six products on one accumulator set.  NP = 2: fp16 {hi, lo * 2^11} planes, three products per block
// on TWO accumulator sets — accA += hi*hi, accB += lo*hi + hi*lo (both carry the 2^11 of one low part) — combined once per
// tile as accA + 2^-11 * accB; the lo*lo term (2^-22 of the product, below the fp32 rounding of the sum) is dropped.
// FOLD: the instantiation whose epilogues are the AdaLN-fold forms (gemm_epilogue.h: consumer QKV / FF1, producer O / FF2) instead
// of the plain ones — a separate kernel, so that neither carries the other's code and registers
// EPK (round 4): which ONE epilogue the instantiation carries — 0: all of the FOLD / plain set (the round-3 form), 1: the QKV epilogue,
// 2: the plain-output one (FF1 planes / rows), 3: the residual one (O / FF2).  The kernel sits at the 256-register limit: every
// epilogue compiled into it costs the main loop spills (adding one store path to the QKV epilogue took the spills from 31 to 81
// registers and 4 - 11 us from EVERY layer), so each launch gets the instantiation with only its own.
template <typename TO, bool LEPI, int NP, int DBG = 0, bool FOLD = false, int EPK = 0>
__global__ __launch_bounds__(512, 1) void linear_x3p_kernel(const ConvGemmDev p) {
    using MF = std::conditional_t<NP == 3, Mfma<bf16>, Mfma<f16>>;
    using Frag = typename MF::Frag;
    constexpr int BM = 128, BN = 128, WM = 64, WN = 64, TM = 2, TN = 2;
#ifndef X3P_NST2
#define X3P_NST2 4
#endif
    // LDS stages: a chunk is requested NST - 1 chunk times before its barrier.  Three for NP = 3 (a chunk time is ~3000 clk);
    // with half the MFMAs per chunk (NP = 2) the same distance in TIME needs one stage more (32 KB stages: 128 KB)
#ifndef X3P_BAR2
#define X3P_BAR2 1          // -DX3P_BAR2=0: the round-3 loop (four stages, a barrier per chunk) for A/B builds
#endif
    // BAR2 (NP = 2, round 4): five stages = the whole 160 KB of LDS, ONE barrier per TWO chunks.  With a barrier per 12 MFMAs both
    // waves of every SIMD park together (PMC: 42 % of the wave cycles in s_waitcnt / s_barrier); here the boundary after an odd
    // chunk makes chunks c+2 AND c+3 visible (only the youngest request, chunk c+4, stays in flight), the even chunk runs into the
    // odd one without a wait, and each body requests chunk c+4 FIRST (under its first MFMAs: two chunk times of lead, like the
    // three-stage ring that measured the same as four) into the stage of chunk c-1 — read two bodies ago, i.e. always behind a
    // barrier.  Same MFMAs in the same order on the same fragments: bit-identical results (checked: int16 waveforms array_equal).
    // Same-box A/B (tools/dbg/ab_bar2.sh, profiles/r4/x3p_bar2_ab.txt): 51.5 -> 50.35 us per launch, step 182.8 -> 178.8 ms.
    constexpr bool BAR2 = (X3P_BAR2 != 0) && NP == 2;             // (round 5: the tuning instantiations run the product loop too)
    constexpr int NST = NP == 3 ? 3 : (BAR2 ? 5 : X3P_NST2);
    constexpr int CHB = NP * X3P_PLANE;                         // bytes of one operand chunk
    constexpr int STAGE = 2 * CHB;                              // 48 | 32 KB: A chunk then B chunk
    constexpr int PER = 2 * NP;                                 // DMA instructions per wave per chunk (1 KB each, 8 waves)
    constexpr int NT = NP == 3 ? 6 : 3;                         // partial products per block
    constexpr int NM = 4 * NT, NR = 4 * NP;                     // MFMAs and fragment reads per wave per chunk
    constexpr int KB = NR + PER;                                // the boundary sits in front of MFMA KB (after the loop if KB == NM)
    static_assert(KB <= NM, "x3p: reads and DMA pieces must fit under the chunk's MFMAs");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE];
    (void)smem;
#if defined(__HIP_DEVICE_COMPILE__)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave >> 2, w4 = wave & 3;                    // k16 half of every chunk ; 64x64 sub-tile
    const int wm = w4 >> 1, wn = w4 & 1, lr = lane & 31, lk = lane >> 5;
    const int P = (int)gridDim.x, R = P >> 3;
    const int xg = (int)blockIdx.x & 7;
    const int l = R - 1 - ((int)blockIdx.x >> 3);               // range index inside the XCD group (gemm_sk.hip: pieces are awaited from lower workgroup ids)
    const int nch = p.K >> 5;
    // XCD group -> band of row tiles x band of column tiles
    const int GR = p.RT, GC = p.RC;
    const int gr = xg / GC, gc = xg - gr * GC;
    const int r0 = (int)((long)gr * p.Tm / GR), r1 = (int)((long)(gr + 1) * p.Tm / GR);
    const int c0 = (int)((long)gc * p.Tn / GC), c1 = (int)((long)(gc + 1) * p.Tn / GC);
    const int bh = r1 - r0;
    const long I = (long)bh * (c1 - c0) * nch;
    long it = (long)l * I / R;
    const long it1 = (long)(l + 1) * I / R;
    const int slot0 = xg * R;

    __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.sk_ws, 0, (int)((long)P * BM * BN * 4), 0x00020000);
    const bool opA = wave < 4;                                  // waves 0-3 stage the A chunk, waves 4-7 the B chunk
    constexpr int OOB = 0x7fffff00;
    const unsigned smem_lds = (unsigned)(unsigned long)(const __attribute__((address_space(3))) void*)smem;
    int* flags = p.sk_flags;
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)(opA ? p.x : p.w3), 0, (int)((long)(opA ? p.Tm : p.Tn) * nch * CHB), 0x00020000);
    const int lane16 = lane * 16;
    const unsigned lds_part = (unsigned)((opA ? 0 : CHB) + w4 * PER * 1024);
    // fragment addresses inside a stage: row (wm*64 + i*32 + lr) of plane pl, k-slot 2*kg + lk, swizzled
    const int sw = (lr >> 2) & 3;
    const unsigned fa_off = (unsigned)((wm * WM + lr) * 64 + (((2 * kg + lk) ^ sw) << 4));
    const unsigned fb_off = (unsigned)(CHB + (wn * WN + lr) * 64 + (((2 * kg + lk) ^ sw) << 4));

    while (it < it1) {
        const int tile_g = (int)(it / nch);
        const int cb = (int)(it - (long)tile_g * nch);
        const int n = (int)((it1 - it) < (long)(nch - cb) ? (it1 - it) : (long)(nch - cb));
        const int ce = cb + n;
        const int nt = c0 + tile_g / bh, mt = r0 + tile_g - (tile_g / bh) * bh;     // row tiles fastest inside the band
        const int m0 = mt * BM, n0 = nt * BN;
        // cyclic K walk: local chunk u of this tile is physical chunk (u + shift) mod nch, shift = distance of the tile's
        // first iteration from the start of the range that owns it (so that owner is at physical chunk (its own elapsed
        // chunks) mod nch, like every other workgroup)
        int shift = 0;
        if (!(p.tail_tiles & 1)) {
            const long s_t = (long)tile_g * nch;
            const long lh = ((s_t + 1) * R - 1) / I;
            shift = (int)((s_t - lh * I / R) % nch);
        }
        // per-lane byte offsets of this wave's six 1 KB pieces of a chunk (A chunk for waves 0-3, B chunk for waves 4-7)
        int vb[PER];
        {
            const int d_base = (int)((long)(opA ? mt : nt) * nch * CHB) + lane16 + w4 * PER * 1024;
#pragma unroll
            for (int j = 0; j < PER; ++j) vb[j] = d_base + j * 1024;
        }
        // byte offset of local chunk `local` inside its panel (wave-uniform); past the piece: out of range -> the range check
        // writes zeros, nothing is fetched (keeps the vmcnt arithmetic uniform)
        auto chunk_off = [&](int local) __attribute__((always_inline)) -> int {
            int ph = local + shift; if (ph >= nch) ph -= nch;
            return __builtin_amdgcn_readfirstlane(local < ce ? ph * CHB : OOB);
        };
        auto issue = [&](int st, int local) __attribute__((always_inline)) {
            const int coff = chunk_off(local);
            const unsigned base = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)(st * STAGE) + lds_part);
            if constexpr (DBG & 1) return;
#pragma unroll
            for (int j = 0; j < PER; ++j) x3p_dma16(rsd, (int)((unsigned)vb[j] + (unsigned)coff), base + (unsigned)(j * 1024));
        };

        // AdaLN fold, consumer side: the LayerNorm statistics of the 32 rows this wave will own in the epilogue (block wm * 2 + kg of the
        // tile), requested now — their L2 round trip and the 16-partial sum run under the main loop instead of in front of the epilogue
        float ln_rs[1] = {0.f}, ln_mr[1] = {0.f};
        if constexpr (FOLD) {
            if (cb == 0 && p.ln_stats_in) ln_rows32(p, (long)p.m_off + m0 + (wm * 2 + kg) * 32, (long)p.m_off + p.M - 1, lr, lk, ln_rs[0], ln_mr[0]);
        }
        constexpr int NACC = NP == 3 ? 1 : 2;
        f32x16 accs[NACC][TM][TN];
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accs[a][i][j][r] = 0.f;
        Frag fa[2][TM][NP], fb[2][TN][NP];                      // [register set][block][plane]
        if constexpr (DBG & 2) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int c = 0; c < NP; ++c) { fa[a][b][c] = Frag{}; fb[a][b][c] = Frag{}; }
        }
        // one of the NR fragment reads of a chunk: q < 2 * NP: A (block q / NP, plane q % NP) ; else B
        auto ldfrag1 = [&](const unsigned char* sa, const unsigned char* sb, auto SET, int q) __attribute__((always_inline)) {
            constexpr int set = decltype(SET)::value;
            if constexpr (DBG & 2) return;
            if (q < 2 * NP) fa[set][q / NP][q % NP] = *reinterpret_cast<const Frag*>(sa + (q / NP) * (32 * 64) + (q % NP) * X3P_PLANE);
            else { const int qq = q - 2 * NP; fb[set][qq / NP][qq % NP] = *reinterpret_cast<const Frag*>(sb + (qq / NP) * (32 * 64) + (qq % NP) * X3P_PLANE); }
        };
        // MFMA k of a chunk (0 .. NM-1): term t = k / 4 (small to large), block k % 4
        auto mma1 = [&](auto SET, int k) __attribute__((always_inline)) {
            constexpr int set = decltype(SET)::value;
            constexpr int TA3[6] = {0, 1, 2, 0, 1, 0}, TB3[6] = {2, 1, 0, 1, 0, 0};
            constexpr int TA2[3] = {1, 0, 0}, TB2[3] = {0, 1, 0}, AC2[3] = {1, 1, 0};   // lo*hi, hi*lo -> accB ; hi*hi -> accA
            const int t = k >> 2, i = (k >> 1) & 1, j = k & 1;
            if constexpr (DBG & 8) return;
            if constexpr (NP == 3) accs[0][i][j] = MF::mma(fa[set][i][TA3[t]], fb[set][j][TB3[t]], accs[0][i][j]);
            else accs[AC2[t]][i][j] = MF::mma(fa[set][i][TA2[t]], fb[set][j][TB2[t]], accs[AC2[t]][i][j]);
        };
#define X3P_SB() __builtin_amdgcn_sched_barrier(0)
        // ---- prologue: chunks cb, cb+1, cb+2 requested; chunk cb's fragments into set 0 ----
        constexpr int NPRE = BAR2 ? NST - 1 : NST;             // chunks requested by the prologue
#pragma unroll
        for (int q = 0; q < NPRE; ++q) issue(q, cb + q);
        x3p_wait_vm<(NPRE - 1) * PER>();
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int q = 0; q < NR; ++q) ldfrag1(smem + fa_off, smem + fb_off, std::integral_constant<int, 0>{}, q);
        if constexpr (DBG & 16) {       // tuning: REAL operand values in both register sets, then a loop of MFMAs only
#pragma unroll
            for (int q = 0; q < NR; ++q) ldfrag1(smem + fa_off, smem + fb_off, std::integral_constant<int, 1>{}, q);
        }
        x3p_wait_vm<(BAR2 ? 1 : NST - 2) * PER>();              // BAR2: chunks cb+1 AND cb+2 landed (cb+3 may be in flight)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // chunk cb+1 landed, stage 0 free
        int st_next = 1, st_free = BAR2 ? 4 : 0;                // stage of chunk c+1 ; stage the chunk requested in body(c) goes to (chunk c+NST -> stage of chunk c ; BAR2: chunk c+4 -> stage of chunk c-1)
        // One chunk: 24 MFMAs on the fragments of chunk c (register set SET).  Under twelve of them the fragments of chunk c+1
        // are read into the other set (past the piece they are stale LDS, never used), under six this wave's six pieces of
        // chunk c+3 are requested; then the boundary, then the last six MFMAs.  Measured (profiles/r3/x3p_ablation_*.txt,
        // QKV shape): MFMAs alone on zero operands 45 us, data movement alone 36 us, together 75 us.  WHERE the six DMA
        // pieces and twelve reads sit among the MFMAs makes no difference (DMA first / last / every third slot, M0 written
        // per piece or twice per chunk: 74.8 - 76.1 us), and giving the two waves of a SIMD opposite slots is slower (83 us):
        // the parts do not hide each other because the package is at its power cap — see DESIGN.md section 4.
        auto body = [&](int c, auto SET) __attribute__((always_inline)) {
            constexpr int set = decltype(SET)::value;
            using NSET = std::integral_constant<int, set ^ 1>;
            const bool more = c + 1 < n;
            const unsigned char* sa = smem + st_next * STAGE + fa_off;
            const unsigned char* sb = smem + st_next * STAGE + fb_off;
            const int coff = chunk_off(cb + c + (BAR2 ? NST - 1 : NST));
            const unsigned ldsd = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)(st_free * STAGE) + lds_part);
            auto boundary = [&]() __attribute__((always_inline)) {
                if constexpr (DBG & 32) return;                 // tuning: no waits, no barriers in the loop (wrong numbers, right MFMA stream)
                // chunk c+2 has landed (this wave's PER pieces of chunk c+3 may stay in flight), the fragments of chunk c+1 are
                // in registers (its stage is free); for NP = 3 the last six MFMAs run behind the barrier
                x3p_wait_vm<(BAR2 ? 1 : NST - 2) * PER>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            };
#pragma unroll
            for (int k = 0; k < NM; ++k) {
                if (k == KB && more) boundary();
                X3P_SB(); mma1(SET, k); X3P_SB();
                if constexpr (DBG & 16) continue;
                // BAR2: the DMA pieces first (slots 0 .. PER-1), the fragment reads behind them
                const int kr = BAR2 ? k - PER : k, kd = BAR2 ? k : k - NR;
                if (kr >= 0 && kr < NR) ldfrag1(sa, sb, NSET{}, kr);
                else if (kd >= 0 && kd < PER) {
                    if constexpr (!(DBG & 1)) {
                        // pieces 0-3 under one M0 value, pieces 4-5 under the next (instruction offsets 0 / 1 / 2 / 3 KB move the
                        // global and the LDS address together)
                        const int dj = kd;
                        if (dj == 0) x3p_set_m0(ldsd); else if (dj == 4) x3p_set_m0(ldsd + 4096u);
                        const int v0 = (int)((unsigned)vb[dj & 4] + (unsigned)coff);
                        if ((dj & 3) == 0) x3p_dma16_off<0>(rsd, v0); else if ((dj & 3) == 1) x3p_dma16_off<1024>(rsd, v0);
                        else if ((dj & 3) == 2) x3p_dma16_off<2048>(rsd, v0); else x3p_dma16_off<3072>(rsd, v0);
                    }
                }
            }
            if constexpr (KB == NM) { if (more && (!BAR2 || (c & 1))) boundary(); }       // BAR2: only behind the odd chunk
        };
        auto rotate = [&]() __attribute__((always_inline)) {
            if constexpr (BAR2) { st_free = st_free + 1 == NST ? 0 : st_free + 1; }
            else st_free = st_next;
            st_next = st_next + 1 == NST ? 0 : st_next + 1;
        };
        for (int c = 0; c < n; c += 2) {
            body(c, std::integral_constant<int, 0>{});
            rotate();
            if (c + 1 < n) {
                body(c + 1, std::integral_constant<int, 1>{});
                rotate();
            }
        }
#undef X3P_SB
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        f32x16 (&acc)[TM][TN] = accs[0];
        if constexpr (NP == 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = __builtin_fmaf(accs[1][i][j][r], 0x1p-11f, acc[i][j][r]);
        }

        // ---- the two k16 groups meet.  Wave (group g, sub-tile w4) holds a 64x64 partial sum; the two waves of a sub-tile SWAP
        //      halves through LDS: group 0 keeps rows 0-31 and receives group 1's, group 1 keeps rows 32-63 and receives group
        //      0's — every wave ends up with a finished 32x64 block, so all eight waves take part in the fix-up and the epilogue
        //      (with waves 4-7 merely handing their sums to waves 0-3 the epilogue phase ran on half the workgroup) ----
        f32x16 h[1][TN];
        {
            x3_u4* red = reinterpret_cast<x3_u4*>(smem);             // [wave][j][q][lane] 16-byte units: 8 KB per wave
            auto park = [&](const f32x16 (&src)[TN]) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        x3_u4 v;
                        v.x = __float_as_uint(src[j][4 * q]); v.y = __float_as_uint(src[j][4 * q + 1]);
                        v.z = __float_as_uint(src[j][4 * q + 2]); v.w = __float_as_uint(src[j][4 * q + 3]);
                        red[(wave * 8 + j * 4 + q) * 64 + lane] = v;
                    }
            };
            auto take = [&](const f32x16 (&mine)[TN]) __attribute__((always_inline)) {
                const int other = wave ^ 4;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const x3_u4 v = red[(other * 8 + j * 4 + q) * 64 + lane];
                        h[0][j][4 * q] = mine[j][4 * q] + __uint_as_float(v.x); h[0][j][4 * q + 1] = mine[j][4 * q + 1] + __uint_as_float(v.y);
                        h[0][j][4 * q + 2] = mine[j][4 * q + 2] + __uint_as_float(v.z); h[0][j][4 * q + 3] = mine[j][4 * q + 3] + __uint_as_float(v.w);
                    }
            };
            if (kg == 0) park(acc[1]); else park(acc[0]);
            __syncthreads();
            if (kg == 0) take(acc[0]); else take(acc[1]);
            __syncthreads();
        }
        const int wm2 = wm * 2 + kg;                            // 32-row block of the tile this wave now owns

        // ---- partial tile: publish or collect (gemm_sk.hip), eight waves x 8 KB ------------------------------------------------
        const int slot_lane = (wave * (TN * 4)) * 64 + lane;
        if (p.dbg & 4) { it += n; continue; }                  // tuning: no fix-up, no epilogue
        if (cb > 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    x3_u4 v;
                    v.x = __float_as_uint(h[0][j][4 * q]); v.y = __float_as_uint(h[0][j][4 * q + 1]);
                    v.z = __float_as_uint(h[0][j][4 * q + 2]); v.w = __float_as_uint(h[0][j][4 * q + 3]);
                    const int unit = slot_lane + (j * 4 + q) * 64;
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
                        // Bounded spin.  The producer of this flag is a LOWER workgroup id; the hand-off assumes in-order workgroup
                        // dispatch (lower ids are resident or finished when this one runs), as gemm_sk.hip documents.  The bound
                        // only turns a broken invariant into an error word (checked by the engine at its next synchronisation:
                        // SkWorkspace::tripped) and wrong numbers instead of a hung GPU.
                        long spins = 0;
                        while (__hip_atomic_load(flags + slot0 + q_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                            __builtin_amdgcn_s_sleep(2);
                            if (++spins > (1L << 26)) { __hip_atomic_store(flags + p.sk_slots - 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                        }
                    }
                    __syncthreads();
                    {
                        x3_u4 v[TN * 4];
#pragma unroll
                        for (int u = 0; u < TN * 4; ++u)
                            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsw, ((slot0 + q_l) * (BM * BN / 4) + slot_lane + u * 64) * 16, 0, 16);
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const x3_u4 w = v[j * 4 + q];
                                h[0][j][4 * q] += __uint_as_float(w.x); h[0][j][4 * q + 1] += __uint_as_float(w.y);
                                h[0][j][4 * q + 2] += __uint_as_float(w.z); h[0][j][4 * q + 3] += __uint_as_float(w.w);
                            }
                    }
                    __syncthreads();
                    if (tid == 0) __hip_atomic_store(flags + slot0 + q_l, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    cov += (int)((q1 - q0) < (long)(nch - cov) ? (q1 - q0) : (long)(nch - cov));
                }
            }
            if constexpr (LEPI) {
                float* stage = reinterpret_cast<float*>(smem) + wave * 2176;        // 32 x 64 floats per wave (32 x 65 for the transposed-V path)
                if constexpr (FOLD) {
                    if ((EPK == 0 || EPK == 1) && (EPK == 1 || p.epi == EPI_QKV_ROPE)) gemm_epilogue_qkv_lds<TO, 1, true, true>(h, p, m0, n0, 0, wm2, wn, lr, lk, stage, ln_rs, ln_mr);       // consumer: QKV
                    else if ((EPK == 0 || EPK == 2) && (EPK == 2 || p.ln_stats_in)) gemm_epilogue_ln_in<TO, 1, TN, NP, true>(h, p, m0 + wm2 * 32, n0 + wn * WN, lr, lk, stage, ln_rs, ln_mr); // consumer: FF1 -> panel planes
                    else if constexpr (EPK == 0 || EPK == 3) gemm_epilogue_resid_ln<float, 1, TN, NP>(h, p, m0 + wm2 * 32, n0 + wn * WN, lr, lk, stage);                      // producer: O / FF2
                } else {
                    if ((EPK == 0 || EPK == 1) && (EPK == 1 || p.epi == EPI_QKV_ROPE)) gemm_epilogue_qkv_lds<TO, 1>(h, p, m0, n0, 0, wm2, wn, lr, lk, stage);
                    else if ((EPK == 0 || EPK == 2) && (EPK == 2 || p.out_planes)) x3p_epilogue_planes<1, TN, NP>(h, p, m0, n0, wm2, wn, lr, lk, stage);
                    else if constexpr (EPK == 0 || EPK == 3) gemm_epilogue_lds<TO, 1, TN, 32, WN>(h, p, m0, n0, 0, 0, wm2, wn, lr, lk, stage);
                }
            } else {
                gemm_epilogue<TO, 1, TN, 32, WN>(h, p, m0, n0, 0, 0, wm2, wn, lr, lk);
            }
            __syncthreads();
        }
        it += n;
    }
#endif
}

static std::atomic<long> g_x3p_noalign = 0, g_x3p_grid = 0;       // A/B switches: 1 = no cyclic K alignment ; grid: 0 = automatic, else GR (1, 2, 4, 8)
void x3p_set_option(int which, long v) { if (which == 0) g_x3p_noalign = v; else g_x3p_grid = v; }

void launch_linear_x3p(const ConvGemmDev& e_in, hipStream_t s) {
    ConvGemmDev e = e_in;
    MI_REQUIRE(!e.out_planes || e.lds_epi, "linear_x3p: out_planes needs the LDS-staged epilogue");
    int dev = 0, cus = 256;
    MI_HIP(hipGetDevice(&dev));
    {
        static int cu_count[16] = {0};
        if (!cu_count[dev & 15]) { hipDeviceProp_t pr; MI_HIP(hipGetDeviceProperties(&pr, dev)); cu_count[dev & 15] = pr.multiProcessorCount; }
        cus = cu_count[dev & 15];
    }
    // exact-fit data-parallel tiling (gemm_x3d.hip) when the output divides into whole rounds of the CUs; else stream-K below
    if (!(e.dbg & ~4)) {
        int tw, rgn, cgn, band;
        if (x3d_plan(e, cus, tw, rgn, cgn, band)) { launch_linear_x3d(e, tw, rgn, cgn, band, s); return; }
    }
    e.Tm = (e.M + 127) / 128; e.Tn = (e.N + 127) / 128;
    // XCD bands: GR x GC = 8.  Fabric-side bytes ~ GC * |A| + GR * |B|; the bands must also balance the tile counts
    // (ranges are cut per group).  Pick the split with the smallest worst-case group, then the smallest traffic.
    int best = 1; double best_cost = 1e300;
    for (int gr = 1; gr <= 8; gr *= 2) {
        const int gc = 8 / gr;
        if (g_x3p_grid && gr != g_x3p_grid) continue;
        long worst = 0, least = 1L << 60;
        for (int a = 0; a < gr; ++a)
            for (int b = 0; b < gc; ++b) {
                const long t = ((long)(a + 1) * e.Tm / gr - (long)a * e.Tm / gr) * ((long)(b + 1) * e.Tn / gc - (long)b * e.Tn / gc);
                worst = std::max(worst, t); least = std::min(least, t);
            }
        if (least == 0 && !g_x3p_grid) continue;
        const double traffic = (double)gc * e.M + (double)gr * e.N;       // x K x 6 bytes, common factor dropped
        const double cost = (double)worst * 1e9 + traffic;
        if (cost < best_cost) { best_cost = cost; best = gr; }
    }
    e.RT = best; e.RC = 8 / best;
    e.tail_tiles = (int)g_x3p_noalign;
    const int P = std::min(cus, e.sk_slots - 8) & ~7;          // the LAST flag word is the watchdog's error word (SkWorkspace::tripped)
    const dim3 grid(P);
    MI_REQUIRE(e.np == 2 || e.np == 3, "linear_x3p: 2 or 3 planes per operand");
    if (e.np == 2) {
#if defined(MI355TTS_TUNING)
#define X2_TUNE(D, LABEL) case D: prof_set_kernel("linear_x3p_kernel<float, true, 2, " LABEL ">", "", ""); hipLaunchKernelGGL((linear_x3p_kernel<float, true, 2, D>), grid, dim3(512), 0, s, e); MI_HIP(hipGetLastError()); return;
        if (e.lds_epi) switch (e.dbg & 59) {
            X2_TUNE(48, "MFMA on real operands only, no loop barriers") X2_TUNE(34, "no LDS reads, no loop barriers") X2_TUNE(32, "no loop barriers")
            X2_TUNE(1, "noDMA") X2_TUNE(2, "noLDSread") X2_TUNE(3, "noDMA noLDSread") X2_TUNE(8, "noMFMA") X2_TUNE(9, "noDMA noMFMA") X2_TUNE(10, "noLDSread noMFMA") X2_TUNE(16, "MFMA on real operands only")
            default: break;
        }
#undef X2_TUNE
#endif
        const bool fold = e.ln_stats_in || e.ln_stats_out;
        // one epilogue per instantiation (EPK): 1 QKV, 2 plain output (planes), 3 residual
        const int epk = e.epi == EPI_QKV_ROPE ? 1 : fold ? (e.ln_stats_in ? 2 : 3) : (e.out_planes ? 2 : 3);
        if (fold) {
            MI_REQUIRE(e.lds_epi, "linear_x3p: the AdaLN fold needs the LDS-staged epilogue");
            if (epk == 1) { prof_set_kernel("linear_x3p_kernel<float, true, 2, AdaLN fold, QKV>", "", ""); hipLaunchKernelGGL((linear_x3p_kernel<float, true, 2, 0, true, 1>), grid, dim3(512), 0, s, e); }
            else if (epk == 2) { prof_set_kernel("linear_x3p_kernel<float, true, 2, AdaLN fold, FF1>", "", ""); hipLaunchKernelGGL((linear_x3p_kernel<float, true, 2, 0, true, 2>), grid, dim3(512), 0, s, e); }
            else { prof_set_kernel("linear_x3p_kernel<float, true, 2, AdaLN fold, O / FF2>", "", ""); hipLaunchKernelGGL((linear_x3p_kernel<float, true, 2, 0, true, 3>), grid, dim3(512), 0, s, e); }
        }
        else if (e.lds_epi) {
            if (epk == 1) { prof_set_kernel("linear_x3p_kernel<float, true, 2, QKV>", "", ""); hipLaunchKernelGGL((linear_x3p_kernel<float, true, 2, 0, false, 1>), grid, dim3(512), 0, s, e); }
            else if (epk == 2) { prof_set_kernel("linear_x3p_kernel<float, true, 2, planes out>", "", ""); hipLaunchKernelGGL((linear_x3p_kernel<float, true, 2, 0, false, 2>), grid, dim3(512), 0, s, e); }
            else { prof_set_kernel("linear_x3p_kernel<float, true, 2>", "", ""); hipLaunchKernelGGL((linear_x3p_kernel<float, true, 2, 0, false, 3>), grid, dim3(512), 0, s, e); }
        }
        else { prof_set_kernel("linear_x3p_kernel<float, false, 2>", "", ""); hipLaunchKernelGGL((linear_x3p_kernel<float, false, 2>), grid, dim3(512), 0, s, e); }
    } else if (e.lds_epi) {
#if defined(MI355TTS_TUNING)
        switch (e.dbg & 27) {       // tuning instantiations (build.py --tuning, MI355TTS_GEMM_DBG)
            case 1: prof_set_kernel("linear_x3p_kernel<float, true, 3, noDMA>", "", ""); hipLaunchKernelGGL((linear_x3p_kernel<float, true, 3, 1>), grid, dim3(512), 0, s, e); MI_HIP(hipGetLastError()); return;
            case 2: prof_set_kernel("linear_x3p_kernel<float, true, 3, noLDSread>", "", ""); hipLaunchKernelGGL((linear_x3p_kernel<float, true, 3, 2>), grid, dim3(512), 0, s, e); MI_HIP(hipGetLastError()); return;
            case 3: prof_set_kernel("linear_x3p_kernel<float, true, 3, noDMA noLDSread>", "", ""); hipLaunchKernelGGL((linear_x3p_kernel<float, true, 3, 3>), grid, dim3(512), 0, s, e); MI_HIP(hipGetLastError()); return;
            case 8: prof_set_kernel("linear_x3p_kernel<float, true, 3, noMFMA>", "", ""); hipLaunchKernelGGL((linear_x3p_kernel<float, true, 3, 8>), grid, dim3(512), 0, s, e); MI_HIP(hipGetLastError()); return;
            case 16: prof_set_kernel("linear_x3p_kernel<float, true, 3, MFMA on real operands only>", "", ""); hipLaunchKernelGGL((linear_x3p_kernel<float, true, 3, 16>), grid, dim3(512), 0, s, e); MI_HIP(hipGetLastError()); return;
            default: break;
        }
#endif
        if (e.ln_stats_in || e.ln_stats_out) {
            prof_set_kernel("linear_x3p_kernel<float, true, 3, AdaLN fold>", "", "");
            hipLaunchKernelGGL((linear_x3p_kernel<float, true, 3, 0, true>), grid, dim3(512), 0, s, e);
        } else {
            prof_set_kernel("linear_x3p_kernel<float, true, 3>", "", "");
            hipLaunchKernelGGL((linear_x3p_kernel<float, true, 3>), grid, dim3(512), 0, s, e);
        }
    } else {
        MI_REQUIRE(!e.ln_stats_in && !e.ln_stats_out, "linear_x3p: the AdaLN fold needs the LDS-staged epilogue");
        prof_set_kernel("linear_x3p_kernel<float, false, 3>", "", "");
        hipLaunchKernelGGL((linear_x3p_kernel<float, false, 3>), grid, dim3(512), 0, s, e);
    }
    MI_HIP(hipGetLastError());
}

}  // namespace mi
