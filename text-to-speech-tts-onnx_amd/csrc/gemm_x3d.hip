// gemm_x3d.hip — the fp32 DiT linear layers on fp16 {hi, lo} pairs as an EXACT-FIT DATA-PARALLEL tiling (round 6).
//
// Why a second kernel beside gemm_x3p.hip.  The stream-K kernel balances any shape, and pays for it per launch: ~240 partial
// tiles of 64 KB travel to HBM and back (33 MB of the 60 MB its tail moves), the owners of a tile wait for them and run the
// whole epilogue on 144 - 256 CUs while the rest idle, and the K walks of its ranges make an XCD re-fetch its panels (TCC hit
// 64 %, 150 MB of fabric traffic per launch against ~40 MB algorithmic) — measured 12 - 17 us of a 31 - 67 us launch
// (profiles/r4/x3p_epilogue_cost.txt).  When the output area divides into whole rounds of the chip's CUs
// with tiles the matrix pipe likes, none of that is needed: every workgroup owns ONE tile over the WHOLE K, all workgroups
// walk K in lockstep (chunk c at the same time: an XCD fetches each operand panel once), nobody waits for anybody, and every
// CU runs a 1 / 256th of the epilogue at the same moment.
//
// Tile: 144 rows x TW columns, TW = 192 | 128 | 64 — at one utterance (M = 2 x 1126 = 2252 rows -> 16 row groups of 144)
// QKV / FF1 / O, FF2 are 16 x 16 = 256 tiles each: one round of 256 CUs, 97.7 % useful.  The launcher (x3d_plan) takes this
// kernel when some TW gives >= 90 % useful area over whole rounds, else the launch stays on the stream-K kernel.
//
// 144 = 9 x 16, so the blocks are v_mfma_f32_16x16x32_f16 (same rate as 32x32x16).  Twelve waves (three per SIMD, <= 168
// VGPRs): 3 along M x 4 along N (48 x 32 | 48 x 48 per wave), or for TW = 64 two k-groups of 3 x 2 waves taking alternate
// chunks (48 x 32 per wave; a 48 x 16 wave tile would read more LDS bytes than the pipe has cycles for).
// Operands are gemm_x3p.hip's panel planes (common.h x3p_slot_offset) unchanged: a 16-row group of one plane of one chunk
// is 1 KB contiguous, so a tile chunk is 18 + TW / 8 LDS-DMA pieces whatever panel boundary the 144 rows straddle.  The
// planes' XOR swizzle was made for 32-row fragments; 16-row fragments read it conflict-free when MFMA row quad q takes tile
// row quad sigma(q) = (0, 2, 3, 1)[q] (a permutation of rows inside a 16-row block, undone when the accumulators are staged).
// Per chunk and wave: phase 0 lo x hi, phase 1 hi x hi, phase 2 hi x lo (acc1 | acc0 | acc1: same order of the partial
// products as gemm_x3p.hip); ONE fragment register set: each fragment group is re-read for the next chunk right behind
// the phase that used it last (lo(A) after phase 0, hi(B) after 1, hi(A) + lo(B) after 2) and is next needed one to two
// phases later.  The ring: 6 | 4 | 3 chunk slots, one barrier per step (TW = 64: a step is the two k-groups' chunks), the DMA
// pieces of a step ride on the MFMAs of phase 1 two (TW = 128: three) steps ahead.
// Epilogue: the tile is staged in LDS as 32 x 64 blocks in the 32x32 accumulator layout's order, each wave takes blocks and
// runs gemm_epilogue.h's per-role epilogues unchanged (QKV + RoPE + V^T | FF1 planes + GELU | gated residual + AdaLN fold).
#include <atomic>
#include "common.h"
#include "mfma.h"
#include "gemm_epilogue.h"
#include "x3_split.h"
#include <cstdlib>
#include <type_traits>
#include <algorithm>

namespace mi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int TW> struct X3dGeom {
    static constexpr int TM = 144, MB = 3;
    static constexpr int KG = TW == 64 ? 2 : 1;                   // k-groups (alternate chunks)
    static constexpr int WNS = KG == 2 ? 2 : 4;                   // waves along N
    static constexpr int WNW = TW / WNS, NB = WNW / 16;           // 32 | 32 | 48 columns per wave
#ifndef X3D_S128
#define X3D_S128 1          // TW = 128: chunks per step (1: a barrier per chunk, the DMA three chunks ahead — FF1 38.3 -> 37.4 us with the layer's
                            // weights cold in HBM, bit-identical, profiles/r6/x3d_s128_ab.txt; 2: a barrier per two chunks, one step ahead)
#endif
    static constexpr int S = TW == 192 ? 1 : TW == 128 ? X3D_S128 : 2;       // chunks per step (between barriers)
    static constexpr int WSTEP = TW == 128 ? X3D_S128 : 1;        // ... of which one wave multiplies
    static constexpr int R = TW == 64 ? 6 : TW == 128 ? 4 : 3;    // ring slots (chunks)
    static constexpr int LB = R / S - 1;                          // batches in flight behind the one being waited for: 2 | 1 | 2
    static constexpr int A_PLANE = TM * 64, B_PLANE = TW * 64;
    static constexpr int A_BYTES = 2 * A_PLANE, CH = A_BYTES + 2 * B_PLANE;     // 26 | 34 | 42 KB per chunk slot
    static constexpr int PC = CH / 1024, P = S * PC;              // DMA pieces per chunk / per step: 52 | 68 | 42
    static constexpr int NHI = (P + 11) / 12, NHIW = P - 12 * (NHI - 1);       // pieces per wave: NHI for waves < NHIW, else NHI - 1
    static constexpr int NCB = TW / 64, NRB = 5, BLK = 2176;      // staged 32 x 64 blocks (8704 B each: the V^T path wants 32 x 65)
    static constexpr int STG = NRB * NCB * BLK * 4, RING = R * CH;
    static constexpr int LNT = RING > STG ? RING : STG;           // behind both: (rstd, mean * rstd) of the tile's rows, [2][160] floats
    static constexpr int SMEM = LNT + 2 * 160 * 4;
    static_assert(SMEM <= 163840, "x3d: LDS");
};

// one 1 KB piece: 64 lanes x 16 bytes from byte offset voff (per lane; beyond the buffer: zeros, nothing fetched) to LDS byte
// address lds_dst + 16 * lane.  The whole offset travels in the VGPR: the range check of a raw buffer does not see soffset.
template <typename RSRC>
__device__ __forceinline__ void x3d_dma(RSRC rsrc, int voff, unsigned lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
#endif
}
template <int N> __device__ __forceinline__ void x3d_wait_lgkm() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory");
#endif
}
// N fragment reads (16 bytes per lane, 1 KB apart) from LDS byte address a + OFF: inline asm, NOT waited for by the compiler
template <int OFF> __device__ __forceinline__ void x3d_lds_rd128(f16x8& d, unsigned a) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF) : "memory");
#endif
}
template <int OFF, int N> struct X3dRd {
    static __device__ __forceinline__ void go(f16x8 (&dst)[N], unsigned a) {
        X3dRd<OFF, N - 1>::go(reinterpret_cast<f16x8 (&)[N - 1]>(dst), a);
        x3d_lds_rd128<OFF + (N - 1) * 1024>(dst[N - 1], a);
    }
};
template <int OFF> struct X3dRd<OFF, 1> {
    static __device__ __forceinline__ void go(f16x8 (&dst)[1], unsigned a) { x3d_lds_rd128<OFF>(dst[0], a); }
};
template <int N, int I = 0, typename F> __device__ __forceinline__ void x3d_unroll(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); x3d_unroll<N, I + 1>(f); }
}
template <int N> __device__ __forceinline__ void x3d_wait_vm() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
#endif
}

// p.x = A panel planes, p.w3 = B panel planes (np = 2), p.Tm x p.Tn = row groups x column groups, p.RT = band height of the
// tile order (row groups an XCD's 32 consecutive tiles span), p.dbg bit 2: no epilogue (tuning)
template <int TW, bool FOLD, int EPK>
__global__ __launch_bounds__(768) void linear_x3d_kernel(const ConvGemmDev p) {
    using G = X3dGeom<TW>;
    using Frag = f16x8;
    constexpr int MB = G::MB, NB = G::NB;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[G::SMEM];
    (void)smem;
#if defined(__HIP_DEVICE_COMPILE__)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- this workgroup's tile: XCD x (blockIdx % 8) takes a contiguous run of the banded tile order --------------------
    const int per = (int)gridDim.x >> 3;
    const int u = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    const int RGn = p.Tm, CGn = p.Tn, band = p.RT;
    if (u >= RGn * CGn) return;
    const int bi = u / (band * CGn);
    const int rem = u - bi * band * CGn;
    const int bh = min(band, RGn - bi * band);
    const int cg = rem / bh, rg = bi * band + rem - cg * bh;
    const int m0 = rg * G::TM, n0 = cg * TW;
    const int nch = p.K >> 5;

    const int kg = G::KG == 2 ? wave / 6 : 0;
    const int w6 = G::KG == 2 ? wave - kg * 6 : wave;
    const int wm = w6 / G::WNS, wn = w6 - wm * G::WNS;

    // ---- LDS-DMA pieces of this wave: piece q = wave + 12 i of the step's P (chunk-in-step, operand, plane, 16-row group) -----
    const int bytesA = (int)((long)((p.M + 127) >> 7) * nch * 16384), bytesB = (int)((long)((p.N + 127) >> 7) * nch * 16384);
    const unsigned smem_lds = (unsigned)(unsigned long)(const __attribute__((address_space(3))) void*)smem;
    const int voff = lane * 16;
    constexpr int OOB = 0x7fffff00;
    int pc_gbase[G::NHI], pc_cs[G::NHI];
    unsigned pc_lofs[G::NHI];
    __amdgpu_buffer_rsrc_t pc_rs[G::NHI];          // the piece's operand: one descriptor per piece, no branch at issue time
#pragma unroll
    for (int i = 0; i < G::NHI; ++i) {
        const int q = wave + 12 * i;
        const int cs = q / G::PC, r = q - cs * G::PC;
        const bool isA = r < 18;
        const int rr = isA ? r : r - 18;
        const int gpp = isA ? 9 : TW / 16;
        const int plane = rr / gpp, g16 = rr - plane * gpp;
        const int row = (isA ? m0 : n0) + 16 * g16;
        pc_cs[i] = cs;
        pc_rs[i] = __builtin_amdgcn_make_buffer_rsrc(isA ? (void*)p.x : (void*)p.w3, 0, isA ? bytesA : bytesB, 0x00020000);
        pc_gbase[i] = __builtin_amdgcn_readfirstlane((row >> 7) * nch * 16384 + plane * 8192 + (row & 127) * 64);
        pc_lofs[i] = (unsigned)__builtin_amdgcn_readfirstlane(isA ? plane * G::A_PLANE + g16 * 1024 : G::A_BYTES + plane * G::B_PLANE + g16 * 1024);
    }
    const bool hi_wave = wave < G::NHIW;                             // this wave issues NHI pieces per step (else NHI - 1)
    // piece i of the step-t batch (past the last step: zero fill, nothing fetched); pieces are issued one per MFMA slot of phase 1
    auto issue_piece = [&](int t, auto I_) __attribute__((always_inline)) {
        constexpr int i = decltype(I_)::value;
        if constexpr (i < G::NHI) {
            if (i == G::NHI - 1 && !hi_wave) return;
            const int chunk = t * G::S + pc_cs[i];
            const int soff = __builtin_amdgcn_readfirstlane(chunk < nch ? pc_gbase[i] + chunk * 16384 : OOB);
            const unsigned dst = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)((chunk % G::R) * G::CH) + pc_lofs[i]);
            x3d_dma(pc_rs[i], (int)((unsigned)voff + (unsigned)soff), dst);
        }
    };
    auto issue_batch = [&](int t) __attribute__((always_inline)) {
        x3d_unroll<G::NHI>([&](auto I_) __attribute__((always_inline)) { issue_piece(t, I_); });
    };
    auto wait_batches = [&](auto NB_) __attribute__((always_inline)) {    // all but the NB_ youngest batches of this wave have landed
        constexpr int n = decltype(NB_)::value;
        if (hi_wave) x3d_wait_vm<n * G::NHI>(); else x3d_wait_vm<n * (G::NHI - 1)>();
    };

    // ---- fragment addresses: MFMA row i = lane & 15 <-> row 4 sigma(i >> 2) + (i & 3) of the 16-row block, k-slot lane >> 4 ------
    const int fi = lane & 15, fs = lane >> 4;
    const int sig = (0x1320 >> (4 * (fi >> 2))) & 3;
    const unsigned f_lane = (unsigned)((4 * sig + (fi & 3)) * 64 + ((fs ^ sig) << 4));
    const unsigned fa_off = (unsigned)(wm * 48 * 64) + f_lane;
    const unsigned fb_off = (unsigned)(G::A_BYTES + wn * G::WNW * 64) + f_lane;
    Frag a_lo[MB], a_hi[MB], b_lo[NB], b_hi[NB];
    f32x4 acc0[MB][NB], acc1[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) { acc0[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // Fragment reads are inline asm with hand-counted lgkmcnt waits (X3D_ASM_RD, the default): with compiler-visible ds_reads
    // under sched_barriers hipcc puts s_waitcnt lgkmcnt(0) in front of phase 0 — behind the reads it has just issued for phases
    // 1 and 2 — and the wave stalls for a full LDS round trip per chunk.  LDS operations of a wave complete in order: a wait
    // for "all but the N youngest" is exact.  (a: LDS byte address of the A part of the slot + this lane; b: of the B part.)
    auto rd_a = [&](unsigned a, auto PL, Frag (&dst)[MB]) __attribute__((always_inline)) {
        constexpr int pl = decltype(PL)::value;
        X3dRd<pl * G::A_PLANE, MB>::go(dst, a);
    };
    auto rd_b = [&](unsigned b, auto PL, Frag (&dst)[NB]) __attribute__((always_inline)) {
        constexpr int pl = decltype(PL)::value;
        X3dRd<pl * G::B_PLANE, NB>::go(dst, b);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    auto mma = [&](const Frag (&a)[MB], const Frag (&b)[NB], f32x4 (&c)[MB][NB]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], c[i][j], 0, 0, 0);
    };
#define X3D_SB() __builtin_amdgcn_sched_barrier(0)

    // ---- prologue: every slot of the ring requested, step 0 landed, the first chunk's fragments read ---------------------------
    const int NJ = nch / G::KG;                                     // chunks this wave multiplies: global chunk j * KG + kg
#pragma unroll
    for (int t = 0; t <= G::LB; ++t) issue_batch(t);
    // AdaLN fold, consumer side (QKV / FF1): the LayerNorm statistics of the tile's rows, finished ONCE per tile by waves 0 - 4 (32
    // rows each) behind the first DMA requests and parked in LDS — the epilogues (any column block, any wave) read two floats per row
    float* lnt = reinterpret_cast<float*>(smem + G::LNT);
    constexpr bool CONSUMER = FOLD && (EPK == 1 || EPK == 2);
    if constexpr (CONSUMER) {
        if (wave < G::NRB && p.ln_stats_in) {
            float rs, mr;
            ln_rows32(p, (long)p.m_off + m0 + wave * 32, (long)p.m_off + p.M - 1, lane & 31, lane >> 5, rs, mr);
            if (lane < 32) { lnt[wave * 32 + lane] = rs; lnt[160 + wave * 32 + lane] = mr; }
        }
    }
    wait_batches(std::integral_constant<int, G::LB>{});
    __builtin_amdgcn_s_barrier();
    {
        const unsigned s0 = smem_lds + (unsigned)((kg % G::R) * G::CH);
        rd_a(s0 + fa_off, P1{}, a_lo); rd_b(s0 + fb_off, P0{}, b_hi); rd_a(s0 + fa_off, P0{}, a_hi); rd_b(s0 + fb_off, P1{}, b_lo);
    }
    // One chunk.  BAR: the next chunk opens a step — behind phase 0 its batch has landed for every wave and every wave is done with
    // the step behind us, whose slots take the batch LB steps ahead: those pieces go out one per MFMA of phase 1 (a piece costs
    // 60 - 180 cycles of issue: in a block of their own behind the barrier, with all three waves of a SIMD at the same point, the
    // matrix pipe idled for them).  Past the last chunk the barrier block runs once more: zero fill into a dead slot.
    auto chunk = [&](int j, auto BAR_) __attribute__((always_inline)) {
        constexpr bool BAR = decltype(BAR_)::value;
        x3d_wait_lgkm<MB + NB>();                                   // lo(A), hi(B) of this chunk are in (hi(A), lo(B) may be on their way)
        X3D_SB(); mma(a_lo, b_hi, acc1); X3D_SB();                  // phase 0: lo(A) x hi(B)
        const int jn = j + 1;
        const int ts = jn / G::WSTEP;
        if constexpr (BAR) {
            wait_batches(std::integral_constant<int, G::LB - 1>{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        const unsigned sn = smem_lds + (unsigned)(((jn * G::KG + kg) % G::R) * G::CH);
        rd_a(sn + fa_off, P1{}, a_lo);
        x3d_wait_lgkm<MB>();                                        // hi(A), lo(B) are in
        X3D_SB();
        x3d_unroll<MB * NB>([&](auto IDX_) __attribute__((always_inline)) {       // phase 1: hi x hi
            constexpr int idx = decltype(IDX_)::value, i = idx / NB, jj = idx % NB;
            acc0[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[i], b_hi[jj], acc0[i][jj], 0, 0, 0);
            if constexpr (BAR) issue_piece(ts + G::LB, IDX_);
            X3D_SB();
        });
        rd_b(sn + fb_off, P0{}, b_hi);
        X3D_SB(); mma(a_hi, b_lo, acc1); X3D_SB();                  // phase 2: hi(A) x lo(B)
        rd_a(sn + fa_off, P0{}, a_hi); rd_b(sn + fb_off, P1{}, b_lo);
    };
    static_assert(G::NHI <= MB * NB, "x3d: a batch's pieces ride on the MFMAs of phase 1");
    for (int j = 0; j < NJ; j += G::WSTEP) {
        if constexpr (G::WSTEP == 2) chunk(j, std::false_type{});
        chunk(j + G::WSTEP - 1, std::true_type{});
    }
#undef X3D_SB
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (p.dbg & 4) return;                                         // tuning: main loop only

    // ---- the tile into LDS as 32 x 64 blocks [row block][column block][32][64] (+ pad): acc0 + 2^-11 acc1 ---------------------
    float* stg = reinterpret_cast<float*>(smem);
    {
        const int rq = 4 * ((0x1320 >> (4 * (lane >> 4))) & 3);                        // tile row of register 0 inside the 16-row block
        const int cq = 4 * ((0x1320 >> (4 * ((lane & 15) >> 2))) & 3) + (lane & 3);    // tile column inside the 16-column block
        auto put = [&](bool add) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const int row = wm * 48 + i * 16 + rq;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int col = wn * G::WNW + j * 16 + cq;
                    float* d = stg + ((row >> 5) * G::NCB + (col >> 6)) * G::BLK + (row & 31) * 64 + (col & 63);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = __builtin_fmaf(acc1[i][j][r], 0x1p-11f, acc0[i][j][r]);
                        d[r * 64] = add ? d[r * 64] + v : v;
                    }
                }
            }
        };
        if (G::KG == 1 || kg == 0) put(false);
        __syncthreads();
        if constexpr (G::KG == 2) {
            if (kg == 1) put(true);
            __syncthreads();
        }
    }
    // ---- blocks -> waves -> the per-role epilogues of gemm_epilogue.h (32x32 accumulator layout, wave-private staging) ---------
    const int lr = lane & 31, lk = lane >> 5;
    const int m_end = min(p.M, m0 + G::TM);                        // rows 144 .. 159 of the last row block belong to the next tile
    constexpr bool HALVES = EPK == 3 && (G::NRB * G::NCB * G::BLK + 12 * 1024) * 4 <= G::LNT;      // (room for the wave-private staging: TW = 64 | 128)
    if constexpr (HALVES) {
        // O / FF2 (and plain row outputs): 32 x 32 half blocks, so that a 144 x 64 tile keeps ten waves busy instead of five — the
        // epilogue of these launches is a chain of LDS hops, residual loads and stores per wave, not a byte count (measured in the
        // model with MI355TTS_GEMM_DBG=4: 8.8 us per launch for 28 MB)
        float* half_stage = stg + G::NRB * G::NCB * G::BLK;            // wave-private 32 x 32 staging behind the tile (inside the dead ring)
        for (int u = wave; u < G::NRB * G::NCB * 2; u += 12) {
            const int b = u >> 1, hf = u & 1;
            const int rb = b / G::NCB, cb = b - rb * G::NCB;
            if (m0 + rb * 32 >= m_end) continue;
            const float* blk = stg + b * G::BLK + hf * 32;
            f32x16 h1[1][1];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float t = blk[((r & 3) + 8 * (r >> 2) + 4 * lk) * 64 + lr];
                h1[0][0][r] = (rb == G::NRB - 1 && r >= 8) ? 0.f : t;     // (rows 144 .. 159 were never staged: see below)
            }
            float* st = half_stage + wave * 1024;
            if constexpr (FOLD) gemm_epilogue_resid_ln<float, 1, 1, 2>(h1, p, m0 + rb * 32, n0 + cb * 64 + hf * 32, lr, lk, st, m_end);
            else gemm_epilogue_lds<float, 1, 1, 32, 32>(h1, p, m0, n0, 0, 0, rb, cb * 2 + hf, lr, lk, st, m_end);
        }
        return;
    }
    for (int b = wave; b < G::NRB * G::NCB; b += 12) {
        const int rb = b / G::NCB, cb = b - rb * G::NCB;
        if (m0 + rb * 32 >= m_end) continue;
        float* blk = stg + b * G::BLK;
        float prs[1] = {0.f}, pmr[1] = {0.f};
        if constexpr (CONSUMER) { prs[0] = lnt[rb * 32 + lr]; pmr[0] = lnt[160 + rb * 32 + lr]; }
        f32x16 h[1][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // the last row block holds 16 staged rows (144 = 4 x 32 + 16): rows 16 .. 31 of its LDS block were never written — whatever
                // the operand ring left there must not reach the epilogue's range watch (their stores are masked anyway)
                const float t = blk[((r & 3) + 8 * (r >> 2) + 4 * lk) * 64 + j * 32 + lr];
                h[0][j][r] = (rb == G::NRB - 1 && r >= 8) ? 0.f : t;
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if constexpr (FOLD) {
            if constexpr (EPK == 1) gemm_epilogue_qkv_lds<float, 1, true, true>(h, p, m0, n0, 0, rb, cb, lr, lk, blk, prs, pmr, m_end);
            else if constexpr (EPK == 2) gemm_epilogue_ln_in<float, 1, 2, 2, true>(h, p, m0 + rb * 32, n0 + cb * 64, lr, lk, blk, prs, pmr, m_end);
            else gemm_epilogue_resid_ln<float, 1, 2, 2>(h, p, m0 + rb * 32, n0 + cb * 64, lr, lk, blk, m_end);
        } else {
            if constexpr (EPK == 1) gemm_epilogue_qkv_lds<float, 1>(h, p, m0, n0, 0, rb, cb, lr, lk, blk, nullptr, nullptr, m_end);
            else if constexpr (EPK == 2) x3p_epilogue_planes<1, 2, 2>(h, p, m0, n0, rb, cb, lr, lk, blk, m_end);
            else gemm_epilogue_lds<float, 1, 2, 32, 64>(h, p, m0, n0, 0, 0, rb, cb, lr, lk, blk, m_end);
        }
    }
#endif
}

static std::atomic<long> g_x3d = 1, g_x3d_min_eff = 90;       // options gemm_x3d (0 off, 1 automatic), gemm_x3d_min_eff (per cent of useful tile area)
void x3d_set_option(int which, long v) { if (which == 0) g_x3d = v; else g_x3d_min_eff = v; }

// Does an exact-fit tiling exist for this launch?  tw: tile width, rgn x cgn tiles, band: row groups per band of the tile order.
bool x3d_plan(const ConvGemmDev& e, int cus, int& tw, int& rgn, int& cgn, int& band) {
    static int env_read = 0;
    if (!env_read) {
        env_read = 1;
        if (const char* s = std::getenv("MI355TTS_X3D")) g_x3d = std::atol(s);
        if (const char* s = std::getenv("MI355TTS_X3D_MIN_EFF")) g_x3d_min_eff = std::atol(s);
    }
    if (!g_x3d || e.np != 2 || !e.lds_epi || e.K % 32 != 0 || cus < 8) return false;
    const int nch = e.K / 32;
    double best = 0.0; int btw = 0;
    for (int w : {192, 128, 64}) {
        if (e.N % w != 0) continue;
        if (w != 192 && nch % 2 != 0) continue;
        if (e.epi == EPI_QKV_ROPE && w % 64 != 0) continue;
        const long tiles = (long)((e.M + 143) / 144) * (e.N / w);
        const long rounds = (tiles + cus - 1) / cus;
        const double eff = (double)e.M * e.N / ((double)rounds * cus * 144.0 * w);
        if (eff > best + 1e-9) { best = eff; btw = w; }         // ties: the wider tile (fewer L2 -> LDS bytes per flop)
    }
    if (!btw || best * 100.0 < (double)g_x3d_min_eff) return false;
    tw = btw; rgn = (e.M + 143) / 144; cgn = e.N / btw;
    // band height a: an XCD's 32 consecutive tiles span a row groups x 32 / a column groups; fabric bytes ~ a * 144 + (32 / a) * tw rows
    int ba = 1; double bc = 1e300;
    for (int a = 1; a <= 32; a *= 2) {
        if (a > rgn && a > 1) break;
        const double c = (double)std::min(a, rgn) * 144.0 + (double)std::min(32 / a, cgn) * tw;
        if (c < bc) { bc = c; ba = a; }
    }
    band = std::min(ba, rgn);
    return true;
}

void launch_linear_x3d(const ConvGemmDev& e_in, int tw, int rgn, int cgn, int band, hipStream_t s) {
    ConvGemmDev e = e_in;
    e.Tm = rgn; e.Tn = cgn; e.RT = band;
    const long tiles = (long)rgn * cgn;
    const dim3 grid((unsigned)(((tiles + 7) / 8) * 8));
    const bool fold = e.ln_stats_in || e.ln_stats_out;
    const int epk = e.epi == EPI_QKV_ROPE ? 1 : fold ? (e.ln_stats_in ? 2 : 3) : (e.out_planes ? 2 : 3);
#define X3D_GO(TW_, FOLD_, EPK_, LABEL)                                                                                           \
    do {                                                                                                                          \
        prof_set_kernel("linear_x3d_kernel<" #TW_ ", " LABEL ">", "", "");                                                        \
        hipLaunchKernelGGL((linear_x3d_kernel<TW_, FOLD_, EPK_>), grid, dim3(768), 0, s, e);                                      \
    } while (0)
#define X3D_ROLE(TW_)                                                                                                             \
    do {                                                                                                                          \
        if (fold) { if (epk == 1) X3D_GO(TW_, true, 1, "AdaLN fold, QKV"); else if (epk == 2) X3D_GO(TW_, true, 2, "AdaLN fold, FF1"); else X3D_GO(TW_, true, 3, "AdaLN fold, O / FF2"); } \
        else { if (epk == 1) X3D_GO(TW_, false, 1, "QKV"); else if (epk == 2) X3D_GO(TW_, false, 2, "planes out"); else X3D_GO(TW_, false, 3, "rows out"); } \
    } while (0)
    if (tw == 192) X3D_ROLE(192); else if (tw == 128) X3D_ROLE(128); else X3D_ROLE(64);
#undef X3D_ROLE
#undef X3D_GO
    MI_HIP(hipGetLastError());
}

}  // namespace mi
