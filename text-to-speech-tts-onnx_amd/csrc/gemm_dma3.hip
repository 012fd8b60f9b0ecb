// gemm_dma3.hip — 16-bit implicit-GEMM main loop v3 (256-row tiles, 8 waves) of the conv / linear kernel family; split
// from gemm_conv.hip so that the two translation units compile in parallel.  Dispatch lives in gemm_conv.hip.
#include "common.h"
#include "mfma.h"
#include "gemm_epilogue.h"

namespace mi {

// ---------------------------------------------------------------------------------------------------
// 16-bit main loop v3: 256x128 tile, 8 waves (two per SIMD, so one wave's LDS/barrier stalls sit under the
// other's MFMAs), three-stage LDS ring filled by LDS-DMA two chunks ahead.  A lone workgroup of the 2-stage
// kernel measured ~1700 cycles per K chunk against 512 cycles of MFMA (DMA issue -> landed -> barrier ->
// ds_read is a serial chain); here the DMA for chunk c+2 is issued before the MFMAs of chunk c, the wait is a
// COUNTED s_waitcnt vmcnt(6) (= this wave's six DMA instructions of chunk c+2 may stay in flight) and the
// barrier is the raw s_barrier, so nothing drains the queue (a __syncthreads() would wait vmcnt(0)).
// Swizzle: slot = kv ^ ((row >> 1) & 7): rows of equal parity inside every ds_read_b128 lane group get eight
// distinct slots => conflict-free fragment reads.
// ---------------------------------------------------------------------------------------------------
template <typename T, typename TO, int BM, int BN, int WM, int WN, int NST, bool BUF = false>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) void conv_gemm_dma3_kernel(const ConvGemmDev p) {
    using MF = Mfma<T>;
    // two configurations: 256x128 tile / 64x64 per wave / 3-stage ring (default) and 256x256 tile / 128x64 per wave /
    // 2-stage ring (1.5x fewer DMA bytes per flop, for problems that fill every CU: the per-CU L2->LDS fill rate,
    // ~22 B/cycle measured, is what bounds this kernel)
    constexpr int KC = 64, TM = WM / 32, TN = WN / 32, WGN = BN / WN;
    // NW = 8 waves (two per SIMD), or NW = 4 waves of 128x128 (one per SIMD, 256 accumulator registers: a third fewer
    // LDS fragment bytes per MFMA than the 128x64 wave tile)
    constexpr int NW = (BM / WM) * WGN;
    static_assert(NW == 8 || NW == 4, "eight or four waves");
    constexpr int TILE = (BM + BN) * KC;
    constexpr int AJ = BM / 8 / NW, BJ = BN / 8 / NW, PERW = AJ + BJ;     // 8-row DMA groups per wave per chunk
    static_assert(NST == 2 || PERW == 6 || PERW == 8, "counted vmcnt immediates of the 3-stage ring");
    __shared__ __attribute__((aligned(1024))) T smem[NST * TILE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN, lr = lane & 31, lk = lane >> 5;
    const int L = blockIdx.x;
    int nt, rowt;
    if (p.RC == 2) {
        // XCD-aware, N tiles fastest: workgroup L lands on XCD L % 8; each XCD walks a contiguous range of the (row tile, N tile)
        // list with the N tiles of one row tile next to each other, so the second .. Tn-th reading of an x row panel hits the
        // XCD's own L2 instead of the fabric (row tiles fastest re-reads every panel one block round later: PMC 2.1x over-fetch)
        const int total = p.RT * p.Tn, per = (total + 7) >> 3;
        const int t = (L & 7) * per + (L >> 3);
        if ((L >> 3) >= per || t >= total) return;
        rowt = t / p.Tn; nt = t - rowt * p.Tn;
    } else {
        nt = L / p.RT; rowt = L - nt * p.RT;               // row tiles fastest: neighbours share the weight panel
    }
    const int b = rowt / p.Tm, mt = rowt - b * p.Tm;
    const int m0 = mt * BM, n0 = nt * BN;
    const int g = blockIdx.y;
    const T* xb = (const T*)p.x + (long)b * p.x_bstride + (long)g * p.x_goff;
    const T* wg = (const T*)p.w + (long)g * p.N * p.K;
    const T* zero = (const T*)p.zero;

    const int lrow = lane >> 3;
    const int kv0 = (lane & 7) ^ ((lane >> 4) & 7);               // even 8-row groups
    const int kv1 = (lane & 7) ^ ((4 + (lane >> 4)) & 7);         // odd 8-row groups
    const int nchunks = (p.K / p.Cin) * ((p.Cin + KC - 1) / KC);

    // BUF: the DMA goes through buffer descriptors (buffer_load_dwordx4 ... lds).  A lane whose offset falls outside
    // [0, num_records) gets ZEROS written to its LDS slot by the hardware range check (tools/ubench/bufload_lds.hip), so
    // the time padding (t < 0 or t >= T_in) and the N tail need no per-lane select, and the per-lane part of the address
    // (row * stride + swizzled k-vector) is loop-invariant: one v_add per DMA instruction instead of ~20 VALU / SALU
    // instructions (two 64-bit multiply-adds, three compares, exec masking) of the flat-address form.  Requires
    // Cin % 64 == 0 (a K tail inside a valid row would read the neighbouring row) — checked by the dispatcher.
    // (the buffer-resource type and builtins exist in the device pass only: the host pass, which just needs the launch
    // stub, must not see them)
#if defined(__HIP_DEVICE_COMPILE__)
    __amdgpu_buffer_rsrc_t rsa, rsb;
    int avo[AJ], bvo[BJ];                                         // per-lane byte offsets, loop-invariant
    if constexpr (BUF) {
        rsa = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)((((long)p.T_in - 1) * p.x_rstride + p.Cin) * (long)sizeof(T)), 0x00020000);
        rsb = __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, (int)((long)p.N * p.K * (long)sizeof(T)), 0x00020000);
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int R0 = (wave * AJ + j) * 8;
            avo[j] = (int)(((long)(m0 + R0 + lrow) * p.x_rstride + (((wave * AJ + j) & 1) ? kv1 : kv0) * 8) * (long)sizeof(T));
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int R0 = (wave * BJ + j) * 8;
            const long n = n0 + R0 + lrow;
            // rows past N: any offset >= num_records (a clamped product cannot wrap into range)
            bvo[j] = n < p.N ? (int)((n * p.K + (((wave * BJ + j) & 1) ? kv1 : kv0) * 8) * (long)sizeof(T)) : 0x7fffff00;
        }
    }
#endif
    auto issue = [&](int st, int tap, int c0) {
        T* base = smem + st * TILE;
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (BUF) {
            const int ca = (int)(((long)(tap * p.dil - p.pad) * p.x_rstride + c0) * (long)sizeof(T));
            const int cb = (tap * p.Cin + c0) * (int)sizeof(T);
#pragma unroll
            for (int j = 0; j < AJ; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_void*)(base + (wave * AJ + j) * 8 * KC), 16, avo[j] + ca, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < BJ; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lds_void*)(base + (BM + (wave * BJ + j) * 8) * KC), 16, bvo[j] + cb, 0, 0, 0);
            return;
        }
#endif
        const int toff = tap * p.dil - p.pad;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int R0 = (wave * AJ + j) * 8;
            const int ci = c0 + (((wave * AJ + j) & 1) ? kv1 : kv0) * 8;
            const int t = m0 + R0 + lrow + toff;
            const T* src = (ci < p.Cin && t >= 0 && t < p.T_in) ? xb + (long)t * p.x_rstride + ci : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (lds_void*)(base + R0 * KC), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int R0 = (wave * BJ + j) * 8;
            const int ci = c0 + (((wave * BJ + j) & 1) ? kv1 : kv0) * 8;
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

    int itap = 0, ic0 = 0;                                        // cursor of the NEXT chunk to issue
    const int ntaps = p.K / p.Cin;
    auto advance = [&]() { if (++itap >= ntaps) { itap = 0; ic0 += KC; } };   // K order = (channel chunk, tap)
    auto wait_next = [&](bool more_in_flight) {
        // everything except (optionally) this wave's newest PERW DMA instructions has landed
        if (more_in_flight) {
            if constexpr (PERW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    constexpr int AHEAD = NST - 1;                                // chunks in flight beyond the one being computed
    issue(0, itap, ic0); advance();
    if (AHEAD == 2 && nchunks > 1) { issue(1, itap, ic0); advance(); wait_next(true); }
    else wait_next(false);
    __builtin_amdgcn_s_barrier();

    if constexpr (NST == 2) {
        // Two stages, software-pipelined ACROSS chunks: the wait + barrier that publish chunk c+1 sit before the LAST k-step
        // of chunk c (whose fragments are already in registers), so the first fragments of chunk c+1 are fetched, and the
        // DMA of chunk c+2 is issued into chunk c's stage, under MFMAs that are ready to issue — the lockstep form left
        // an LDS round trip (and the barrier skew) exposed in front of the first MFMA of every chunk.
        typename MF::Frag fa[2][TM], fb[2][TN];
        auto ldfrag = [&](const T* As, int ks, int set) {
            const T* Bs = As + BM * KC;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * WM + i * 32 + lr;
                fa[set][i] = *reinterpret_cast<const typename MF::Frag*>(As + row * KC + (((ks * 2 + lk) ^ ((row >> 1) & 7)) << 3));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * WN + j * 32 + lr;
                fb[set][j] = *reinterpret_cast<const typename MF::Frag*>(Bs + row * KC + (((ks * 2 + lk) ^ ((row >> 1) & 7)) << 3));
            }
        };
        auto mmas = [&](int set) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma(fa[set][i], fb[set][j], acc[i][j]);
        };
        if (nchunks > 1) { issue(1, itap, ic0); advance(); }
        ldfrag(smem, 0, 0);
        for (int c = 0; c < nchunks; ++c) {
            const T* As = smem + (c & 1) * TILE;
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                ldfrag(As, ks + 1, (ks + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);                // keep the next k-step's reads AHEAD of these MFMAs
                mmas(ks & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (c + 1 < nchunks) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // chunk c+1 (issued a chunk period ago) has landed
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");// every read of chunk c's stage has retired
                __builtin_amdgcn_s_barrier();
                if (c + 2 < nchunks) { issue(c & 1, itap, ic0); advance(); }
                ldfrag(smem + ((c + 1) & 1) * TILE, 0, 0);        // register set 0 is free: k-step 3 runs from set 1
            }
            __builtin_amdgcn_sched_barrier(0);
            mmas(1);
            __builtin_amdgcn_sched_barrier(0);
        }
        gemm_epilogue<TO, TM, TN, WM, WN>(acc, p, m0, n0, b, g, wm, wn, lr, lk);
        return;
    }
    int st = 0;
    for (int c = 0; c < nchunks; ++c) {
        const bool pre = c + AHEAD < nchunks;
        if (pre) {
            int st2 = st + AHEAD; if (st2 >= NST) st2 -= NST;
            if (p.dbg != 1) issue(st2, itap, ic0);
            advance();
        }
        const T* As = smem + st * TILE;
        const T* Bs = As + BM * KC;
        if (p.dbg != 2) {
            // fragment loads run one k-step ahead of the MFMAs (two register sets): the ~300-cycle ds_read_b128
            // latency is then covered by this wave's MFMAs plus the partner wave's on the same SIMD
            typename MF::Frag fa[2][TM], fb[2][TN];
            auto ldfrag = [&](int ks, int set) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = wm * WM + i * 32 + lr;
                    fa[set][i] = *reinterpret_cast<const typename MF::Frag*>(As + row * KC + (((ks * 2 + lk) ^ ((row >> 1) & 7)) << 3));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = wn * WN + j * 32 + lr;
                    fb[set][j] = *reinterpret_cast<const typename MF::Frag*>(Bs + row * KC + (((ks * 2 + lk) ^ ((row >> 1) & 7)) << 3));
                }
            };
            constexpr bool PIPE = true;                        // (unpipelined was measured 4% slower even on the 128x64 wave tile)
            ldfrag(0, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (PIPE && ks + 1 < 4) ldfrag(ks + 1, (ks + 1) & 1);
                if (PIPE) __builtin_amdgcn_sched_barrier(0);  // keep the next k-step's reads AHEAD of these MFMAs
                constexpr int dummy = 0; (void)dummy;
                const int set = PIPE ? (ks & 1) : 0;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma(fa[set][i], fb[set][j], acc[i][j]);
                if (PIPE) __builtin_amdgcn_sched_barrier(0);
                if (!PIPE && ks + 1 < 4) ldfrag(ks + 1, 0);
            }
        }
        // the next chunk to be computed must have landed for every wave; the newest one may stay in flight (3-stage)
        if (p.dbg == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else wait_next(AHEAD == 2 && pre);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (++st == NST) st = 0;
    }
    gemm_epilogue<TO, TM, TN, WM, WN>(acc, p, m0, n0, b, g, wm, wn, lr, lk);
}

// bn = 192: 256x192 tile / 64x96 per wave / 2 stages ; bn = 256: 256x256 / 128x64 / 2 stages (both with buffer-descriptor
// DMA: the dispatcher only selects them for whole 64-deep chunks) ; bn = 128: 256x128 / 64x64 / 3-stage ring (flat addresses)
template <typename T, typename TO>
void launch_conv_gemm_dma3(const ConvGemmDev& e, int bn, hipStream_t s) {
    const dim3 grid(e.RC == 2 ? 8 * ((e.RT * e.Tn + 7) / 8) : e.RT * e.Tn, e.G);
    if (bn == 192) MI_LAUNCH((conv_gemm_dma3_kernel<T, TO, 256, 192, 64, 96, 2, true>), T, TO, grid, dim3(512), 0, s, e);
    else if (bn == 256) MI_LAUNCH((conv_gemm_dma3_kernel<T, TO, 256, 256, 128, 64, 2, true>), T, TO, grid, dim3(512), 0, s, e);
    else MI_LAUNCH((conv_gemm_dma3_kernel<T, TO, 256, 128, 64, 64, 3>), T, TO, grid, dim3(512), 0, s, e);
}

template void launch_conv_gemm_dma3<f16, f16>(const ConvGemmDev&, int, hipStream_t);
template void launch_conv_gemm_dma3<f16, float>(const ConvGemmDev&, int, hipStream_t);
template void launch_conv_gemm_dma3<bf16, bf16>(const ConvGemmDev&, int, hipStream_t);
template void launch_conv_gemm_dma3<bf16, float>(const ConvGemmDev&, int, hipStream_t);

}  // namespace mi
