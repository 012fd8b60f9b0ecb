// attention.hip — flash-style attention for the DiT blocks on gfx950 MFMA.
//
// Reference: AttnProcessor.__call__ (F5_TTS/modeling_modified/F5/modules.py:467):
//     softmax(q @ k, dim=-1, dtype=float32) @ v      — no mask, no 1/sqrt(d) (folded into W_q, W_k).
// The reference materialises the (2,16,N,N) logits; here K/V stream through LDS in 64-key stages and
// the softmax is online (running max / sum in fp32), so HBM traffic is q,k,v,o only.
//
// Tiling (wave64, 32x32 MFMA tiles, head_dim = 64):
//   workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries.
//   S^T = K Q^T   (rows = keys, cols = queries): lane l holds query (l&31) and 16 of the 32 keys
//                 [key kappa(r) = (r&3) + 8*(r>>2) + 4*(l>>5)], so a query's softmax statistics are
//                 per-lane: 16 in-register values + ONE cross-lane exchange with lane^32.
//   O^T = V^T P^T (rows = head-dim, cols = queries): the probabilities a lane holds are already in the
//                 MFMA B-operand position (k = key, j = query) — keys are simply consumed in kappa
//                 order, and V is fetched in that same order — no LDS round trip for P, and the
//                 rescale factor exp(m_old - m_new) is again per-lane.
//   fp32 : v_mfma_f32_32x32x2_f32 (exact fp32 products, K = 2 keys per instruction)
//   f16/bf16 : v_mfma_f32_32x32x16 — V is staged transposed ([d][key]) so that the 8 keys a lane feeds
//                 per instruction are two contiguous 8-byte LDS reads.
#include <atomic>
#include "common.h"
#include <functional>
#include <map>
#include <mutex>
#include "mfma.h"
#include "f5_kernels.h"
#include "x3_split.h"
#include <cstdlib>
#include <type_traits>
#include <string>
#include <algorithm>

namespace mi {

// SPLIT2 = false: workgroup = 4 waves x 32 queries, every wave walks all keys.
// SPLIT2 = true : workgroup = 2 x 32 queries; waves (2g, 2g+1) share query group g and take the even / odd 32-key tile of
//                 every 64-key stage, then wave 2g+1 hands its (max, sum, O) to wave 2g through LDS.  Twice as many, half
//                 as long workgroups: the fp32 kernel is MFMA-bound, and one utterance is 9 x 32 = 288 workgroups on 256
//                 CUs, so the CUs that got two workgroups set the makespan (2 units); 576 half-size ones finish in 1.5.
// X3S (fp32 only): the S^T = K Q^T products as exact three-way bf16 splits on the bf16 pipes (gemm_x3.hip has the
//                 arithmetic): 24 v_mfma_f32_32x32x16_bf16 (768 cycles) instead of 32 v_mfma_f32_32x32x2_f32 (2048) per 32x32
//                 tile; Q is split once per workgroup, the K fragments on their way from LDS (176 VALU instructions per
//                 tile, overlapped by the co-resident waves).  P V stays on the native fp32 MFMA (it wants V transposed).
// REFH (f16 only): the rounding points of the reference's fp16-transformer export (F5/fp16/modules.py:467, Export_F5.py:321-326):
//                 q / k carry an extra x0.1 each (folded into the projections before they are rounded to fp16), the scores
//                 leave the q k product as fp16 values, are widened and multiplied by `sscale` (= 100) in fp32, softmax in fp32,
//                 probabilities rounded to fp16 for P V.  Q is therefore NOT pre-multiplied by log2(e) (that would re-round
//                 it): log2(e) rides on `sscale`.
// Workgroup -> (query tile, head).  Consecutive workgroup ids (x fastest) go to consecutive XCDs, so the query tiles of one head
// — which all stream that head's K and V — land on up to eight different L2s and each fetches K / V from the fabric for itself.
// With `xmap` (heads a multiple of 8) the ids are re-read so that XCD x serves heads x, x + 8, ...: a head's query tiles share
// one L2.  Each z plane (key slice) starts at XCD 0 because gridDim.x * gridDim.y is then a multiple of 8.
#define ATTN_XCD_MAP                                                                          \
    int bx_ = (int)blockIdx.x, by_ = (int)blockIdx.y;                                         \
    if (xmap) {                                                                               \
        const int l2 = bx_ + (int)gridDim.x * by_, slot = l2 >> 3;                            \
        by_ = (l2 & 7) + 8 * (slot / (int)gridDim.x);                                         \
        bx_ = slot % (int)gridDim.x;                                                          \
    }

// lane <-> lane ^ 32 exchange of the online softmax (row max, row sum: a query column lives in lanes l and l + 32) on
// v_permlane32_swap instead of ds_bpermute (no LDS-crossbar round trip on the S -> max -> exp chain).  Two copies of the value
// go in; the instruction leaves {own low | low} in one register and {high | own high} in the other, so their max / sum is the
// pair's in every lane, and a + b == b + a bit for bit: both halves still agree and the values are those of the shuffle form.
// Inline asm: on this toolchain the builtin returns the first register twice (tools/ubench/permlane32b.hip); the s_nop covers
// the VALU-write -> permlane-read hazard (two wait states).
__device__ __forceinline__ void xor32_pair(float v, float& a, float& b) {
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float xor32_max(float v) { float a, b; xor32_pair(v, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float xor32_sum(float v) { float a, b; xor32_pair(v, a, b); return a + b; }

template <typename T, bool SPLIT2 = false, bool X3S = false, bool REFH = false>
__global__ __launch_bounds__(256, 3) void attn_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                   const T* __restrict__ v, T* __restrict__ o, int H, int N,
                                                   float* __restrict__ ws, int* __restrict__ cnt, float sscale = 1.f, int xmap = 0) {
    static_assert(!REFH || (sizeof(T) == 2 && !X3S), "REFH is the fp16 form");
    ATTN_XCD_MAP
    using MF = Mfma<T>;
    constexpr int KP = MF::KP;
    constexpr int D = 64, KT = 64;
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int KS = D / (2 * KP);                       // MFMA k-steps over head_dim for S^T
    static_assert(!X3S || sizeof(T) == 4, "the split form is the fp32 kernel");
    constexpr int LDK = D + (sizeof(T) == 4 ? (X3S ? 4 : 1) : 8);         // X3S: 16-byte aligned rows (float4 fragment reads)
    constexpr int LDV = sizeof(T) == 4 ? D : KT + 4;       // fp32: Vs[key][d] ; 16-bit: Vt[d][key]
    constexpr int NV = KT * D / VEC / 256;                 // 16-byte vectors per thread per tile
    __shared__ __attribute__((aligned(16))) T smem[KT * LDK + (sizeof(T) == 4 ? KT * D : D * LDV)];
    T* Ks = smem;
    T* Vs = smem + KT * LDK;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, hi = lane >> 5;
    const int bh = by_;
    const int q0 = SPLIT2 ? bx_ * 64 + (wave >> 1) * 32 : bx_ * 128 + wave * 32;
    const T* qb = q + (long)bh * N * D;
    const T* kb = k + (long)bh * N * D;
    const long vld = sizeof(T) == 4 ? 0 : (long)((N + 7) / 8 * 8);
    const T* vb = v + (sizeof(T) == 4 ? (long)bh * N * D : (long)bh * D * vld);

    // ---- Q fragments (B operand of S^T): Q[q = q0+lr][d = ks*2KP + hi*KP ..] ------------------
    typename MF::Frag qf[X3S ? 1 : KS];
    bf16x8 qf3[X3S ? 4 : 1][3];                            // X3S: Q[q][16 ks + 8 hi .. +8] as three bf16 pieces, pre-scaled by log2(e)
    if constexpr (X3S) {
        const int qr = q0 + lr;
        const bool ok = qr < N;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float4 a = float4{0.f, 0.f, 0.f, 0.f}, b = a;
            if (ok) {
                a = *reinterpret_cast<const float4*>(qb + (long)qr * D + ks * 16 + hi * 8);
                b = *reinterpret_cast<const float4*>(qb + (long)qr * D + ks * 16 + hi * 8 + 4);
            }
            constexpr float L2E = 1.4426950408889634f;
            unsigned u1[4], u2[4], u3[4];
            x3_split_pair(a.x * L2E, a.y * L2E, u1[0], u2[0], u3[0]);
            x3_split_pair(a.z * L2E, a.w * L2E, u1[1], u2[1], u3[1]);
            x3_split_pair(b.x * L2E, b.y * L2E, u1[2], u2[2], u3[2]);
            x3_split_pair(b.z * L2E, b.w * L2E, u1[3], u2[3], u3[3]);
            qf3[ks][0] = __builtin_bit_cast(bf16x8, x3_u4{u1[0], u1[1], u1[2], u1[3]});
            qf3[ks][1] = __builtin_bit_cast(bf16x8, x3_u4{u2[0], u2[1], u2[2], u2[3]});
            qf3[ks][2] = __builtin_bit_cast(bf16x8, x3_u4{u3[0], u3[1], u3[2], u3[3]});
        }
    } else {
        const int qr = q0 + lr;
        const bool ok = qr < N;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            // Q is pre-multiplied by log2(e): the softmax then needs v_exp_f32 (2^x) only, no expf expansion
            if constexpr (KP == 1) {
                qf[ks] = ok ? qb[(long)qr * D + 2 * ks + hi] * 1.4426950408889634f : 0.f;
            } else {
                uint4 raw = make_uint4(0, 0, 0, 0);
                if (ok) raw = *reinterpret_cast<const uint4*>(qb + (long)qr * D + ks * 16 + hi * 8);
                T* e = reinterpret_cast<T*>(&raw);
#pragma unroll
                for (int j = 0; j < 8; ++j) if constexpr (!REFH) e[j] = from_f32<T>(to_f32(e[j]) * 1.4426950408889634f);
                qf[ks] = *reinterpret_cast<const typename MF::Frag*>(&raw);
            }
        }
    }

    uint4 kreg[NV], vreg[NV];
    auto load_regs = [&](int key0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = tid + i * 256;
            const int key = vi / (D / VEC), dv = vi - key * (D / VEC);
            uint4 a = make_uint4(0, 0, 0, 0), b = make_uint4(0, 0, 0, 0);
            if (key0 + key < N) a = *reinterpret_cast<const uint4*>(kb + (long)(key0 + key) * D + dv * VEC);
            if constexpr (sizeof(T) == 4) {
                if (key0 + key < N) b = *reinterpret_cast<const uint4*>(vb + (long)(key0 + key) * D + dv * VEC);
            } else {
                // V arrives transposed from the QKV epilogue: row d = vi / 8, eight consecutive keys per vector
                const int d = vi >> 3, kv = vi & 7;
                if (key0 + kv * 8 < N) b = *reinterpret_cast<const uint4*>(vb + (long)d * vld + key0 + kv * 8);
            }
            kreg[i] = a; vreg[i] = b;
        }
    };
    auto store_lds = [&]() {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = tid + i * 256;
            const int key = vi / (D / VEC), dv = vi - key * (D / VEC);
            if constexpr (sizeof(T) == 4) {
                float* d = reinterpret_cast<float*>(Ks) + key * LDK + dv * VEC;
                d[0] = __uint_as_float(kreg[i].x); d[1] = __uint_as_float(kreg[i].y);
                d[2] = __uint_as_float(kreg[i].z); d[3] = __uint_as_float(kreg[i].w);
                *reinterpret_cast<uint4*>(Vs + key * LDV + dv * VEC) = vreg[i];
            } else {
                *reinterpret_cast<uint4*>(Ks + key * LDK + dv * VEC) = kreg[i];
                const int d = vi >> 3, kv = vi & 7;                                           // Vt[d][key], 8-byte aligned rows
                *reinterpret_cast<uint2*>(Vs + d * LDV + kv * 8) = make_uint2(vreg[i].x, vreg[i].y);
                *reinterpret_cast<uint2*>(Vs + d * LDV + kv * 8 + 4) = make_uint2(vreg[i].z, vreg[i].w);
            }
        }
    };

    f32x16 oacc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    // gridDim.z > 1 (SPLIT2 only): the 64-key stages are cut into gridDim.z contiguous slices, one workgroup each; the
    // slices of a query tile meet in `ws` and the LAST one to arrive merges them in slice order (see the end of the kernel)
    const int nstage_all = (N + KT - 1) / KT;
    const int st0 = SPLIT2 ? (int)((long)blockIdx.z * nstage_all / gridDim.z) : 0;
    const int nstage = SPLIT2 ? (int)((long)(blockIdx.z + 1) * nstage_all / gridDim.z) : nstage_all;
    if (st0 < nstage) {
        load_regs(st0 * KT);
        store_lds();
        __syncthreads();
    }
    for (int st = st0; st < nstage; ++st) {
        if (st + 1 < nstage) load_regs((st + 1) * KT);
        auto tile = [&](int kt) {                                        // one 32-key tile
            const int key0 = st * KT + kt * 32;
            // ---- S^T tile: 32 keys x 32 queries ----------------------------------------------
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            if constexpr (X3S) {
                const float* krow = reinterpret_cast<const float*>(Ks) + (kt * 32 + lr) * LDK + hi * 8;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const float4 a = *reinterpret_cast<const float4*>(krow + ks * 16), b = *reinterpret_cast<const float4*>(krow + ks * 16 + 4);
                    unsigned u1[4], u2[4], u3[4];
                    x3_split_pair(a.x, a.y, u1[0], u2[0], u3[0]);
                    x3_split_pair(a.z, a.w, u1[1], u2[1], u3[1]);
                    x3_split_pair(b.x, b.y, u1[2], u2[2], u3[2]);
                    x3_split_pair(b.z, b.w, u1[3], u2[3], u3[3]);
                    const bf16x8 k1 = __builtin_bit_cast(bf16x8, x3_u4{u1[0], u1[1], u1[2], u1[3]});
                    const bf16x8 k2 = __builtin_bit_cast(bf16x8, x3_u4{u2[0], u2[1], u2[2], u2[3]});
                    const bf16x8 k3 = __builtin_bit_cast(bf16x8, x3_u4{u3[0], u3[1], u3[2], u3[3]});
                    // six partial products, small terms first
                    sacc = Mfma<bf16>::mma(k1, qf3[ks][2], sacc);
                    sacc = Mfma<bf16>::mma(k2, qf3[ks][1], sacc);
                    sacc = Mfma<bf16>::mma(k3, qf3[ks][0], sacc);
                    sacc = Mfma<bf16>::mma(k1, qf3[ks][1], sacc);
                    sacc = Mfma<bf16>::mma(k2, qf3[ks][0], sacc);
                    sacc = Mfma<bf16>::mma(k1, qf3[ks][0], sacc);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const typename MF::Frag a =
                        *reinterpret_cast<const typename MF::Frag*>(Ks + (kt * 32 + lr) * LDK + ks * 2 * KP + hi * KP);
                    sacc = MF::mma(a, qf[ks], sacc);
                }
            }
            if constexpr (REFH) {
                // torch.matmul(query, key) is an fp16 tensor; `.float() * 100.0` follows (fp16/modules.py:467)
                const float c = sscale * 1.4426950408889634f;
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = to_f32(from_f32<T>(sacc[r])) * c;
            }
            // ---- online softmax (per lane = per query) -----------------------------------------
            // only the tile that straddles N needs the key >= N select (2 VALU issues per score, a third of the
            // softmax's VALU time when done on every tile): wave-uniform branch around it
            if (key0 + 32 > N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= N) sacc[r] = -INFINITY;
                }
            }
            float mloc = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, sacc[r]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32));         // (v_permlane32_swap measured 1 % SLOWER in this kernel, 3.5 % faster in attn_x3f_kernel)
            // lazy rescale: keep the old reference max while it is within 2^8 of the new one (P <= 256, exact in
            // fp32 accumulation); rescale O and l only when some query of the wave needs it (wave-uniform branch)
            float alpha = 1.f;
            if (!__all(mloc - m_run <= 8.0f)) {
                const float m_new = fmaxf(m_run, mloc);
                alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
#pragma unroll
                for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; }
            }
            // score - max and the row sum as packed fp32 pairs (v_pk_add_f32: two per VALU issue)
            typedef float f2 __attribute__((ext_vector_type(2)));
            float p[16];
            f2 ls2 = f2{0.f, 0.f};
            const f2 m2 = f2{m_run, m_run};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f2 d = f2{sacc[r], sacc[r + 1]} - m2;
                const f2 e = f2{__builtin_amdgcn_exp2f(d.x), __builtin_amdgcn_exp2f(d.y)};
                p[r] = e.x; p[r + 1] = e.y;
                ls2 += e;
            }
            float lsum = ls2.x + ls2.y;
            lsum += __shfl_xor(lsum, 32);
            l_run = l_run * alpha + lsum;
            // ---- O^T += V^T P^T ------------------------------------------------------------------
            if constexpr (KP == 1) {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int kk = kt * 32 + (s & 3) + 8 * (s >> 2) + 4 * hi;
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
                        oacc[dt] = MF::mma(reinterpret_cast<const float*>(Vs)[kk * LDV + dt * 32 + lr], p[s], oacc[dt]);
                }
            } else {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    T pb[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) pb[e] = from_f32<T>(p[8 * s2 + e]);
                    const typename MF::Frag bfrag = *reinterpret_cast<const typename MF::Frag*>(pb);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const T* row = Vs + (dt * 32 + lr) * LDV + kt * 32 + 16 * s2 + 4 * hi;
                        uint2 a2[2];
                        a2[0] = *reinterpret_cast<const uint2*>(row);
                        a2[1] = *reinterpret_cast<const uint2*>(row + 8);
                        oacc[dt] = MF::mma(*reinterpret_cast<const typename MF::Frag*>(a2), bfrag, oacc[dt]);
                    }
                }
            }
        };
        if constexpr (SPLIT2) {
            if (st * KT + (wave & 1) * 32 < N) tile(wave & 1);           // this wave's half of the stage
        } else {
#pragma unroll
            for (int kt = 0; kt < KT / 32; ++kt) {
                const int key0 = st * KT + kt * 32;
                if (key0 < N) tile(kt);                                  // wave-uniform
            }
        }
        __syncthreads();
        if (st + 1 < nstage) {
            store_lds();
            __syncthreads();
        }
    }

    bool owner = true;                                              // this wave holds a finished 32-query result
    if constexpr (SPLIT2) {
        // ---- merge the two key halves: m = max(m0, m1) ; l = l0 2^(m0-m) + l1 2^(m1-m) ; O likewise ----------------------
        __syncthreads();                                            // every wave is done with the K / V stage
        float* comb = reinterpret_cast<float*>(smem);               // [2 groups][32 accumulator registers][64 lanes]
        float* stats = comb + 2 * 32 * 64;                          // [2 groups][64 lanes][m, l]
        static_assert(sizeof(smem) >= (2 * 32 * 64 + 2 * 64 * 2 + 4) * sizeof(float), "merge buffer fits the stage");
        const int grp = wave >> 1;
        owner = !(wave & 1);
        if (!owner) {
            stats[(grp * 64 + lane) * 2] = m_run; stats[(grp * 64 + lane) * 2 + 1] = l_run;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) comb[((grp * 2 + dt) * 16 + r) * 64 + lane] = oacc[dt][r];
        }
        __syncthreads();
        if (owner) {
            const float m1 = stats[(grp * 64 + lane) * 2], l1 = stats[(grp * 64 + lane) * 2 + 1];
            const float m = fmaxf(m_run, m1);
            const float s0 = m_run == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m_run - m);
            const float s1 = m1 == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m1 - m);
            l_run = l_run * s0 + l1 * s1;
            m_run = m;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] = oacc[dt][r] * s0 + comb[((grp * 2 + dt) * 16 + r) * 64 + lane] * s1;
        }
        // ---- key slices (gridDim.z > 1).  One utterance in fp32 is 18 x 32 = 576 of these workgroups on 768 slots (three per
        // CU): 2.25 per CU, so the CUs that got three set the makespan and a quarter of the chip idles.  With Z slices the
        // work comes in pieces of 1 / Z (Z = 4: exactly 9 per CU).  Every slice publishes (m, l, O) of its keys with
        // write-through stores, drains them, and takes a ticket (relaxed agent-scope fetch-add: the gemm_sk.hip hand-off);
        // the workgroup that draws the last ticket adds the slices IN SLICE ORDER (its own from registers), so the result
        // does not depend on which one that is, resets the ticket counter (hipGraph replays find it at zero) and stores.
        // Nobody waits for anybody: there is no residency requirement. ---------------------------------------------------
        const int Z = (int)gridDim.z;
        if (Z > 1) {
            const int z = (int)blockIdx.z;
            const int unit = (int)(by_ * gridDim.x + bx_);
            constexpr int SLOT = 2 * 32 * 64 + 2 * 64 * 2;           // floats per (unit, slice)
            // write-through (sc1) stores / sc1 loads through a buffer descriptor, as in gemm_sk.hip: no L2-wide write-back fence
            typedef unsigned int u4 __attribute__((ext_vector_type(4)));
            typedef unsigned int u2 __attribute__((ext_vector_type(2)));
            __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)(ws + (long)unit * Z * SLOT), 0, Z * SLOT * 4, 0x00020000);
            if (owner) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        u4 val;
                        val.x = __float_as_uint(oacc[dt][4 * g4]); val.y = __float_as_uint(oacc[dt][4 * g4 + 1]);
                        val.z = __float_as_uint(oacc[dt][4 * g4 + 2]); val.w = __float_as_uint(oacc[dt][4 * g4 + 3]);
                        __builtin_amdgcn_raw_buffer_store_b128(val, rsw, (z * SLOT + (((grp * 2 + dt) * 4 + g4) * 64 + lane) * 4) * 4, 0, 16);
                    }
                u2 st2; st2.x = __float_as_uint(m_run); st2.y = __float_as_uint(l_run);
                __builtin_amdgcn_raw_buffer_store_b64(st2, rsw, (z * SLOT + 2 * 32 * 64 + (grp * 64 + lane) * 2) * 4, 0, 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int* ticket = reinterpret_cast<int*>(stats + 2 * 64 * 2);
            if (tid == 0) *ticket = __hip_atomic_fetch_add(cnt + unit, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (*ticket != Z - 1) return;                            // not the last slice of this query tile
            if (tid == 0) __hip_atomic_store(cnt + unit, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!owner) return;
            float mz[4], lz[4], M = -INFINITY;
#pragma unroll
            for (int zz = 0; zz < 4; ++zz) {
                mz[zz] = -INFINITY; lz[zz] = 0.f;
                if (zz < Z) {
                    if (zz == z) { mz[zz] = m_run; lz[zz] = l_run; }
                    else {
                        const u2 st2 = __builtin_amdgcn_raw_buffer_load_b64(rsw, (zz * SLOT + 2 * 32 * 64 + (grp * 64 + lane) * 2) * 4, 0, 16);
                        mz[zz] = __uint_as_float(st2.x); lz[zz] = __uint_as_float(st2.y);
                    }
                    M = fmaxf(M, mz[zz]);
                }
            }
            f32x16 osum[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) { osum[0][r] = 0.f; osum[1][r] = 0.f; }
            float lsum = 0.f;
#pragma unroll
            for (int zz = 0; zz < 4; ++zz) {
                if (zz >= Z) break;
                const float sc = mz[zz] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mz[zz] - M);
                lsum += lz[zz] * sc;
                if (zz == z) {
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) osum[dt][r] += oacc[dt][r] * sc;
                } else {
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const u4 val = __builtin_amdgcn_raw_buffer_load_b128(rsw, (zz * SLOT + (((grp * 2 + dt) * 4 + g4) * 64 + lane) * 4) * 4, 0, 16);
                            osum[dt][4 * g4] += __uint_as_float(val.x) * sc; osum[dt][4 * g4 + 1] += __uint_as_float(val.y) * sc;
                            osum[dt][4 * g4 + 2] += __uint_as_float(val.z) * sc; osum[dt][4 * g4 + 3] += __uint_as_float(val.w) * sc;
                        }
                }
            }
            oacc[0] = osum[0]; oacc[1] = osum[1];
            l_run = lsum;
        }
    }
    if (!owner) return;
    // ---- normalise + store o[b][n][h*64 + d] ---------------------------------------------------------
    const int qr = q0 + lr;
    if (qr < N) {
        const float inv = 1.0f / l_run;
        const int b = bh / H, h = bh - b * H;
        T* ob = o + ((long)b * N + qr) * H * D + h * D;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                T vals[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) vals[e] = from_f32<T>(oacc[dt][g * 4 + e] * inv);
                T* dst = ob + dt * 32 + 8 * g + 4 * hi;
                if constexpr (sizeof(T) == 4) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(vals);
                else *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(vals);
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// attn_x3f_kernel — fp32 attention with BOTH products as exact three-way bf16 splits (the arithmetic of gemm_x3.hip).
// Layout and tiling of the 16-bit kernel above (V arrives transposed, [bh][64][v_ld]; K planes [key][64 + 8], V planes
// [d][64 + 4] in LDS), with fp32 in HBM: K and V^T are split ONCE per 64-key stage on their way into LDS (three bf16 planes
// each, shared by the four waves — 22 VALU instructions per float4 instead of a split per wave and tile), Q once per
// workgroup, the probabilities per tile.  Per 32x32 tile: 24 + 24 v_mfma_f32_32x32x16_bf16 (1536 matrix-core cycles)
// against 32 + 32 v_mfma_f32_32x32x2_f32 (4096).  SPLIT2 / key slices / merge: as attn_kernel.
// ---------------------------------------------------------------------------------------------------------------------
// KVP: K and V^T arrive ALREADY split, as the three bf16 planes the QKV epilogue wrote (k = [bh][3][kld][64] bf16,
// v = [bh][3][64][vld] bf16, kld = vld = N rounded up to the 64-key stage): a stage is 12 16-byte loads per thread copied to
// LDS as they are, instead of 8 float4 loads + 176 VALU of splitting per thread and stage — work that every one of the
// N / 64 query tiles of a head repeated on the same K / V values (with the key-split 64-query workgroups a wave multiplies
// ONE 32x32 tile per stage: the stage split was as many VALU cycles as the tile's softmax and P split together).
// NP = 2 (only with KVP): K / V^T arrive as fp16 {hi, lo} pairs with the low part UNSCALED (x3_split.h: x2u_split_pair — the
// operands of attention are of order one, their residuals stay above fp16's subnormal floor where it matters) and Q / P are
// split the same way: three partial products per block, small terms first, into the one accumulator — half the MFMAs of the
// three-plane form, 4 instead of 6 bytes per K / V element, no second accumulator set.
#if defined(MI355TTS_ATTN_TRACE)      // tuning only: s_memtime stamps at the phase boundaries of one wave (tools/dbg/attn_trace.sh)
#define TSTAMP(v) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define TACC(k, t1, t0) tr_[k] += (t1) - (t0)
#else
#define TSTAMP(v) do { } while (0)
#define TACC(k, t1, t0) do { } while (0)
#endif
template <bool SPLIT2, bool KVP = false, int NP = 3>
__global__ __launch_bounds__(256, NP == 2 ? 3 : 2) void attn_x3f_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, float* __restrict__ o, int H, int N,
                                                       float* __restrict__ ws, int* __restrict__ cnt,
                                                       unsigned char* __restrict__ o_planes, int o_np, int xmap = 0,
                                                       int cut1 = 0, int cut2 = 0, int cut3 = 0) {
    ATTN_XCD_MAP
    // o_planes != null: the output leaves as gemm_x3p.hip panel planes of the [B * N][H * 64] matrix (the A operand of the O
    // projection), split here (o_np = 3 bf16 planes | 2 fp16 planes), instead of fp32 rows in o
    static_assert(NP == 3 || (NP == 2 && KVP), "attn_x3f: the two-plane form reads pre-split K / V");
    using MF = std::conditional_t<NP == 3, Mfma<bf16>, Mfma<f16>>;
    using Frag = typename MF::Frag;
    constexpr int D = 64, KT = 64;
    constexpr int LDK = D + 8, LDV = KT + 4;                // bf16 elements per plane row
    constexpr int KPL = KT * LDK, VPL = D * LDV;            // elements per plane
    __shared__ __attribute__((aligned(16))) bf16 smem[NP * KPL + NP * VPL];            // 16-bit storage (bf16 or fp16 bit patterns)
    static_assert(sizeof(smem) >= (2 * 32 * 64 + 2 * 64 * 2 + 4) * sizeof(float), "merge buffer fits the stage");
    bf16* Ks = smem;
    bf16* Vs = smem + NP * KPL;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, hi = lane >> 5;
    const int bh = by_;
    const int q0 = SPLIT2 ? bx_ * 64 + (wave >> 1) * 32 : bx_ * 128 + wave * 32;
    const float* qb = q + (long)bh * N * D;
    const float* kb = k + (long)bh * N * D;
    const long vld = KVP ? (long)((N + 63) / 64 * 64) : (long)((N + 7) / 8 * 8);
    const float* vb = v + (long)bh * D * vld;
    const bf16* kpb = reinterpret_cast<const bf16*>(k) + (long)bh * NP * vld * D;     // KVP: planes [NP][kld][64], kld = vld
    const bf16* vpb = reinterpret_cast<const bf16*>(v) + (long)bh * NP * D * vld;     // KVP: planes [NP][64][vld]

    auto split4 = [&](const float4 a, uint2& w1, uint2& w2, uint2& w3) __attribute__((always_inline)) {
        x3_split_pair(a.x, a.y, w1.x, w2.x, w3.x);
        x3_split_pair(a.z, a.w, w1.y, w2.y, w3.y);
    };
    auto split8 = [&](const float4 a, const float4 b, Frag& f1, Frag& f2, Frag& f3) __attribute__((always_inline)) {
        uint2 a1, a2, a3, b1, b2, b3;
        split4(a, a1, a2, a3); split4(b, b1, b2, b3);
        f1 = __builtin_bit_cast(Frag, x3_u4{a1.x, a1.y, b1.x, b1.y});
        f2 = __builtin_bit_cast(Frag, x3_u4{a2.x, a2.y, b2.x, b2.y});
        f3 = __builtin_bit_cast(Frag, x3_u4{a3.x, a3.y, b3.x, b3.y});
    };

    auto split8h = [&](const float4 a, const float4 b, Frag& f1, Frag& f2) __attribute__((always_inline)) {
        unsigned h[4], l[4];
        x2u_split_pair(a.x, a.y, h[0], l[0]); x2u_split_pair(a.z, a.w, h[1], l[1]);
        x2u_split_pair(b.x, b.y, h[2], l[2]); x2u_split_pair(b.z, b.w, h[3], l[3]);
        f1 = __builtin_bit_cast(Frag, x3_u4{h[0], h[1], h[2], h[3]});
        f2 = __builtin_bit_cast(Frag, x3_u4{l[0], l[1], l[2], l[3]});
    };
    // the probabilities: in [0, 1] by construction, no range clamp
    auto split8p = [&](const float4 a, const float4 b, Frag& f1, Frag& f2) __attribute__((always_inline)) {
        unsigned h[4], l[4];
        x2u_split_pair_raw_mix(a.x, a.y, h[0], l[0]); x2u_split_pair_raw_mix(a.z, a.w, h[1], l[1]);
        x2u_split_pair_raw_mix(b.x, b.y, h[2], l[2]); x2u_split_pair_raw_mix(b.z, b.w, h[3], l[3]);
        f1 = __builtin_bit_cast(Frag, x3_u4{h[0], h[1], h[2], h[3]});
        f2 = __builtin_bit_cast(Frag, x3_u4{l[0], l[1], l[2], l[3]});
    };

    // ---- Q: the NP pieces of log2(e) * Q[q0 + lr][16 ks + 8 hi .. +8] ------------------------------------------------
    Frag qf3[4][NP];
    {
        const int qr = q0 + lr;
        const bool ok = qr < N;
        constexpr float L2E = 1.4426950408889634f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float4 a = float4{0.f, 0.f, 0.f, 0.f}, b = a;
            if (ok) {
                a = *reinterpret_cast<const float4*>(qb + (long)qr * D + ks * 16 + hi * 8);
                b = *reinterpret_cast<const float4*>(qb + (long)qr * D + ks * 16 + hi * 8 + 4);
            }
            a.x *= L2E; a.y *= L2E; a.z *= L2E; a.w *= L2E; b.x *= L2E; b.y *= L2E; b.z *= L2E; b.w *= L2E;
            if constexpr (NP == 3) split8(a, b, qf3[ks][0], qf3[ks][1], qf3[ks][2]); else split8h(a, b, qf3[ks][0], qf3[ks][1]);
        }
    }

    // ---- stage loads: 64 keys x 64 d of K (rows = keys) and of V^T (rows = d), four float4 per thread each ----------------
    float4 kreg[KVP ? 1 : 4], vreg[KVP ? 1 : 4];
    constexpr int NU = 2 * NP;                              // KVP: 16-byte units per thread and operand per stage
    x3_u4 kpl[KVP ? NU : 1], vpl[KVP ? NU : 1];
    auto load_regs = [&](int key0) {
        if constexpr (KVP) {
            // unit u = tid + 256 i: plane u / 512 ; K: key (u % 512) / 8, 8 d's ; V^T: d (u % 512) / 8, 8 keys.  Rows past N are
            // inside the padded planes (K: any finite-or-not value, masked below; V: zero since allocation)
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const int u = tid + i * 256, pl = u >> 9, r = (u & 511) >> 3, c = u & 7;
                kpl[i] = *reinterpret_cast<const x3_u4*>(kpb + ((long)pl * vld + key0 + r) * D + c * 8);
                vpl[i] = *reinterpret_cast<const x3_u4*>(vpb + ((long)pl * D + r) * vld + key0 + c * 8);
            }
            if (key0 + KT > N) {
                // last stage: the V^T values of keys >= N meet probabilities that are exactly zero, but 0 x NaN is NaN and
                // the pad columns hold whatever the buffer held before (another layout, another mode): clear them here
#pragma unroll
                for (int i = 0; i < NU; ++i) {
                    const int keep = N - (key0 + ((tid + i * 256) & 7) * 8);          // valid keys among this unit's eight
                    if (keep < 8) {
                        unsigned w[4] = {vpl[i].x, vpl[i].y, vpl[i].z, vpl[i].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] &= (2 * e < keep ? 0x0000ffffu : 0u) | (2 * e + 1 < keep ? 0xffff0000u : 0u);
                        vpl[i] = x3_u4{w[0], w[1], w[2], w[3]};
                    }
                }
            }
        } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int vi = tid + i * 256;
            const int key = vi >> 4, dv = vi & 15;              // K: row key, floats 4 dv .. 4 dv + 3
            const int d = vi >> 4, kv = vi & 15;                // V^T: row d, keys key0 + 4 kv .. + 3
            float4 a = float4{0.f, 0.f, 0.f, 0.f}, b = a;
            if (key0 + key < N) a = *reinterpret_cast<const float4*>(kb + (long)(key0 + key) * D + dv * 4);
            if (key0 + kv * 4 < N) b = *reinterpret_cast<const float4*>(vb + (long)d * vld + key0 + kv * 4);
            kreg[i] = a; vreg[i] = b;
        }
        }
    };
    auto store_lds = [&]() {
        if constexpr (KVP) {
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const int u = tid + i * 256, pl = u >> 9, r = (u & 511) >> 3, c = u & 7;
                *reinterpret_cast<x3_u4*>(Ks + pl * KPL + r * LDK + c * 8) = kpl[i];                 // 144-byte rows: 16-byte aligned
                uint2* vd = reinterpret_cast<uint2*>(Vs + pl * VPL + r * LDV + c * 8);                // 136-byte rows: 8-byte aligned
                vd[0] = uint2{vpl[i].x, vpl[i].y}; vd[1] = uint2{vpl[i].z, vpl[i].w};
            }
        } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int vi = tid + i * 256;
            const int r = vi >> 4, c = vi & 15;
            uint2 w1, w2, w3;
            split4(kreg[i], w1, w2, w3);
            *reinterpret_cast<uint2*>(Ks + r * LDK + c * 4) = w1;
            *reinterpret_cast<uint2*>(Ks + KPL + r * LDK + c * 4) = w2;
            *reinterpret_cast<uint2*>(Ks + 2 * KPL + r * LDK + c * 4) = w3;
            split4(vreg[i], w1, w2, w3);
            *reinterpret_cast<uint2*>(Vs + r * LDV + c * 4) = w1;
            *reinterpret_cast<uint2*>(Vs + VPL + r * LDV + c * 4) = w2;
            *reinterpret_cast<uint2*>(Vs + 2 * VPL + r * LDV + c * 4) = w3;
        }
        }
    };

    f32x16 oacc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
#if defined(MI355TTS_ATTN_TRACE)
    unsigned long long tr_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ta_ = 0, tb_ = 0, tc_ = 0, td_ = 0, te_ = 0, tk_ = 0;
    TSTAMP(tk_);
#endif

    const int nstage_all = (N + KT - 1) / KT;
    // key slices (gridDim.z > 1): see the merge below.  cut1 > 0 (round 6): UNEVEN slices [0, cut1), [cut1, cut2), ... chosen by the
    // launcher so that the z-major dispatch order is longest-piece-first on the 3 x CUs slots (attn_pick_slices)
    int st0 = (int)((long)blockIdx.z * nstage_all / gridDim.z);
    int nstage = (int)((long)(blockIdx.z + 1) * nstage_all / gridDim.z);
    if (cut1 > 0) {
        const int zz = (int)blockIdx.z, ZZ = (int)gridDim.z;
        st0 = zz == 0 ? 0 : zz == 1 ? cut1 : zz == 2 ? cut2 : cut3;
        nstage = zz + 1 == ZZ ? nstage_all : zz == 0 ? cut1 : zz == 1 ? cut2 : cut3;
    }
    if (st0 < nstage) {
        load_regs(st0 * KT);
        store_lds();
        __syncthreads();
    }
    constexpr int TA[6] = {0, 1, 2, 0, 1, 0}, TB[6] = {2, 1, 0, 1, 0, 0};      // piece pairs, small terms first
    for (int st = st0; st < nstage; ++st) {
        if (st + 1 < nstage) load_regs((st + 1) * KT);
        auto tile = [&](int kt) {
            const int key0 = st * KT + kt * 32;
            TSTAMP(ta_);
            // ---- S^T tile: 32 keys x 32 queries, six partial products per k16 step ---------------------------------------
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                Frag kf[NP];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    kf[pl] = *reinterpret_cast<const Frag*>(Ks + pl * KPL + (kt * 32 + lr) * LDK + ks * 16 + hi * 8);
                if constexpr (NP == 3) {
#pragma unroll
                    for (int t = 0; t < 6; ++t) sacc = MF::mma(kf[TA[t]], qf3[ks][TB[t]], sacc);
                } else {                          // small terms first, one accumulator (unscaled low parts)
                    sacc = MF::mma(kf[1], qf3[ks][0], sacc);
                    sacc = MF::mma(kf[0], qf3[ks][1], sacc);
                    sacc = MF::mma(kf[0], qf3[ks][0], sacc);
                }
            }
            if (key0 + 32 > N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= N) sacc[r] = -INFINITY;
                }
            }
            float mloc = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, sacc[r]);
            mloc = xor32_max(mloc);
#if defined(MI355TTS_ATTN_TRACE)
            { const float keep_ = mloc; asm volatile("" :: "v"(keep_)); }
#endif
            TSTAMP(tb_); TACC(0, tb_, ta_);            // K fragment reads + score MFMAs + tile maximum
            float alpha = 1.f;
            if (!__all(mloc - m_run <= 8.0f)) {
                const float m_new = fmaxf(m_run, mloc);
                alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
#pragma unroll
                for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; }
            }
            typedef float f2 __attribute__((ext_vector_type(2)));
            float p[16];
            f2 ls2 = f2{0.f, 0.f};
            const f2 m2 = f2{m_run, m_run};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f2 dd = f2{sacc[r], sacc[r + 1]} - m2;
                const f2 e = f2{__builtin_amdgcn_exp2f(dd.x), __builtin_amdgcn_exp2f(dd.y)};
                p[r] = e.x; p[r + 1] = e.y;
                ls2 += e;
            }
            float lsum = ls2.x + ls2.y;
            lsum = xor32_sum(lsum);
            l_run = l_run * alpha + lsum;
#if defined(MI355TTS_ATTN_TRACE)
            { const float keep_ = l_run + p[0] + p[15]; asm volatile("" :: "v"(keep_)); }
#endif
            TSTAMP(tc_); TACC(1, tc_, tb_);            // exponentials + row sum
            // ---- O^T += V^T P^T: the probabilities a lane holds are the B operand (keys in accumulator-row order) ------------
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                Frag pf[NP];
                if constexpr (NP == 3)
                    split8(float4{p[8 * s2], p[8 * s2 + 1], p[8 * s2 + 2], p[8 * s2 + 3]},
                           float4{p[8 * s2 + 4], p[8 * s2 + 5], p[8 * s2 + 6], p[8 * s2 + 7]}, pf[0], pf[1], pf[2]);
                else
                    split8p(float4{p[8 * s2], p[8 * s2 + 1], p[8 * s2 + 2], p[8 * s2 + 3]},
                            float4{p[8 * s2 + 4], p[8 * s2 + 5], p[8 * s2 + 6], p[8 * s2 + 7]}, pf[0], pf[1]);
                // the two head-dim halves alternate: an MFMA that waits for the previous one's accumulator costs 52 cycles instead
                // of 32 (tools/ubench/mfma_valu_coexec.hip); the order inside each accumulator is unchanged (bit-identical).
                // (The same for the two score tiles of a stage in the 128-query form was measured too: 27 spilled registers and
                // the stage loads issued one tile later cost more than the chains gain, 56.9 -> 70.8 us.)
                Frag vf[2][NP];
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) {
                        const bf16* row = Vs + pl * VPL + (dt * 32 + lr) * LDV + kt * 32 + 16 * s2 + 4 * hi;
                        uint2 a2[2];
                        a2[0] = *reinterpret_cast<const uint2*>(row);
                        a2[1] = *reinterpret_cast<const uint2*>(row + 8);
                        vf[dt][pl] = __builtin_bit_cast(Frag, x3_u4{a2[0].x, a2[0].y, a2[1].x, a2[1].y});
                    }
                if constexpr (NP == 3) {
#pragma unroll
                    for (int t = 0; t < 6; ++t)
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) oacc[dt] = MF::mma(vf[dt][TA[t]], pf[TB[t]], oacc[dt]);
                } else {
                    oacc[0] = MF::mma(vf[0][1], pf[0], oacc[0]);
                    oacc[1] = MF::mma(vf[1][1], pf[0], oacc[1]);
                    oacc[0] = MF::mma(vf[0][0], pf[1], oacc[0]);
                    oacc[1] = MF::mma(vf[1][0], pf[1], oacc[1]);
                    oacc[0] = MF::mma(vf[0][0], pf[0], oacc[0]);
                    oacc[1] = MF::mma(vf[1][0], pf[0], oacc[1]);
                }
            }
            TSTAMP(td_); TACC(2, td_, tc_);            // P split + V fragment reads + P V MFMAs (issued)
        };
        if constexpr (SPLIT2) {
            if (st * KT + (wave & 1) * 32 < N) tile(wave & 1);
        } else {
#pragma unroll
            for (int kt = 0; kt < KT / 32; ++kt) {
                const int key0 = st * KT + kt * 32;
                if (key0 < N) tile(kt);
            }
        }
        TSTAMP(ta_);
        __syncthreads();
        TSTAMP(tb_); TACC(3, tb_, ta_);                // barrier 1 (includes the tail of the P V MFMAs)
        if (st + 1 < nstage) {
            store_lds();
            TSTAMP(tc_); TACC(4, tc_, tb_);            // wait for the stage loads + LDS stores
            __syncthreads();
            TSTAMP(te_); TACC(5, te_, tc_);            // barrier 2
        }
    }
#if defined(MI355TTS_ATTN_TRACE)
    TSTAMP(te_);
    if (blockIdx.x == 3 && blockIdx.y == 5 && (threadIdx.x & 63) == 0)
        printf("ATTN_TRACE z %d wave %d stages %d total %llu | S+max %llu exp %llu PV %llu bar1 %llu store %llu bar2 %llu\n", (int)blockIdx.z, wave, nstage - st0,
               te_ - tk_, tr_[0], tr_[1], tr_[2], tr_[3], tr_[4], tr_[5]);
#endif
    bool owner = true;                                              // this wave holds a finished 32-query result
    constexpr int NG = SPLIT2 ? 2 : 4;                              // 32-query groups of the workgroup
    int grp = wave;
    float* comb = reinterpret_cast<float*>(smem);                   // [2 groups][32 accumulator registers][64 lanes] (SPLIT2 pair merge)
    float* stats = comb + 2 * 32 * 64;                              // [2 groups][64 lanes][m, l]
    if constexpr (SPLIT2) {
        // ---- merge the two key halves: m = max(m0, m1) ; l = l0 2^(m0-m) + l1 2^(m1-m) ; O likewise ----------------------
        __syncthreads();                                            // every wave is done with the K / V stage
        grp = wave >> 1;
        owner = !(wave & 1);
        if (!owner) {
            stats[(grp * 64 + lane) * 2] = m_run; stats[(grp * 64 + lane) * 2 + 1] = l_run;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) comb[((grp * 2 + dt) * 16 + r) * 64 + lane] = oacc[dt][r];
        }
        __syncthreads();
        if (owner) {
            const float m1 = stats[(grp * 64 + lane) * 2], l1 = stats[(grp * 64 + lane) * 2 + 1];
            const float m = fmaxf(m_run, m1);
            const float s0 = m_run == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m_run - m);
            const float s1 = m1 == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m1 - m);
            l_run = l_run * s0 + l1 * s1;
            m_run = m;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] = oacc[dt][r] * s0 + comb[((grp * 2 + dt) * 16 + r) * 64 + lane] * s1;
        }
    }
    {
        // ---- key slices (gridDim.z > 1).  One utterance in fp32 is 18 x 32 = 576 of these workgroups on 768 slots (three per
        // CU): 2.25 per CU, so the CUs that got three set the makespan and a quarter of the chip idles.  With Z slices the
        // work comes in pieces of 1 / Z (Z = 4: exactly 9 per CU).  Every slice publishes (m, l, O) of its keys with
        // write-through stores, drains them, and takes a ticket (relaxed agent-scope fetch-add: the gemm_sk.hip hand-off);
        // the workgroup that draws the last ticket adds the slices IN SLICE ORDER (its own from registers), so the result
        // does not depend on which one that is, resets the ticket counter (hipGraph replays find it at zero) and stores.
        // Nobody waits for anybody: there is no residency requirement. ---------------------------------------------------
        const int Z = (int)gridDim.z;
        if (Z > 1) {
            const int z = (int)blockIdx.z;
            const int unit = (int)(by_ * gridDim.x + bx_);
            constexpr int SLOT = NG * (32 * 64 + 64 * 2);            // floats per (unit, slice)
            // write-through (sc1) stores / sc1 loads through a buffer descriptor, as in gemm_sk.hip: no L2-wide write-back fence
            typedef unsigned int u4 __attribute__((ext_vector_type(4)));
            typedef unsigned int u2 __attribute__((ext_vector_type(2)));
            __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)(ws + (long)unit * Z * SLOT), 0, Z * SLOT * 4, 0x00020000);
            if (owner) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        u4 val;
                        val.x = __float_as_uint(oacc[dt][4 * g4]); val.y = __float_as_uint(oacc[dt][4 * g4 + 1]);
                        val.z = __float_as_uint(oacc[dt][4 * g4 + 2]); val.w = __float_as_uint(oacc[dt][4 * g4 + 3]);
                        __builtin_amdgcn_raw_buffer_store_b128(val, rsw, (z * SLOT + (((grp * 2 + dt) * 4 + g4) * 64 + lane) * 4) * 4, 0, 16);
                    }
                u2 st2; st2.x = __float_as_uint(m_run); st2.y = __float_as_uint(l_run);
                __builtin_amdgcn_raw_buffer_store_b64(st2, rsw, (z * SLOT + NG * 32 * 64 + (grp * 64 + lane) * 2) * 4, 0, 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                         // (also: every wave is done with the K / V stage the ticket word lies in)
            int* ticket = reinterpret_cast<int*>(stats + 2 * 64 * 2);
            if (tid == 0) *ticket = __hip_atomic_fetch_add(cnt + unit, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (*ticket != Z - 1) return;                            // not the last slice of this query tile
            if (tid == 0) __hip_atomic_store(cnt + unit, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!owner) return;
            float mz[4], lz[4], M = -INFINITY;
#pragma unroll
            for (int zz = 0; zz < 4; ++zz) {
                mz[zz] = -INFINITY; lz[zz] = 0.f;
                if (zz < Z) {
                    if (zz == z) { mz[zz] = m_run; lz[zz] = l_run; }
                    else {
                        const u2 st2 = __builtin_amdgcn_raw_buffer_load_b64(rsw, (zz * SLOT + NG * 32 * 64 + (grp * 64 + lane) * 2) * 4, 0, 16);
                        mz[zz] = __uint_as_float(st2.x); lz[zz] = __uint_as_float(st2.y);
                    }
                    M = fmaxf(M, mz[zz]);
                }
            }
            f32x16 osum[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) { osum[0][r] = 0.f; osum[1][r] = 0.f; }
            float lsum = 0.f;
#pragma unroll
            for (int zz = 0; zz < 4; ++zz) {
                if (zz >= Z) break;
                const float sc = mz[zz] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mz[zz] - M);
                lsum += lz[zz] * sc;
                if (zz == z) {
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) osum[dt][r] += oacc[dt][r] * sc;
                } else {
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const u4 val = __builtin_amdgcn_raw_buffer_load_b128(rsw, (zz * SLOT + (((grp * 2 + dt) * 4 + g4) * 64 + lane) * 4) * 4, 0, 16);
                            osum[dt][4 * g4] += __uint_as_float(val.x) * sc; osum[dt][4 * g4 + 1] += __uint_as_float(val.y) * sc;
                            osum[dt][4 * g4 + 2] += __uint_as_float(val.z) * sc; osum[dt][4 * g4 + 3] += __uint_as_float(val.w) * sc;
                        }
                }
            }
            oacc[0] = osum[0]; oacc[1] = osum[1];
            l_run = lsum;
        }
    }
    if (!owner) return;
    // ---- normalise + store o[b][n][h*64 + d] ---------------------------------------------------------
    const int qr = q0 + lr;
    if (qr < N) {
        const float inv = 1.0f / l_run;
        const int b = bh / H, h = bh - b * H;
        float* ob = o + ((long)b * N + qr) * H * D + h * D;
        const long row = (long)b * N + qr;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float vals[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) vals[e] = (oacc[dt][g * 4 + e] * inv);
                if (o_planes) {
                    // columns h*64 + dt*32 + 8g + 4hi .. +3: half (8 bytes) of the k-slot s8 = h*8 + dt*4 + g of every plane
                    uint2 w1, w2, w3;
                    unsigned char* dst = o_planes + x3p_slot_offset(row, h * 8 + dt * 4 + g, (H * D) >> 5, o_np) + 8 * hi;
                    if (o_np == 3) {
                        x3_split_pair(vals[0], vals[1], w1.x, w2.x, w3.x);
                        x3_split_pair(vals[2], vals[3], w1.y, w2.y, w3.y);
                        *reinterpret_cast<uint2*>(dst + 2 * X3P_PLANE) = w3;
                    } else {
                        x2_split_pair(vals[0], vals[1], w1.x, w2.x);
                        x2_split_pair(vals[2], vals[3], w1.y, w2.y);
                    }
                    *reinterpret_cast<uint2*>(dst) = w1;
                    *reinterpret_cast<uint2*>(dst + X3P_PLANE) = w2;
                } else {
                    float* dst = ob + dt * 32 + 8 * g + 4 * hi;
                    *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(vals);
                }
            }
    }
}

// key slices of the SPLIT2 form: at most g_attn_zmax for fp32, g_attn_z16 for 16-bit operands (1 = off: no gain measured)
// fp32 attention: 0 = native fp32 MFMA ; 1 = q.k as exact bf16 splits (attn_kernel X3S) ; 2 = both products (attn_x3f_kernel;
// V then arrives transposed like in the 16-bit engines: attention_v_ld() tells the QKV epilogue)
static std::atomic<int> g_attn_x3 = 2;
static std::atomic<int> g_attn_np = 2;                                // format of the pre-split K / V^T (and of Q / P inside the kernel): 2 fp16 pairs | 3 bf16 planes
static std::atomic<int> g_attn_split = 2;                             // small grids: 1 = 64-query workgroups with the keys split between wave pairs (+ key slices) ; 2 = fp32 pairs kernel: 128-query workgroups + key slices
static std::atomic<int> g_attn_zmax = 4, g_attn_z16 = 1, g_attn_zforce = 0;      // zforce (tests): exactly that many slices, even empty ones
static std::atomic<int> g_attn_xmap = 1;                              // XCD-aware (query tile, head) map of the workgroup ids (A/B: attn_xcd_map; -1 % per launch, bit-neutral)
static std::atomic<int> g_attn_lpt = 1;                               // fp32 128-query kernel: uneven key slices, longest first (A/B: attn_lpt)
static std::atomic<int> g_attn_kvp = 1;                               // fp32, both products split: K / V^T pre-split by the QKV epilogue (A/B: attn_kv_planes)
// The MI355TTS_ATTN_* environment overrides are read ONCE, before the first use of any of the globals above by ANY of the
// three entry points: F5::dit_eval asks attention_v_ld() for the V layout of the QKV epilogue before the first
// launch_attention() of the process, and a lazy read inside launch_attention() made that first block write V transposed
// for a kernel that then read it untransposed.  mi_set_option() applied later overrides the environment.
static inline int opt_attn_x3() { const int o = arith_tls().attn_x3; return o >= 0 ? o : (int)g_attn_x3; }      // the engine's arithmetic first (ArithScope, common.h)
static inline int opt_attn_np() { const int o = arith_tls().attn_np; return (o == 2 || o == 3) ? o : (int)g_attn_np; }
static void attn_env_once() {
    static std::once_flag once;
    std::call_once(once, [] {
        const char* e = std::getenv("MI355TTS_ATTN_NO_SPLIT"); if (e && e[0] == '1') g_attn_split = 0;
        if (const char* z = std::getenv("MI355TTS_ATTN_Z")) g_attn_zmax = std::max(1, std::min(4, std::atoi(z)));
        if (const char* z = std::getenv("MI355TTS_ATTN_X3")) g_attn_x3 = std::max(0, std::min(2, std::atoi(z)));
        if (const char* z = std::getenv("MI355TTS_ATTN_Z16")) g_attn_z16 = std::max(1, std::min(4, std::atoi(z)));
        if (const char* z = std::getenv("MI355TTS_ATTN_KVP")) g_attn_kvp = std::atoi(z) != 0;
        if (const char* z = std::getenv("MI355TTS_ATTN_LPT")) g_attn_lpt = std::atoi(z) != 0;
        if (const char* z = std::getenv("MI355TTS_ATTN_PLANES")) g_attn_np = std::atoi(z) == 3 ? 3 : 2;
    });
}
long attention_v_ld(int N, int dtype) {
    attn_env_once();
    return (dtype == MI_F32 && opt_attn_x3() != 2) ? 0 : (long)((N + 7) / 8 * 8);
}
bool attn_set_option(const char* key, long v) {
    attn_env_once();
    const std::string k(key);
    if (k == "attn_f32_x3") g_attn_x3 = (int)std::max(0L, std::min(2L, v));
    else if (k == "attn_split") g_attn_split = (int)std::max(0L, std::min(2L, v));
    else if (k == "attn_xcd_map") g_attn_xmap = v != 0;
    else if (k == "attn_kv_planes") g_attn_kvp = v != 0;
    else if (k == "attn_lpt") g_attn_lpt = v != 0;
    else if (k == "attn_f32_planes") { if (v != 2 && v != 3) return false; g_attn_np = (int)v; }
    else if (k == "attn_z_force") g_attn_zforce = (int)std::max(0L, std::min(4L, v));
    else return false;
    return true;
}

bool attention_takes_kv_planes(int N, int BH, int dtype) {
    attn_env_once();
    (void)N; (void)BH;
    return dtype == MI_F32 && opt_attn_x3() == 2 && g_attn_kvp != 0;
}

int attention_kv_planes_format() { attn_env_once(); return opt_attn_np(); }

bool attention_can_write_planes(int N, int BH, int dtype) {
    attn_env_once();
    (void)N; (void)BH;
    return dtype == MI_F32 && opt_attn_x3() == 2;
}

// Key slices of the 128-query fp32 kernel, sized for the dispatch order (round 6).  The launch is (query tiles x heads) units x Z
// slices on 3 x CUs workgroup slots, dispatched z-major.  Equal slices make 288 x 3 = 864 workgroups for one utterance: a full
// round of 768 and an eighth of a second one, which costs as much as half a round (the second round's workgroups run alone on their
// SIMDs: ~18 of the launch's 50 us).  With slices of 7, 7 and 4 stages the long pieces start first and the short ones fill the slots
// they free: list scheduling, makespan 8 stage times instead of 12.  The search simulates that for every non-increasing split of the
// stages into 1 .. zmax slices (identical pieces handled in bulk: a few map operations per split) and keeps the best.
struct AttnSlices { int Z = 1, cut[3] = {0, 0, 0}; };
static AttnSlices attn_pick_slices(long units, int S, int slots, int zmax) {
    AttnSlices best;
    double best_t = 1e30;
    auto makespan = [&](const int* len, int Z) -> double {
        std::map<double, long> free_at;                          // time -> slots that become free then
        free_at[0.0] = slots;
        double end = 0.0;
        for (int z = 0; z < Z; ++z) {
            const double cost = len[z] + (Z > 1 ? 0.6 : 0.3);      // stages + prologue / publish / merge (in stage times)
            long left = units;
            while (left > 0) {
                auto it = free_at.begin();
                const long take = std::min(left, it->second);
                const double t = it->first + cost;
                it->second -= take;
                if (it->second == 0) free_at.erase(it);
                free_at[t] += take;
                end = std::max(end, t);
                left -= take;
            }
        }
        return end;
    };
    int len[4];
    for (int Z = 1; Z <= zmax; ++Z) {
        if (S < 2 * Z && Z > 1) break;
        // non-increasing len[0] >= ... >= len[Z-1] >= 1 with sum S
        std::function<void(int, int, int)> rec = [&](int i, int left, int cap) {
            if (i == Z - 1) {
                if (left < 1 || left > cap) return;
                len[i] = left;
                const double t = makespan(len, Z);
                if (t < best_t - 1e-9) {
                    best_t = t; best.Z = Z;
                    int acc = 0;
                    for (int c = 0; c < 3; ++c) { acc += c < Z ? len[c] : 0; best.cut[c] = c + 1 < Z ? acc : S; }
                }
                return;
            }
            for (int a = std::min(cap, left - (Z - 1 - i)); a >= (left + (Z - i) - 1) / (Z - i); --a) { len[i] = a; rec(i + 1, left - a, a); }
        };
        rec(0, S, S);
    }
    return best;
}

void launch_attention(const void* q, const void* k, const void* v, void* o, int BH, int H, int N, int dtype, hipStream_t s,
                      float* ws, long ws_floats, int* cnt, long cnt_n, void* o_planes, int kv_planes, int o_np, float ref_fp16_scale) {
    MI_REQUIRE(ref_fp16_scale == 0.f || (dtype == MI_F16 && ref_fp16_scale > 0.f), "attention: the reference-fp16 score form needs f16 operands");
    MI_REQUIRE(o_np == 2 || o_np == 3, "attention: 2 or 3 output planes");
    MI_REQUIRE(!o_planes || attention_can_write_planes(N, BH, dtype), "attention: panel-plane output needs the fp32 split kernel");
    MI_REQUIRE(!kv_planes || attention_can_write_planes(N, BH, dtype), "attention: pre-split K / V need the fp32 split kernel");
    MI_REQUIRE(kv_planes == 0 || kv_planes == 1 || kv_planes == 2 || kv_planes == 3, "attention: kv_planes is 0, 2 or 3 (1 = 3)");
    MI_REQUIRE(BH % H == 0 && N > 0, "attention: bad shape");
    const double esz = (double)dtype_size(dtype);
    ProfScope ps(FAM_ATTN, s, 4.0 * BH * N * 64.0 * esz, 4.0 * BH * (double)N * N * 64.0);
#define ATTN_LAUNCH(TT, SP, ...)                                                \
    do {                                                                        \
        prof_set_kernel("attn_kernel<T, " #SP ">", type_label<TT>());           \
        hipLaunchKernelGGL((attn_kernel<TT, SP>), __VA_ARGS__);                 \
    } while (0)
    attn_env_once();
    const int xm = (g_attn_xmap != 0 && BH % 8 == 0) ? 1 : 0;
    const int zmax = g_attn_zmax;
    // key slices for the SPLIT2 form (see attn_kernel): makespan(Z) = ceil(units * Z / CUs) / Z in units of one unsliced
    // workgroup, + 6 % per extra slice for the prologue and the merge (measured, fp32, one utterance = 576 units:
    // Z = 1 / 2 / 3 / 4 -> 135 / 122 / 121 / 126 us in round 2; 62.1 / 58.8 / 60.8 / 60.7 us with the fp16-pair kernel of
    // round 3, whose slices are shorter against the same merge: + 9 % for fp32)
    auto pick_z = [&](int zlimit16, bool wide = false) -> int {
        const long units = (long)(wide ? (N + 127) / 128 : (N + 63) / 64) * BH;
        const long slot = wide ? 4 * (32 * 64 + 64 * 2) : 2 * 32 * 64 + 2 * 64 * 2;
        const int nstage = (N + 63) / 64;
        int dev = 0, cus = 256;
        MI_HIP(hipGetDevice(&dev));
        {
            static int cu_count[16] = {0};
            if (!cu_count[dev & 15]) { hipDeviceProp_t pr; MI_HIP(hipGetDeviceProperties(&pr, dev)); cu_count[dev & 15] = pr.multiProcessorCount; }
            cus = cu_count[dev & 15];
        }
        if (g_attn_zforce > 0)
            return (ws && cnt && units * g_attn_zforce * slot <= ws_floats && units <= cnt_n) ? (int)g_attn_zforce : 1;
        int Z = 1;
        double best = 1e30;
        const int zm = dtype == MI_F32 ? zmax : std::min(zmax, zlimit16);
        for (int z = 1; z <= zm; ++z) {
            if (z > 1 && (!ws || !cnt || units * z * slot > ws_floats || units > cnt_n || nstage < 2 * z)) break;
            const double cost = (double)((units * z + cus - 1) / cus) / z * (1.0 + (dtype == MI_F32 ? 0.09 : 0.06) * (z - 1));
            if (cost < best - 1e-9) { best = cost; Z = z; }
        }
        return Z;
    };
    const int z16 = g_attn_z16;
    if (dtype == MI_F32) {
        // few 128-query workgroups (one or two utterances): halve them along the keys, see attn_kernel — or (attn_split = 2, the
        // default for the pre-split fp16-pair kernel, round 4) keep the 128-query workgroups, whose four waves share every K / V
        // stage and multiply two tiles per barrier pair, and cut only the key range into slices: 288 x 3 workgroups for one
        // utterance, 56.3 against 60.0 us per launch on the same box (step 177.8 -> 174.3 ms)
        if (g_attn_split && (long)((N + 127) / 128) * BH < 1024 && N >= 64) {
            // ... and cut the key range into Z slices when that evens out the workgroups per CU
            const bool wide = g_attn_split == 2 && opt_attn_x3() == 2 && kv_planes == 2;
            const int Z = pick_z(1, wide);
            if (wide) {
                // uneven slices, longest first (attn_pick_slices): cached per (N, BH)
                AttnSlices sl;
                sl.Z = Z;
                if (g_attn_lpt && g_attn_zforce == 0 && Z >= 1 && ws && cnt) {
                    static std::mutex mu;
                    static std::map<std::pair<int, int>, AttnSlices> cache;
                    std::lock_guard<std::mutex> lk(mu);
                    auto key = std::make_pair(N, BH);
                    auto it = cache.find(key);
                    if (it == cache.end()) {
                        int dev = 0, cus = 256;
                        MI_HIP(hipGetDevice(&dev));
                        hipDeviceProp_t pr; MI_HIP(hipGetDeviceProperties(&pr, dev)); cus = pr.multiProcessorCount;
                        const long units = (long)((N + 127) / 128) * BH;
                        const long slot = 4 * (32 * 64 + 64 * 2);
                        int zm = std::min((int)g_attn_zmax, 4);
                        while (zm > 1 && (units * zm * slot > ws_floats || units > cnt_n)) --zm;
                        it = cache.emplace(key, attn_pick_slices(units, (N + 63) / 64, 3 * cus, zm)).first;
                    }
                    sl = it->second;
                    if (const char* e = std::getenv("MI355TTS_ATTN_CUTS")) {          // experiments: "7,14" = slices of stages [0,7) [7,14) [14,S)
                        const int S = (N + 63) / 64;
                        int c[3] = {S, S, S}, n = std::sscanf(e, "%d,%d,%d", &c[0], &c[1], &c[2]);
                        const long units = (long)((N + 127) / 128) * BH;
                        if (n >= 1 && c[0] > 0 && c[0] < S && units * (n + 1) * 4 * (32 * 64 + 64 * 2) <= ws_floats && units <= cnt_n) {
                            sl.Z = n + 1; sl.cut[0] = c[0]; sl.cut[1] = n >= 2 ? c[1] : S; sl.cut[2] = n >= 3 ? c[2] : S;
                        }
                    }
                }
                prof_set_kernel("attn_x3f_kernel<false, pre-split K V, fp16 pairs> + key slices", "", "");
                hipLaunchKernelGGL((attn_x3f_kernel<false, true, 2>), dim3((N + 127) / 128, BH, sl.Z), dim3(256), 0, s, (const float*)q, (const float*)k, (const float*)v, (float*)o, H, N, ws, cnt, (unsigned char*)o_planes, o_np, xm,
                                   sl.Z > 1 ? sl.cut[0] : 0, sl.cut[1], sl.cut[2]);
            } else if (opt_attn_x3() == 2) {
                if (kv_planes == 2) {
                    prof_set_kernel("attn_x3f_kernel<true, pre-split K V, fp16 pairs>", "", "");
                    hipLaunchKernelGGL((attn_x3f_kernel<true, true, 2>), dim3((N + 63) / 64, BH, Z), dim3(256), 0, s, (const float*)q, (const float*)k, (const float*)v, (float*)o, H, N, ws, cnt, (unsigned char*)o_planes, o_np, xm);
                } else if (kv_planes) {
                    prof_set_kernel("attn_x3f_kernel<true, pre-split K V>", "", "");
                    hipLaunchKernelGGL((attn_x3f_kernel<true, true>), dim3((N + 63) / 64, BH, Z), dim3(256), 0, s, (const float*)q, (const float*)k, (const float*)v, (float*)o, H, N, ws, cnt, (unsigned char*)o_planes, o_np);
                } else {
                prof_set_kernel("attn_x3f_kernel<true>", "", "");
                hipLaunchKernelGGL((attn_x3f_kernel<true>), dim3((N + 63) / 64, BH, Z), dim3(256), 0, s, (const float*)q, (const float*)k, (const float*)v, (float*)o, H, N, ws, cnt, (unsigned char*)o_planes, o_np);
                }
            } else if (opt_attn_x3() == 1) {
                prof_set_kernel("attn_kernel<float, true, x3>", "", "");
                hipLaunchKernelGGL((attn_kernel<float, true, true>), dim3((N + 63) / 64, BH, Z), dim3(256), 0, s, (const float*)q, (const float*)k, (const float*)v, (float*)o, H, N, ws, cnt);
            } else
                ATTN_LAUNCH(float, true, dim3((N + 63) / 64, BH, Z), dim3(256), 0, s, (const float*)q, (const float*)k, (const float*)v, (float*)o, H, N, ws, cnt);
        } else if (opt_attn_x3() == 2) {
            if (kv_planes == 2) {
                prof_set_kernel("attn_x3f_kernel<false, pre-split K V, fp16 pairs>", "", "");
                hipLaunchKernelGGL((attn_x3f_kernel<false, true, 2>), dim3((N + 127) / 128, BH), dim3(256), 0, s, (const float*)q, (const float*)k, (const float*)v, (float*)o, H, N, ws, cnt, (unsigned char*)o_planes, o_np, xm);
            } else if (kv_planes) {
                prof_set_kernel("attn_x3f_kernel<false, pre-split K V>", "", "");
                hipLaunchKernelGGL((attn_x3f_kernel<false, true>), dim3((N + 127) / 128, BH), dim3(256), 0, s, (const float*)q, (const float*)k, (const float*)v, (float*)o, H, N, ws, cnt, (unsigned char*)o_planes, o_np);
            } else {
            prof_set_kernel("attn_x3f_kernel<false>", "", "");
            hipLaunchKernelGGL((attn_x3f_kernel<false>), dim3((N + 127) / 128, BH), dim3(256), 0, s, (const float*)q, (const float*)k, (const float*)v, (float*)o, H, N, ws, cnt, (unsigned char*)o_planes, o_np);
            }
        } else if (opt_attn_x3() == 1) {
            prof_set_kernel("attn_kernel<float, false, x3>", "", "");
            hipLaunchKernelGGL((attn_kernel<float, false, true>), dim3((N + 127) / 128, BH), dim3(256), 0, s, (const float*)q, (const float*)k, (const float*)v, (float*)o, H, N, ws, cnt);
        } else
            ATTN_LAUNCH(float, false, dim3((N + 127) / 128, BH), dim3(256), 0, s, (const float*)q, (const float*)k, (const float*)v, (float*)o, H, N, ws, cnt);
    } else {
        // 16-bit: the same split below 512 workgroups (one utterance: attention 24.3 -> 21.1 ms per step; at two utterances,
        // 576 workgroups, the 128-query form is already balanced and shares each K / V stage among more waves)
        const bool sp = g_attn_split && (long)((N + 127) / 128) * BH < 512 && N >= 64;
        const dim3 grid(sp ? (N + 63) / 64 : (N + 127) / 128, BH, sp ? pick_z(z16) : 1);
        if (dtype == MI_F16 && ref_fp16_scale != 0.f) {
            prof_set_kernel(sp ? "attn_kernel<T, true, reference-fp16 scores>" : "attn_kernel<T, false, reference-fp16 scores>", type_label<f16>());
            if (sp) hipLaunchKernelGGL((attn_kernel<f16, true, false, true>), grid, dim3(256), 0, s, (const f16*)q, (const f16*)k, (const f16*)v, (f16*)o, H, N, ws, cnt, ref_fp16_scale);
            else hipLaunchKernelGGL((attn_kernel<f16, false, false, true>), grid, dim3(256), 0, s, (const f16*)q, (const f16*)k, (const f16*)v, (f16*)o, H, N, ws, cnt, ref_fp16_scale);
        } else if (dtype == MI_F16) {
            if (sp) ATTN_LAUNCH(f16, true, grid, dim3(256), 0, s, (const f16*)q, (const f16*)k, (const f16*)v, (f16*)o, H, N, ws, cnt, 1.f, xm);
            else ATTN_LAUNCH(f16, false, grid, dim3(256), 0, s, (const f16*)q, (const f16*)k, (const f16*)v, (f16*)o, H, N, ws, cnt, 1.f, xm);
        } else {
            if (sp) ATTN_LAUNCH(bf16, true, grid, dim3(256), 0, s, (const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)o, H, N, ws, cnt, 1.f, xm);
            else ATTN_LAUNCH(bf16, false, grid, dim3(256), 0, s, (const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)o, H, N, ws, cnt, 1.f, xm);
        }
    }
#undef ATTN_LAUNCH
    MI_HIP(hipGetLastError());
}

}  // namespace mi
