// Weight + bias gradients of one BSARecBlock at hidden = 64 without LDS staging ("direct" split-K products).
//
//   dW[o][i] = sum_t G[t][o] . Act[t][i]      (six products: query/key/value/dense, dense_1, dense_2;
//   db[o]    = sum_t G[t][o]                   src/model/_modules.py:29-32,89-96 through autograd)
//
// Both operands are token-major, i.e. k-major for this product, which is exactly the MFMA operand shape: for
// v_mfma_f32_32x32x2_f32 lane l supplies A[k = l>>5][m = l&31] and B[k = l>>5][n = l&31], so a wave reads its
// fragments STRAIGHT from global memory -- per instruction two 128-byte row segments -- and needs neither LDS nor
// barriers.  Every wave is an independent unit: one (problem, 32x64 or 64x32 output tile, split-K slice); it keeps
// STAGES k-blocks of 8 token rows in flight in registers, accumulates in two 32x32 accumulator tiles and writes its
// slab; the step's flat reduction sums the slabs (deterministic).  The tiled LDS kernel this replaces at the fused
// shape spent ~1 us per 32-deep k-step on store -> barrier -> fragment-read latency with 16 MFMAs per wave in it.
#pragma once
#include "fused_layer.h"

#define DW_MAX_PROB 6
#define DW_MAX_UNITS 32
#define DW_STAGES 4

struct DwProblem {
    const float* A; const float* B;     // gradient rows [K][M] (lda), activation rows [K][N] (ldb)
    long lda, ldb;
    int M, N, K, kchunk;                // K tokens in total, kchunk per slice (multiple of 8)
    float* slab;                        // [nsplit][M][N]
    float* bslab;                       // [nsplit][M]
    int gelu;                           // erf-GELU on the activation operand while loading (dW2 = dT2^T . gelu(u))
};

struct DwUnit { short prob, m0, n0, wide_m; };      // wide_m: 64(m) x 32(n) tile, else 32(m) x 64(n)

struct DwP {
    DwProblem P[DW_MAX_PROB];
    DwUnit U[DW_MAX_UNITS];
    int units_per_split, nsplit;
};

// one k-block (8 token rows) of operand registers: lane half h holds rows 4h .. 4h+3
struct DwStage { float a[2][4]; float b[2][4]; };

// Issue only: no predicate, no branch (a predicated load becomes a branch, after which the compiler can no longer
// count the loads in flight and falls back to s_waitcnt vmcnt(0), i.e. no prefetch).  Reads may run up to
// DW_STAGES + 1 k-blocks past the slice: inside the buffer that is the next slice's rows, past the buffer it is
// the workspace's guard pad (bsarec_hip.hip, carve); such rows are zeroed when consumed.
template <bool WIDE_M>
__device__ __forceinline__ void dw_issue(DwStage& st, const float* __restrict__ pa, const float* __restrict__ pb, long lda,
                                         long ldb) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        st.a[0][s] = gld(pa + s * lda);
        if (WIDE_M) st.a[1][s] = gld(pa + s * lda + 32);
        st.b[0][s] = gld(pb + s * ldb);
        if (!WIDE_M) st.b[1][s] = gld(pb + s * ldb + 32);
    }
}

template <bool WIDE_M, bool GELU, bool MASK>
__device__ __forceinline__ void dw_loop(const float* __restrict__ pa, const float* __restrict__ pb, long lda, long ldb, int nkb,
                                        int crow, int kend, f32x16& acc0, f32x16& acc1, float& bs0, float& bs1) {
    DwStage st[DW_STAGES];
#pragma unroll
    for (int u = 0; u < DW_STAGES; ++u) {
        dw_issue<WIDE_M>(st[u], pa, pb, lda, ldb);
        pa += 8 * lda; pb += 8 * ldb;
    }
    for (int kb = 0; kb < nkb; kb += DW_STAGES) {       // nkb is a multiple of DW_STAGES (host: kchunk % 32 == 0)
#pragma unroll
        for (int u = 0; u < DW_STAGES; ++u) {
            DwStage& cur = st[u];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (MASK) {
                    const bool ok = crow + s < kend;
                    cur.a[0][s] = ok ? cur.a[0][s] : 0.f; cur.b[0][s] = ok ? cur.b[0][s] : 0.f;
                    if (WIDE_M) cur.a[1][s] = ok ? cur.a[1][s] : 0.f; else cur.b[1][s] = ok ? cur.b[1][s] : 0.f;
                }
                if (GELU) {
                    cur.b[0][s] = gelu_f(cur.b[0][s]);
                    if (!WIDE_M) cur.b[1][s] = gelu_f(cur.b[1][s]);
                }
                bs0 += cur.a[0][s];
                if (WIDE_M) bs1 += cur.a[1][s];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[0][s], cur.b[0][s], acc0, 0, 0, 0);
                if (WIDE_M) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[1][s], cur.b[0][s], acc1, 0, 0, 0);
                else acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[0][s], cur.b[1][s], acc1, 0, 0, 0);
            }
            crow += 8;
            // Refill the stage just consumed, DW_STAGES k-blocks ahead.  Pinned between scheduling barriers: left
            // free, the scheduler sinks these loads next to their next-iteration uses and waits vmcnt(0) before every
            // MFMA pair, i.e. no prefetch at all.
            __builtin_amdgcn_sched_barrier(0);
            dw_issue<WIDE_M>(st[u], pa, pb, lda, ldb);
            pa += 8 * lda; pb += 8 * ldb;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <bool WIDE_M, bool GELU>
__device__ __forceinline__ void dw_unit(const DwProblem& Q, int m0, int n0, int split) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    const int kbeg = split * Q.kchunk, kend = min(Q.K, kbeg + Q.kchunk);
    const int nkb = kend > kbeg ? ((kend - kbeg + 8 * DW_STAGES - 1) / (8 * DW_STAGES)) * DW_STAGES : 0;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float bs0 = 0.f, bs1 = 0.f;
    if (nkb > 0) {                                       // wave-uniform; empty slices (pruned top block) only write zeros
        const long lda = Q.lda, ldb = Q.ldb;
        const float* pa = Q.A + (long)(kbeg + 4 * half) * lda + m0 + l31;
        const float* pb = Q.B + (long)(kbeg + 4 * half) * ldb + n0 + l31;
        const int crow = kbeg + 4 * half;
        if (kbeg + 8 * nkb <= kend) dw_loop<WIDE_M, GELU, false>(pa, pb, lda, ldb, nkb, crow, kend, acc0, acc1, bs0, bs1);
        else dw_loop<WIDE_M, GELU, true>(pa, pb, lda, ldb, nkb, crow, kend, acc0, acc1, bs0, bs1);
    }
    // slabs: accumulator register r of lane (l31, half) is C[m = rho(r) + 4 half][n = l31] of its 32x32 tile
    float* C = Q.slab + (long)split * Q.M * Q.N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mr = m0 + rho(r) + 4 * half;
        gst(C + (long)mr * Q.N + n0 + l31, acc0[r]);
        if (WIDE_M) gst(C + (long)(mr + 32) * Q.N + n0 + l31, acc1[r]);
        else gst(C + (long)mr * Q.N + n0 + 32 + l31, acc1[r]);
    }
    if (n0 == 0) {        // bias gradient: column sums of the gradient operand (the two lane halves hold different rows)
        bs0 = xor32_sum(bs0);
        if (WIDE_M) bs1 = xor32_sum(bs1);
        float* bo = Q.bslab + (long)split * Q.M + m0 + l31;
        if (half == 0) { gst(bo, bs0); if (WIDE_M) gst(bo + 32, bs1); }
    }
}

// The host builds the unit table so that the GELU problem (dense_2: M = 64, N = 256) is tiled 64(m) x 32(n) -- every
// u element is loaded and activated by exactly one wave -- and everything else 32(m) x 64(n).
__global__ void __launch_bounds__(256)
dw_direct_kernel(const DwP G) {
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // XCD-aware mapping: workgroups are dealt round-robin to the 8 XCDs (id % 8), each with its own L2.  All units of
    // one split-K slice read the same token rows (X feeds the q/k/v tiles, hmix and dT2 eight tiles each), so a slice
    // is kept on ONE XCD so that the re-reads hit that L2 (measured at C1: 22.9 us vs 23.7 us for the plain order;
    // FETCH_SIZE stays near the 52 MB of unique operand bytes either way -- the kernel is latency-, not byte-bound).
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int wgs_per_split = (G.units_per_split + 3) >> 2;
    const int split = xcd + 8 * (j / wgs_per_split);
    const int unit_in_split = (j % wgs_per_split) * 4 + wv;
    if (split >= G.nsplit || unit_in_split >= G.units_per_split) return;
    const DwUnit u = G.U[unit_in_split];
    const DwProblem& Q = G.P[u.prob];
    if (u.wide_m) dw_unit<true, true>(Q, u.m0, u.n0, split);
    else dw_unit<false, false>(Q, u.m0, u.n0, split);
}
