// Weight + bias gradients of one BSARecBlock at hidden = 64 without LDS staging ("direct" split-K products).
//
//   dW[o][i] = sum_t G[t][o] . Act[t][i]      (six products: query/key/value/dense, dense_1, dense_2;
//   db[o]    = sum_t G[t][o]                   src/model/_modules.py:29-32,89-96 through autograd)
//
// Both operands are token-major, i.e. k-major for this product, which is the MFMA operand shape itself: for
// v_mfma_f32_32x32x2_f32 lane l supplies A[k = l>>5][m(l&31)] and B[k = l>>5][n(l&31)], and the 32 m (n) values of a
// tile may be ANY 32 columns.  A wave owns a 64 x 64 output tile as 2 x 2 interleaved 32 x 32 tiles -- tile (i, j)
// holds rows m0 + 2a + i, columns n0 + 2b + j -- so that one 8-byte load per lane per token row feeds both m tiles
// (resp. both n tiles): fragments come STRAIGHT from global memory, 512 contiguous bytes per wave instruction, with
// neither LDS staging nor barriers in the k loop.  Per k-block of 8 token rows a wave issues 8 such loads for 16
// MFMAs and keeps DW_STAGES k-blocks in flight in registers.
//
// Work decomposition: unit = (problem, 64 x 64 tile); the token axis is cut into `nslab` slab slices, each slice
// again into 4 quarters taken by the 4 waves of a workgroup, which add their accumulators through LDS and write ONE
// slab; the step's flat reduction sums the nslab slabs (deterministic, fixed order).  The LDS-tiled kernel this
// replaces at the fused shape spent ~1 us per 32-deep k-step on store -> barrier -> fragment-read latency.
#pragma once
#include "fused_layer.h"

#define DW_MAX_PROB 12        // two blocks' worth: the pruned top block's problems ride in the next block's launch
#define DW_MAX_UNITS 24
#define DW_STAGES 5

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gld2(const float* p) { return *reinterpret_cast<const AS_GLOBAL f32x2*>((const AS_GLOBAL float*)p); }
__device__ __forceinline__ void gst2(float* p, f32x2 v) { *reinterpret_cast<AS_GLOBAL f32x2*>((AS_GLOBAL float*)p) = v; }

struct DwProblem {
    const float* A; const float* B;     // gradient rows [K][M] (lda), activation rows [K][N] (ldb); M, N multiples of 64
    long lda, ldb;
    int M, N, K, kchunk;                // K tokens in total, kchunk per slab slice (multiple of 32)
    int nslab;                          // slab slices of THIS problem (<= DwP::nslab; the tiny top-block problems use fewer)
    float* slab;                        // [nslab][M][N]
    float* bslab;                       // [nslab][M]
    int gelu;                           // erf-GELU on the activation operand while loading (dW2 = dT2^T . gelu(u))
    int bf16;                           // both operands are bf16 tensors (cfg.storage = 1): lda / ldb still count elements; the
                                        // products run on the bf16 matrix cores (dw_loop_m; with `gelu`: widened, fp32 MFMA)
};

struct DwUnit { short prob, m0, n0, pad; };

struct DwP {
    DwProblem P[DW_MAX_PROB];
    DwUnit U[DW_MAX_UNITS];
    int nunits, nslab;
    // Units [0, nsmall) belong to the tiny problems of the pruned top block (K = B or B*h rows, `small_slabs` slab
    // slices each): they take the LAST nsmall * small_slabs workgroups of the grid on their own -- no empty
    // workgroups; the big problems' 8 * ceil(nslab / 8) * (nunits - nsmall) workgroups come first so that all of them
    // are resident from the start (384 of the 512 slots at C1), the small ones fill the rest and finish early.
    int nsmall, small_slabs;
};

// one k-block (8 token rows) of operand registers: lane half h holds rows 4h .. 4h+3, two columns of each operand
// (bf16 operands: the two columns are one dword)
template <bool BF> struct DwStage;
template <> struct DwStage<false> { f32x2 a[4]; f32x2 b[4]; };
template <> struct DwStage<true> { unsigned a[4]; unsigned b[4]; };

// Issue only: no predicate, no branch (a predicated load becomes a branch, after which the compiler can no longer
// count the loads in flight and falls back to s_waitcnt vmcnt(0), i.e. no prefetch).  Buffer loads: the per-lane byte
// offset (voff) is loop-invariant and the row offset is a scalar, so the k loop has no vector address arithmetic at
// all.  Reads may run up to DW_STAGES + 1 k-blocks past the wave's rows: inside the operand those are the next
// slice's rows, past it the buffer descriptor's range check returns 0 (num_records = the operand's bytes); rows past
// the slice are zeroed when consumed.
typedef int i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 bld2(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
template <bool BF>
__device__ __forceinline__ void dw_issue(DwStage<BF>& st, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, int voa, int vob,
                                         int soa, int sob, int rowa, int rowb) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if constexpr (BF) {
            st.a[s] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(ra, voa, soa + s * rowa, 0);
            st.b[s] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rb, vob, sob + s * rowb, 0);
        } else {
            st.a[s] = bld2(ra, voa, soa + s * rowa);
            st.b[s] = bld2(rb, vob, sob + s * rowb);
        }
    }
}

template <bool GELU, bool MASK, bool BF, int STAGES>
__device__ __forceinline__ void dw_loop(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, int voa, int vob, int soa, int sob,
                                        int rowa, int rowb, int nkb, int crow, int kend, f32x16 (&acc)[2][2], f32x2& bs) {
    DwStage<BF> st[STAGES];
#pragma unroll
    for (int u = 0; u < STAGES; ++u) {
        dw_issue(st[u], ra, rb, voa, vob, soa, sob, rowa, rowb);
        soa += 8 * rowa; sob += 8 * rowb;
    }
    for (int kb = 0; kb < nkb; kb += STAGES) {          // the last group may be short (wave-uniform test per stage)
#pragma unroll
        for (int u = 0; u < STAGES; ++u) {
            if (kb + u >= nkb) break;
            DwStage<BF>& cur = st[u];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f32x2 a, b;
                if constexpr (BF) { a = f32x2{bf_lo(cur.a[s]), bf_hi(cur.a[s])}; b = f32x2{bf_lo(cur.b[s]), bf_hi(cur.b[s])}; }
                else { a = cur.a[s]; b = cur.b[s]; }
                if (MASK) {
                    const bool ok = crow + s < kend;
                    a.x = ok ? a.x : 0.f; a.y = ok ? a.y : 0.f; b.x = ok ? b.x : 0.f; b.y = ok ? b.y : 0.f;
                }
                if (GELU) { b.x = gelu_f(b.x); b.y = gelu_f(b.y); }
                bs += a;
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.y, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.x, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[1][1], 0, 0, 0);
            }
            crow += 8;
            // Refill the stage just consumed, DW_STAGES k-blocks ahead.  Pinned between scheduling barriers: left
            // free, the scheduler sinks these loads next to their next-iteration uses and waits vmcnt(0) before every
            // MFMA group, i.e. no prefetch at all.
            __builtin_amdgcn_sched_barrier(0);
            dw_issue(st[u], ra, rb, voa, vob, soa, sob, rowa, rowb);
            soa += 8 * rowa; sob += 8 * rowb;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// bf16 tensors (cfg.storage = 1) multiplied on the bf16 matrix cores: v_mfma_f32_32x32x16_bf16 wants 8 consecutive k (token
// rows) of ONE column packed in a lane, the tensors are token-major.  One k-block = 16 token rows; lane half h loads rows
// 8h .. 8h+7 as 8 dwords (its two adjacent columns of each operand, as before) and forms the (row 2j, row 2j+1) pair of each
// column with one byte permute: 32 permutes + 4 matrix instructions per 16 rows where the widening form (DwStage<true> +
// fp32 MFMA above) spends 64 conversions + 32 fp32 MFMAs = ~2,100 cycles of the SIMD's vector ALU.
struct DwStageM { unsigned a[8]; unsigned b[8]; };
__device__ __forceinline__ void dw_issue_m(DwStageM& st, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, int voa, int vob,
                                           int soa, int sob, int rowa, int rowb) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        st.a[s] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(ra, voa, soa + s * rowa, 0);
        st.b[s] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rb, vob, sob + s * rowb, 0);
    }
}
// nkb: k-blocks of 16 rows; crow: first row of this lane half's 8; BIAS: also the column sums of the gradient operand
template <bool MASK, bool BIAS, int STAGES>
__device__ __forceinline__ void dw_loop_m(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, int voa, int vob, int soa, int sob,
                                          int rowa, int rowb, int nkb, int crow, int kend, f32x16 (&acc)[2][2], f32x2& bs) {
    DwStageM st[STAGES];
#pragma unroll
    for (int u = 0; u < STAGES; ++u) {
        dw_issue_m(st[u], ra, rb, voa, vob, soa, sob, rowa, rowb);
        soa += 16 * rowa; sob += 16 * rowb;
    }
    for (int kb = 0; kb < nkb; kb += STAGES) {
#pragma unroll
        for (int u = 0; u < STAGES; ++u) {
            if (kb + u >= nkb) break;
            DwStageM& cur = st[u];
            u32x4 a0, a1, b0, b1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned x0 = cur.a[2 * j], x1 = cur.a[2 * j + 1], y0 = cur.b[2 * j], y1 = cur.b[2 * j + 1];
                if (MASK) {
                    const bool ok0 = crow + 2 * j < kend, ok1 = crow + 2 * j + 1 < kend;
                    x0 = ok0 ? x0 : 0u; y0 = ok0 ? y0 : 0u; x1 = ok1 ? x1 : 0u; y1 = ok1 ? y1 : 0u;
                }
                a0[j] = (x0 & 0xFFFFu) | (x1 << 16); a1[j] = (x0 >> 16) | (x1 & 0xFFFF0000u);
                b0[j] = (y0 & 0xFFFFu) | (y1 << 16); b1[j] = (y0 >> 16) | (y1 & 0xFFFF0000u);
                if (BIAS) { bs.x += bf_lo(x0) + bf_lo(x1); bs.y += bf_hi(x0) + bf_hi(x1); }
            }
            acc[0][0] = mfma_bf16(a0, b0, acc[0][0]);
            acc[0][1] = mfma_bf16(a0, b1, acc[0][1]);
            acc[1][0] = mfma_bf16(a1, b0, acc[1][0]);
            acc[1][1] = mfma_bf16(a1, b1, acc[1][1]);
            crow += 16;
            __builtin_amdgcn_sched_barrier(0);
            dw_issue_m(st[u], ra, rb, voa, vob, soa, sob, rowa, rowb);
            soa += 16 * rowa; sob += 16 * rowb;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
// The same 16-row k-blocks for fp32 operands whose PRODUCT may be bf16 (the loss head under bf16 storage: d loss / d logits and
// h_last are fp32 tensors): lane half h loads rows 8h .. 8h+7 as 8 float2 and rounds the (row 2j, row 2j+1) pairs of its two
// columns with v_cvt_pk_bf16_f32 -- 16 conversions + 4 matrix instructions per 16 rows instead of 32 fp32 MFMAs.
struct DwStageC { f32x2 a[8]; f32x2 b[8]; };
__device__ __forceinline__ void dw_issue_c(DwStageC& st, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, int voa, int vob,
                                           int soa, int sob, int rowa, int rowb) {
#pragma unroll
    for (int s = 0; s < 8; ++s) { st.a[s] = bld2(ra, voa, soa + s * rowa); st.b[s] = bld2(rb, vob, sob + s * rowb); }
}
template <bool MASK, int STAGES>
__device__ __forceinline__ void dw_loop_c(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, int voa, int vob, int soa, int sob,
                                          int rowa, int rowb, int nkb, int crow, int kend, f32x16 (&acc)[2][2]) {
    DwStageC st[STAGES];
#pragma unroll
    for (int u = 0; u < STAGES; ++u) {
        dw_issue_c(st[u], ra, rb, voa, vob, soa, sob, rowa, rowb);
        soa += 16 * rowa; sob += 16 * rowb;
    }
    for (int kb = 0; kb < nkb; kb += STAGES) {
#pragma unroll
        for (int u = 0; u < STAGES; ++u) {
            if (kb + u >= nkb) break;
            DwStageC& cur = st[u];
            u32x4 a0, a1, b0, b1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x2 x0 = cur.a[2 * j], x1 = cur.a[2 * j + 1], y0 = cur.b[2 * j], y1 = cur.b[2 * j + 1];
                if (MASK) {
                    const bool ok0 = crow + 2 * j < kend, ok1 = crow + 2 * j + 1 < kend;
                    if (!ok0) { x0 = f32x2{0.f, 0.f}; y0 = x0; }
                    if (!ok1) { x1 = f32x2{0.f, 0.f}; y1 = x1; }
                }
                a0[j] = pk_bf16(x0.x, x1.x); a1[j] = pk_bf16(x0.y, x1.y);
                b0[j] = pk_bf16(y0.x, y1.x); b1[j] = pk_bf16(y0.y, y1.y);
            }
            acc[0][0] = mfma_bf16(a0, b0, acc[0][0]);
            acc[0][1] = mfma_bf16(a0, b1, acc[0][1]);
            acc[1][0] = mfma_bf16(a1, b0, acc[1][0]);
            acc[1][1] = mfma_bf16(a1, b1, acc[1][1]);
            crow += 16;
            __builtin_amdgcn_sched_barrier(0);
            dw_issue_c(st[u], ra, rb, voa, vob, soa, sob, rowa, rowb);
            soa += 16 * rowa; sob += 16 * rowb;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
#define DW_STAGES_M 3         // 48 token rows in flight per wave (the widening form: 5 x 8)

// One workgroup = one (problem, 64 x 64 tile, slab slice): its 4 waves take the 4 quarters of the slice, meet in LDS,
// wave 0 writes the slab.
// STAGES = k-blocks in flight per wave = the granule the wave's k range is rounded up to: 5 for the block weight
// gradients (10 k-blocks per wave at C1), 4 for the logits backward (8 resp. 16 k-blocks per wave: no padding MFMAs).
// BFM: the instantiation for plans with bf16 storage -- bf16 operands go to the bf16 matrix cores (dw_loop_m); the fp32
// instantiation (every other plan, the logits backward) carries neither that loop nor the widening one
// CVT: fp32 operands, bf16 product (dw_loop_c; the logits backward of a bf16-storage plan; no bias sums on that path)
template <int STAGES = DW_STAGES, bool BFM = false, bool CVT = false>
__device__ __forceinline__ void dw_wg_body(const DwProblem& Q, int m0, int n0, int slab, float (*red)[66][64]) {
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    // this wave's quarter of the slab slice
    const int sub = Q.kchunk >> 2;                                       // multiple of 8
    const int kbeg = slab * Q.kchunk + wv * sub, kend = min(Q.K, kbeg + sub);
    const int nkb = kend > kbeg ? (kend - kbeg + 7) >> 3 : 0;           // k-blocks of 8 token rows (rows past kend count as zero)
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;
    f32x2 bs = {0.f, 0.f};
    if (nkb > 0) {                                       // wave-uniform; empty quarters (pruned top block) add zeros
        // raw buffer descriptors over exactly the operands' bytes ([K] rows at their stride, the last one M resp. N wide;
        // < 2 GB, checked by the host): the un-predicated prefetch runs up to DW_STAGES + 1 k-blocks past a slice, and
        // what falls past the operand returns 0 from the hardware range check instead of touching memory
        const int esz = Q.bf16 ? 2 : 4;                                         // bytes per operand element
        const int recs_a = (int)(((long)(Q.K - 1) * Q.lda + Q.M) * esz), recs_b = (int)(((long)(Q.K - 1) * Q.ldb + Q.N) * esz);
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)Q.A, 0, recs_a, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)Q.B, 0, recs_b, 0x00020000);
        const int rowa = (int)Q.lda * esz, rowb = (int)Q.ldb * esz;             // bytes per token row
        const int voa = 4 * half * rowa + (m0 + 2 * l31) * esz, vob = 4 * half * rowb + (n0 + 2 * l31) * esz;
        const int soa = kbeg * rowa, sob = kbeg * rowb;
        const int crow = kbeg + 4 * half;
        const bool full = kbeg + 8 * nkb <= kend;                            // no partial last k-block: the unmasked loop
#define DW_RUN(BFV) \
        if (Q.gelu) { \
            if (full) dw_loop<true, false, BFV, STAGES>(ra, rb, voa, vob, soa, sob, rowa, rowb, nkb, crow, kend, acc, bs); \
            else dw_loop<true, true, BFV, STAGES>(ra, rb, voa, vob, soa, sob, rowa, rowb, nkb, crow, kend, acc, bs); \
        } else { \
            if (full) dw_loop<false, false, BFV, STAGES>(ra, rb, voa, vob, soa, sob, rowa, rowb, nkb, crow, kend, acc, bs); \
            else dw_loop<false, true, BFV, STAGES>(ra, rb, voa, vob, soa, sob, rowa, rowb, nkb, crow, kend, acc, bs); \
        }
        if constexpr (CVT) {
            const int nkb16 = (kend - kbeg + 15) >> 4;
            const int voa_c = 8 * half * rowa + (m0 + 2 * l31) * 4, vob_c = 8 * half * rowb + (n0 + 2 * l31) * 4;
            const int crow_c = kbeg + 8 * half;
            if (kbeg + 16 * nkb16 <= kend) dw_loop_c<false, STAGES>(ra, rb, voa_c, vob_c, soa, sob, rowa, rowb, nkb16, crow_c, kend, acc);
            else dw_loop_c<true, STAGES>(ra, rb, voa_c, vob_c, soa, sob, rowa, rowb, nkb16, crow_c, kend, acc);
        } else if constexpr (BFM) {
        if (Q.bf16 && !Q.gelu) {                             // bf16 operands on the bf16 matrix cores (16-row k-blocks)
            const int nkb16 = (kend - kbeg + 15) >> 4;
            const int voa_m = 8 * half * rowa + (m0 + 2 * l31) * 2, vob_m = 8 * half * rowb + (n0 + 2 * l31) * 2;
            const int crow_m = kbeg + 8 * half;
            const bool full16 = kbeg + 16 * nkb16 <= kend, bias = Q.bslab != nullptr && n0 == 0;
            if (full16) {
                if (bias) dw_loop_m<false, true, DW_STAGES_M>(ra, rb, voa_m, vob_m, soa, sob, rowa, rowb, nkb16, crow_m, kend, acc, bs);
                else dw_loop_m<false, false, DW_STAGES_M>(ra, rb, voa_m, vob_m, soa, sob, rowa, rowb, nkb16, crow_m, kend, acc, bs);
            } else dw_loop_m<true, true, DW_STAGES_M>(ra, rb, voa_m, vob_m, soa, sob, rowa, rowb, nkb16, crow_m, kend, acc, bs);
        } else if (Q.bf16) { DW_RUN(true) } else { DW_RUN(false) }
        } else { DW_RUN(false) }
#undef DW_RUN
    }
    // bias gradient: column sums of the gradient operand; the two lane halves hold different token rows
    bs.x = xor32_sum(bs.x); bs.y = xor32_sum(bs.y);
    // the four quarters meet in LDS: waves 1..3 park their registers, wave 0 adds them in a fixed order
    if (wv > 0) {
        float (*mine)[64] = red[wv - 1];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[(i * 2 + jn) * 16 + r][lane] = acc[i][jn][r];
        mine[64][lane] = bs.x; mine[65][lane] = bs.y;
    }
    __syncthreads();
    if (wv > 0) return;
#pragma unroll 1
    for (int w = 0; w < 3; ++w) {              // not unrolled: one wave's 66 values live at a time
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][jn][r] += red[w][(i * 2 + jn) * 16 + r][lane];
        bs.x += red[w][64][lane]; bs.y += red[w][65][lane];
    }
    // slab: register r of lane (l31, half) of tile (i, jn) is C[m0 + 2 (rho(r) + 4 half) + i][n0 + 2 l31 + jn]
    float* C = Q.slab + (long)slab * Q.M * Q.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 2 * (rho(r) + 4 * half) + i;
            if (m < Q.M) gst2(C + (long)m * Q.N + n0 + 2 * l31, f32x2{acc[i][0][r], acc[i][1][r]});
        }
    if (Q.bslab && n0 == 0 && half == 0 && m0 + 2 * l31 + 1 < Q.M) gst2(Q.bslab + (long)slab * Q.M + m0 + 2 * l31, bs);
}


// The host builds the unit table (64 x 64 tiles of the six problems: 4 + 4 + 4 = 12 units at hidden = 64).
// tk.state != null: one extra (last) workgroup closes the optimisation step here -- mean loss, Adam's t and bias
// corrections, next dropout step, batch cursor (kernels.h, tick_body) -- so that the step's final kernel can reduce AND
// update with a read-only state.  Every reader of the dropout step / cursor of this step has run before this launch.
// sc.nblocks > 0: the lookup-path scatter of the item-table gradient (embed_scatter_block, 64-token chunks) rides here
// too, as the workgroups after the weight-gradient ones: light, latency-bound blocks that fill the slots the big
// workgroups leave free instead of a launch of their own (11.6 us + a launch boundary at C1).
struct ScatterP { const float* de; const int* ids32; int T; float* dE; int nblocks; };
// (One instantiation for fp32 and bf16-storage plans: with the bf16 loop compiled in, the fp32 path of this kernel measured 0.5 %
// of the step FASTER than an fp32-only instantiation -- 0.1568 against 0.1575 ms, three A/B pairs on one box; scheduling luck,
// kept because it is measured.  The logits backward below is the opposite case and has two.)
__global__ void __launch_bounds__(256)
dw_direct_kernel(const DwP G, const TickP tk, const ScatterP sc) {
    __shared__ __attribute__((aligned(16))) float red[3][66][64];      // accumulators (64) + bias sums (2) of waves 1..3
    static_assert(sizeof(red) >= (SCATTER_FLOATS + 2 * (SCATTER_FLOATS / 64)) * 4, "scatter scratch fits the reduction scratch");
    {
        const int nmain = 8 * ((G.nslab + 7) >> 3) * (G.nunits - G.nsmall) + G.nsmall * G.small_slabs;
        if ((int)blockIdx.x >= nmain) {
            const int x = blockIdx.x - nmain;
            if (x < sc.nblocks) embed_scatter_block<16>(sc.de, sc.ids32, sc.T, 64, sc.dE, x, &red[0][0][0]);
            else if (tk.state) tick_body(tk, &red[0][0][0]);
            return;
        }
    }
    // XCD-aware mapping: workgroups are dealt round-robin to the 8 XCDs (id % 8), each with its own L2.  All units of
    // one slab slice read the same token rows (X feeds the q/k/v tiles, hmix and dT2 four tiles each), so a slice is
    // kept on ONE XCD and the re-reads hit that L2.
    int slab, ui;
    const int nbig = G.nunits - G.nsmall, nbw = 8 * ((G.nslab + 7) >> 3) * nbig;     // big problems first: all resident at once
    if ((int)blockIdx.x < nbw) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        slab = xcd + 8 * (j / nbig);
        if (slab >= G.nslab) return;
        ui = G.nsmall + j % nbig;
    } else {                                              // the small ones fill the free slots and finish early
        const int bid = blockIdx.x - nbw;
        ui = bid % G.nsmall; slab = bid / G.nsmall;
    }
    const DwUnit u = G.U[ui];
    const DwProblem& Q = G.P[u.prob];
    if (slab >= Q.nslab) return;
    dw_wg_body<DW_STAGES, true>(Q, u.m0, u.n0, slab, red);
}

// =============================================================================================
// d(h_last) = dlogits . E, split-K over the catalogue, the same way: a wave owns 32 batch rows x 64 features and one
// catalogue slice.  dlogits is row-major [B][Vp]: lane (l31, half) takes 4 consecutive catalogue columns of batch row
// 32 mt + l31 with ONE 16-byte load per k-block; E is k-major [V][64]: 8-byte loads feed the two interleaved 32-column
// tiles (column 2 l31 + j).  Slab [split][B][64]; the consumer (top block backward) adds the splits.
// =============================================================================================
struct DhP {
    const float* A; long lda;          // dlogits [B][lda]
    const float* E;                    // [V][64]
    int B, V, kchunk, nsplit;          // kchunk: catalogue columns per split, multiple of 8
    float* slab;                       // [nsplit][B][64]
};

template <int STAGES = DW_STAGES, bool CVT = false>
__device__ __forceinline__ void dh_wave_body(const DhP& G, int wg) {
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    const int mtiles = (G.B + 31) >> 5;
    const int unit = wg * 4 + wv, mt = unit % mtiles, split = unit / mtiles;
    if (split >= G.nsplit) return;
    const int kbeg = split * G.kchunk, kend = min(G.V, kbeg + G.kchunk);
    const int nkb = kend > kbeg ? (kend - kbeg + 7) >> 3 : 0;
    const int m = min(32 * mt + l31, G.B - 1);                       // rows past B re-read the last row; never stored
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    if (nkb > 0) {
        // descriptors over exactly dlogits [B][lda] and E [V][64]: prefetch past either end reads 0, not memory
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)G.A, 0, (int)((long)G.B * G.lda * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)G.E, 0, (int)((long)G.V * 256), 0x00020000);
        if constexpr (CVT) {
            // bf16 product of the fp32 operands: 16 catalogue columns per k-block, lane half h takes columns 8h .. 8h+7 of its
            // batch row (two 16-byte loads, rounded in pairs) and rows 8h .. 8h+7 of E for its two feature columns
            const int nkb16 = (kend - kbeg + 15) >> 4;
            const int voa = (int)((long)m * G.lda * 4) + 32 * half, vob = 8 * half * 256 + 8 * l31;
            int soa = kbeg * 4, sob = kbeg * 256, ccol = kbeg + 8 * half;
            f32x4 sa[STAGES][2]; f32x2 sb[STAGES][8];
            auto issue = [&](int u) {
                sa[u][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, voa, soa, 0));
                sa[u][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, voa + 16, soa, 0));
#pragma unroll
                for (int s = 0; s < 8; ++s) sb[u][s] = bld2(rb, vob, sob + s * 256);
                soa += 64; sob += 16 * 256;
            };
#pragma unroll
            for (int u = 0; u < STAGES; ++u) issue(u);
            for (int kb = 0; kb < nkb16; kb += STAGES) {
#pragma unroll
                for (int u = 0; u < STAGES; ++u) {
                    if (kb + u >= nkb16) break;
                    f32x4 x0 = sa[u][0], x1 = sa[u][1];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {              // columns past the slice: other splits' / the next row's data
                        if (ccol + e >= kend) x0[e] = 0.f;
                        if (ccol + 4 + e >= kend) x1[e] = 0.f;
                    }
                    const u32x4 a = pk8(x0, x1);
                    u32x4 b0, b1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        b0[j] = pk_bf16(sb[u][2 * j].x, sb[u][2 * j + 1].x);
                        b1[j] = pk_bf16(sb[u][2 * j].y, sb[u][2 * j + 1].y);
                    }
                    acc0 = mfma_bf16(a, b0, acc0);
                    acc1 = mfma_bf16(a, b1, acc1);
                    ccol += 16;
                    __builtin_amdgcn_sched_barrier(0);
                    issue(u);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
        const int voa = (int)((long)m * G.lda * 4) + 16 * half, vob = 4 * half * 256 + 8 * l31;      // bytes
        int soa = kbeg * 4, sob = kbeg * 256, ccol = kbeg + 4 * half;
        f32x4 sa[STAGES]; f32x2 sb[STAGES][4];
        auto issue = [&](int u) {
            sa[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, voa, soa, 0));
#pragma unroll
            for (int s = 0; s < 4; ++s) sb[u][s] = bld2(rb, vob, sob + s * 256);
            soa += 32; sob += 8 * 256;
        };
#pragma unroll
        for (int u = 0; u < STAGES; ++u) issue(u);
        for (int kb = 0; kb < nkb; kb += STAGES) {
#pragma unroll
            for (int u = 0; u < STAGES; ++u) {
                if (kb + u >= nkb) break;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float a = ccol + s < kend ? sa[u][s] : 0.f;      // columns past the slice (other splits' / next row's data)
                    const f32x2 b = sb[u][s];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.y, acc1, 0, 0, 0);
                }
                ccol += 8;
                __builtin_amdgcn_sched_barrier(0);
                issue(u);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    }
    float* C = G.slab + (long)split * G.B * 64;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = 32 * mt + rho(r) + 4 * half;
        if (row < G.B) gst2(C + (long)row * 64 + 2 * l31, f32x2{acc0[r], acc1[r]});
    }
}

// The whole logits backward at the fused shape in ONE launch: workgroups [0, tiles) form dE = dlogits^T . h_last (the
// dense gradient of the item table, written straight into the gradient buffer: problem Q, 64-row tiles, rows >= M not
// stored), the rest form the split-K slabs of d(h_last).
// BFH: both products on the bf16 matrix cores, operands rounded in registers (plans with bf16 storage; the forward logits and the
// cross-entropy stay fp32)
template <bool BFH>
__global__ void __launch_bounds__(256)
logits_bwd_direct_kernel(const DwProblem Q, int tiles, const DhP H) {
    __shared__ __attribute__((aligned(16))) float red[3][66][64];
    if ((int)blockIdx.x < tiles) dw_wg_body<4, false, BFH>(Q, 64 * (int)blockIdx.x, 0, 0, red);
    else dh_wave_body<4, BFH>(H, (int)blockIdx.x - tiles);
}

// =============================================================================================
// logits[b][v] = h_last[b] . E[v] at hidden = 64, the same direct way (no LDS, no barriers): a wave owns 32 batch rows x 32
// items; lane (l31, half) loads the k half [32 half, 32 half + 32) of ITS batch row and of ITS item row as 8 + 8 16-byte
// loads (both operands are k-contiguous; the k order is any bijection as long as A and B share it) and chains 32 MFMAs.  The
// tiled GEMM this replaces on the fused path staged a K = 64 product through LDS twice (operands, then the C tile) for a
// 6.5 us launch at C1, 45 us at C4's 1,024 x 20,034.  Pad columns V .. Vp are written as 0 (BSAREC_BUF_LOGITS contract).
// =============================================================================================
struct LogitsP { const float* H; long ldh; const float* E; float* C; int B, V, Vp; };
__global__ void __launch_bounds__(256)
logits_direct_kernel(const LogitsP G) {
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    const int ntile = (G.Vp + 31) >> 5, mtiles = (G.B + 31) >> 5;
    const int unit = blockIdx.x * 4 + wv;                 // item tile fastest: the 4 waves of a workgroup share one batch-row block
    const int nt = unit % ntile, mt = unit / ntile;
    if (mt >= mtiles) return;
    const int m = min(32 * mt + l31, G.B - 1);             // rows / items past the end re-read the last one; never stored
    const int v = min(32 * nt + l31, G.V - 1);
    const float* ap = G.H + (long)m * G.ldh + 32 * half;
    const float* bp = G.E + (long)v * 64 + 32 * half;
    f32x4 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = gld4(ap + 4 * i);
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = gld4(bp + 4 * i);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], b[i][j], acc, 0, 0, 0);
    const int col = 32 * nt + l31;
    if (col >= G.Vp) return;
    const bool real = col < G.V;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = 32 * mt + rho(r) + 4 * half;
        if (row < G.B) gst(G.C + (long)row * G.Vp + col, real ? acc[r] : 0.f);
    }
}
