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
                                        // products stay fp32 MFMA on the widened values (the kernel is bound by operand bytes)
    int ce;                             // the gradient operand holds LOGITS [K = batch rows][M = catalogue]: d loss / d logits =
    float ce_scale;                     // (exp(x - lse[row]) - [col == answer[row]]) * ce_scale is formed while loading (CeP)
};

// Cross-entropy folded into the logits backward (src/model/bsarec.py:35 + its autograd): the logits kernel leaves
// per-(32-item tile, row) maxima and exp-sums; every workgroup of the backward combines them into lse[row] in LDS
// (ce_prologue) and turns logits into d loss / d logits on the way into the MFMA operands -- no CE launch, no dlogits
// round trip through memory.
#define CE_MAX_B 1024
struct CeP {
    const float* pmax; const float* psum;     // [ntile][B]
    int ntile, B, V;
    const int64_t* answers;                   // [B]
    const float* logits; long ldl;            // [B][ldl]
    float* loss_rows;                         // [B] (written by workgroup 0)
};
__device__ __forceinline__ void ce_prologue(const CeP& C, float* __restrict__ s_lse, int* __restrict__ s_ans) {
    // one thread per batch row; 16 tiles' (max, sum) pairs in flight per step (a rolled loop would pay one L2 round trip
    // per tile: 2 x 107 dependent trips were 16 us), combined online: m' = max(m, pm), s = s e^(m - m') + ps e^(pm - m')
    for (int b = threadIdx.x; b < C.B; b += blockDim.x) {
        float m = -INFINITY, sum = 0.f;
        for (int t0 = 0; t0 < C.ntile; t0 += 16) {
            float pm[16], ps[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int t = min(t0 + k, C.ntile - 1);                 // branch-free: the tail re-reads the last tile, weight 0
                pm[k] = C.pmax[(long)t * C.B + b]; ps[k] = C.psum[(long)t * C.B + b];
            }
            float mm = m;
#pragma unroll
            for (int k = 0; k < 16; ++k) mm = fmaxf(mm, pm[k]);
            float acc = sum * expf(m - mm);                             // m = -inf at the start: exp(-inf) = 0
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += (t0 + k < C.ntile ? ps[k] : 0.f) * expf(pm[k] - mm);
            m = mm; sum = acc;
        }
        const float lse = m + logf(sum);
        int a = (int)C.answers[b];
        a = a < 0 ? 0 : (a >= C.V ? C.V - 1 : a);
        s_lse[b] = lse; s_ans[b] = a;
        if (blockIdx.x == 0) C.loss_rows[b] = lse - C.logits[(long)b * C.ldl + a];
    }
    __syncthreads();
}
__device__ __forceinline__ float ce_grad(float x, float lse, int ans, int col, int V, float scale) {
    return col < V ? (expf(x - lse) - (col == ans ? 1.0f : 0.0f)) * scale : 0.f;
}

struct DwUnit { short prob, m0, n0, pad; };

struct DwP {
    DwProblem P[DW_MAX_PROB];
    DwUnit U[DW_MAX_UNITS];
    int nunits, nslab;
    // Units [0, nsmall) belong to the tiny problems of the pruned top block (K = B or B*h rows, `small_slabs` slab
    // slices each): they take the LAST nsmall * small_slabs workgroups of the grid on their own -- no empty
    // workgroups; the big problems' 8 * ceil(nslab / 8) * (nunits - nsmall) workgroups come first so that all of them
    // are resident from the start (480 of the 512 slots at C1), the small ones fill the rest and finish early.
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

// (LDS pointers carry their address space: through a generic pointer these reads become flat_load, which counts on vmcnt
// AND lgkmcnt -- every read would first drain the whole global prefetch pipeline)
#define AS_LDS __attribute__((address_space(3)))
struct CeLoop { const AS_LDS float* s_lse; const AS_LDS int* s_ans; int col0, V, rows; float scale; };    // col0: this lane's first column
template <bool GELU, bool MASK, bool BF, bool CE = false>
__device__ __forceinline__ void dw_loop(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, int voa, int vob, int soa, int sob,
                                        int rowa, int rowb, int nkb, int crow, int kend, f32x16 (&acc)[2][2], f32x2& bs,
                                        const CeLoop ce = CeLoop{}) {
    DwStage<BF> st[DW_STAGES];
#pragma unroll
    for (int u = 0; u < DW_STAGES; ++u) {
        dw_issue(st[u], ra, rb, voa, vob, soa, sob, rowa, rowb);
        soa += 8 * rowa; sob += 8 * rowb;
    }
    for (int kb = 0; kb < nkb; kb += DW_STAGES) {       // nkb is a multiple of DW_STAGES (rows past kend count as zero)
#pragma unroll
        for (int u = 0; u < DW_STAGES; ++u) {
            DwStage<BF>& cur = st[u];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f32x2 a, b;
                if constexpr (BF) { a = f32x2{bf_lo(cur.a[s]), bf_hi(cur.a[s])}; b = f32x2{bf_lo(cur.b[s]), bf_hi(cur.b[s])}; }
                else { a = cur.a[s]; b = cur.b[s]; }
                if constexpr (CE) {            // logits -> d loss / d logits of batch row crow + s (rows past the batch: masked below)
                    const int row = min(crow + s, ce.rows - 1);
                    const float l = ce.s_lse[row];
                    const int an = ce.s_ans[row];
                    a.x = ce_grad(a.x, l, an, ce.col0, ce.V, ce.scale);
                    a.y = ce_grad(a.y, l, an, ce.col0 + 1, ce.V, ce.scale);
                }
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

// One workgroup = one (problem, 64 x 64 tile, slab slice): its 4 waves take the 4 quarters of the slice, meet in LDS,
// wave 0 writes the slab.
__device__ __forceinline__ void dw_wg_body(const DwProblem& Q, int m0, int n0, int slab, float (*red)[66][64],
                                           const float* s_lse = nullptr, const int* s_ans = nullptr) {
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    // this wave's quarter of the slab slice
    const int sub = Q.kchunk >> 2;                                       // multiple of 8
    const int kbeg = slab * Q.kchunk + wv * sub, kend = min(Q.K, kbeg + sub);
    const int nkb = kend > kbeg ? ((kend - kbeg + 8 * DW_STAGES - 1) / (8 * DW_STAGES)) * DW_STAGES : 0;
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
        const bool full = kbeg + 8 * nkb <= kend;
#define DW_RUN(BFV) \
        if (Q.gelu) { \
            if (full) dw_loop<true, false, BFV>(ra, rb, voa, vob, soa, sob, rowa, rowb, nkb, crow, kend, acc, bs); \
            else dw_loop<true, true, BFV>(ra, rb, voa, vob, soa, sob, rowa, rowb, nkb, crow, kend, acc, bs); \
        } else { \
            if (full) dw_loop<false, false, BFV>(ra, rb, voa, vob, soa, sob, rowa, rowb, nkb, crow, kend, acc, bs); \
            else dw_loop<false, true, BFV>(ra, rb, voa, vob, soa, sob, rowa, rowb, nkb, crow, kend, acc, bs); \
        }
        if (Q.ce) {                                      // the dE problem of the logits backward with the CE folded in (fp32, no GELU)
            const CeLoop ce{(const AS_LDS float*)s_lse, (const AS_LDS int*)s_ans, m0 + 2 * l31, Q.M, Q.K, Q.ce_scale};
            dw_loop<false, true, false, true>(ra, rb, voa, vob, soa, sob, rowa, rowb, nkb, crow, kend, acc, bs, ce);
        } else if (Q.bf16) { DW_RUN(true) } else { DW_RUN(false) }
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
    dw_wg_body(Q, u.m0, u.n0, slab, red);
}

// =============================================================================================
// d(h_last) = dlogits . E, split-K over the catalogue, the same way: a wave owns 32 batch rows x 64 features and one
// catalogue slice.  dlogits is row-major [B][Vp]: lane (l31, half) takes 4 consecutive catalogue columns of batch row
// 32 mt + l31 with ONE 16-byte load per k-block; E is k-major [V][64]: 8-byte loads feed the two interleaved 32-column
// tiles (column 2 l31 + j).  Slab [split][B][64]; the consumer (top block backward) adds the splits.
// =============================================================================================
struct DhP {
    const float* A; long lda;          // dlogits [B][lda] (ce != 0: the logits, turned into dlogits while loading)
    const float* E;                    // [V][64]
    int B, V, kchunk, nsplit;          // kchunk: catalogue columns per split, multiple of 8
    float* slab;                       // [nsplit][B][64]
    int ce; float ce_scale;
};

__device__ __forceinline__ void dh_wave_body(const DhP& G, int wg, const float* s_lse = nullptr, const int* s_ans = nullptr) {
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    const int mtiles = (G.B + 31) >> 5;
    const int unit = wg * 4 + wv, mt = unit % mtiles, split = unit / mtiles;
    if (split >= G.nsplit) return;
    const int kbeg = split * G.kchunk, kend = min(G.V, kbeg + G.kchunk);
    const int nkb = kend > kbeg ? ((kend - kbeg + 8 * DW_STAGES - 1) / (8 * DW_STAGES)) * DW_STAGES : 0;
    const int m = min(32 * mt + l31, G.B - 1);                       // rows past B re-read the last row; never stored
    const float lse_m = G.ce ? ((const AS_LDS float*)s_lse)[m] : 0.f;
    const int ans_m = G.ce ? ((const AS_LDS int*)s_ans)[m] : 0;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    if (nkb > 0) {
        // descriptors over exactly dlogits [B][lda] and E [V][64]: prefetch past either end reads 0, not memory
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)G.A, 0, (int)((long)G.B * G.lda * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)G.E, 0, (int)((long)G.V * 256), 0x00020000);
        const int voa = (int)((long)m * G.lda * 4) + 16 * half, vob = 4 * half * 256 + 8 * l31;      // bytes
        int soa = kbeg * 4, sob = kbeg * 256, ccol = kbeg + 4 * half;
        f32x4 sa[DW_STAGES]; f32x2 sb[DW_STAGES][4];
        auto issue = [&](int u) {
            sa[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, voa, soa, 0));
#pragma unroll
            for (int s = 0; s < 4; ++s) sb[u][s] = bld2(rb, vob, sob + s * 256);
            soa += 32; sob += 8 * 256;
        };
#pragma unroll
        for (int u = 0; u < DW_STAGES; ++u) issue(u);
        for (int kb = 0; kb < nkb; kb += DW_STAGES) {
#pragma unroll
            for (int u = 0; u < DW_STAGES; ++u) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    float a = sa[u][s];
                    if (G.ce) a = ce_grad(a, lse_m, ans_m, ccol + s, G.V, G.ce_scale);
                    a = ccol + s < kend ? a : 0.f;                         // columns past the slice (other splits' / next row's data)
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
__global__ void __launch_bounds__(256)
logits_bwd_direct_kernel(const DwProblem Q, int tiles, const DhP H, const CeP C) {
    __shared__ __attribute__((aligned(16))) float red[3][66][64];
    __shared__ float s_lse[CE_MAX_B];
    __shared__ int s_ans[CE_MAX_B];
    if (Q.ce) ce_prologue(C, s_lse, s_ans);              // every workgroup: lse / answer of all batch rows -> LDS
    if ((int)blockIdx.x < tiles) dw_wg_body(Q, 64 * (int)blockIdx.x, 0, 0, red, s_lse, s_ans);
    else dh_wave_body(H, (int)blockIdx.x - tiles, s_lse, s_ans);
}

// =============================================================================================
// logits = h_last . E^T at hidden = 64 (src/model/bsarec.py:33-34) with the cross-entropy statistics as its epilogue:
// one wave per (64-item tile, 32-row tile), operands straight from global memory (K = 64: eight 16-byte loads per
// lane and operand).  Computed transposed -- A = E rows (lane = item), B = h rows (lane = batch row) -- so that the
// accumulators have the batch row on the lane and the 64 items in registers: the row maximum and the exp-sum of the
// tile are register reductions plus one cross-half exchange.  Writes logits [B][ldl] (pad columns 0) and the per-tile
// (max, sum exp(x - max)) pair of every row; the backward combines them (ce_prologue).
// =============================================================================================
struct LogitsStatsP {
    const float* h; long ldh;          // h_last rows [B][ldh]
    const float* E;                    // [V][64]
    int B, V; long ldl;
    float* logits; float* pmax; float* psum;       // [B][ldl]; [ntile][B] each
};
__global__ void __launch_bounds__(256)
logits_stats_kernel(const LogitsStatsP P) {
    // one wave = 64 items (two 32-item accumulators sharing the h fragment) x 32 batch rows; statistic tiles of 64 items
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    const int mtiles = (P.B + 31) >> 5, ntiles = (P.V + 63) >> 6;
    const int unit = blockIdx.x * 4 + wv, nt = unit / mtiles, mt = unit - nt * mtiles;
    if (nt >= ntiles) return;
    const int brow = min(32 * mt + l31, P.B - 1);
    const float* hb = P.h + (long)brow * P.ldh + 4 * half;
    const float* ea0 = P.E + (long)min(64 * nt + l31, P.V - 1) * 64 + 4 * half;
    const float* ea1 = P.E + (long)min(64 * nt + 32 + l31, P.V - 1) * 64 + 4 * half;
    f32x4 a0[8], a1[8], b[8];
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) { b[kb] = gld4(hb + 8 * kb); a0[kb] = gld4(ea0 + 8 * kb); a1[kb] = gld4(ea1 + 8 * kb); }
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[kb][s], b[kb][s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[kb][s], b[kb][s], acc[1], 0, 0, 0);
        }
    // lane = batch row 32 mt + l31; register r of accumulator i = item 64 nt + 32 i + rho(r) + 4 half
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int v = 64 * nt + 32 * i + rho(r) + 4 * half;
            if (v >= P.V) acc[i][r] = 0.f;                 // pad columns of the logits buffer are 0
            else mx = fmaxf(mx, acc[i][r]);
        }
    mx = xor32_max(mx);
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int v = 64 * nt + 32 * i + rho(r) + 4 * half;
            if (v < P.V) sm += expf(acc[i][r] - mx);
        }
    sm = xor32_sum(sm);
    const int bi = 32 * mt + l31;
    if (bi < P.B) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float* out = P.logits + (long)bi * P.ldl + 64 * nt + 32 * i + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (64 * nt + 32 * i + 8 * g + 4 * half < P.ldl)
                    gst4(out + 8 * g, f32x4{acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]});
        }
        if (half == 0) { gst(P.pmax + (long)nt * P.B + bi, mx); gst(P.psum + (long)nt * P.B + bi, sm); }
    }
}
