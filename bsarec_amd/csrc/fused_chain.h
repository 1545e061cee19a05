// Register-chain BSARecBlock kernels for the headline shape class (hidden = 64, L <= 64, fp32): round 3.
//
// The round-2 block kernels (fused_layer.h) keep the sequence's activations in LDS and run ~25 workgroup-wide phases
// (MFMA burst -> accumulators to LDS -> barrier -> row pass -> barrier ...): every wave reaches its matrix bursts and its
// vector epilogues together, so the matrix pipe idles through every epilogue (23 % MFMA-busy, 51 % of the wave cycles
// parked on barriers / waitcnt -- profiles/r02_pmc_C1.csv).  Here the block is evaluated TRANSPOSED with
// v_mfma_f32_16x16x4_f32 (D = A.B, 16 x 16 output, K = 4):
//
//     Y^T [out feature][token] = W [out][in] . X^T [in][token]
//
//   * A = weight rows straight from L2 (lane (i, g) = (lane & 15, lane >> 4) reads the 16 bytes W[16 o + i][16 c + 4 g ..+3]),
//   * B = an activation "slab" = 16 features x 16 tokens = one f32x4 per lane: lane (n, g) holds X[token n][16 c + 4 g + r],
//   * the accumulator of a 16 x 16 product is lane (n, g), register r = Y[token n][16 o + 4 g + r] -- AGAIN a slab.
//     MFMA number r of a 4-deep k-group multiplies A element r with B register r, so k-slot g <-> feature 4 g + r on both
//     operands and the accumulator of one Linear layer is the B operand of the next as it stands.
// One wave owns a 16-token tile and carries it through the WHOLE attention branch + feed-forward in registers: bias,
// softmax (lane = query, registers = keys), dropout, residual, LayerNorm (a token's 64 features = 16 registers x the 4
// lanes n, n+16, n+32, n+48: two v_permlane swaps), erf-GELU, all on accumulators.  LDS carries only what tokens exchange:
// the x tile, K, V^T (attention mixes tokens), the FrequencyLayer output and three small hand-overs inside a wave pair.
// Measured on the way (tools/micro/, profiles/r03_micro_*): an fp32 MFMA occupies the SIMD's vector ALU -- while it
// executes NO vector instruction of either resident wave issues (a bf16 MFMA does not do that) -- and one wave alone
// issues a vector instruction only every ~4.4 cycles, two waves every ~2.  So for fp32 the SIMD's time is MFMA cycles +
// VALU cycles, additive, and both waves of every SIMD must carry work.  Waves w and w + 4 (one SIMD) form a PAIR that owns
// token tile w: they split the Q / K / V output slabs, the attention heads, the dense product (K-split), the
// FrequencyLayer rows and the 16 inner slabs of the feed-forward, and meet three times through LDS (dense partial sums,
// the mixed activation, the feed-forward partial sums).  All 8 waves stream the block's weights, coalesced, into a
// two-slot LDS ring in MFMA-fragment order; one workgroup barrier per 16 KiB unit (14 units) + 3 hand-over barriers.
// Same arithmetic (fp32 MFMA = fmaf chains), same Philox stream, same saved tensors as fused_layer_fwd_kernel; parity
// against the oracle by the same tests (tests/test_gpu_parity.py runs both kernel sets).
// src/model/bsarec.py:56-104, src/model/_modules.py:22-140.
#pragma once
#include "fused_layer.h"

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// (rows_of / quad_rows_sum / quad_rows_max: common.h)

// LayerNorm of a token held as 4 slabs (16 registers x 4 lanes)
__device__ __forceinline__ void ln_slabs(const f32x4 (&v)[4], float eps, f32x4 (&xh)[4], float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
    const float mean = quad_rows_sum(s) * (1.0f / 64.0f);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) { xh[c] = v[c] - mean; q += (xh[c].x * xh[c].x + xh[c].y * xh[c].y) + (xh[c].z * xh[c].z + xh[c].w * xh[c].w); }
    const float var = quad_rows_sum(q) * (1.0f / 64.0f);
    rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int c = 0; c < 4; ++c) xh[c] = xh[c] * rstd;
}

// embedding front-end / x-tile load of phase 0 (shared with nothing: the round-2 kernel keeps its own copy inline)
__device__ __forceinline__ void chain_phase0(float* sX, int* sIds, float* sTab, const DropSeed& dseed, int L, int cb, long tok0, int b) {
    const int tid = threadIdx.x;
    const float* const eE = KARG(FusedFwdP, e_E);
    if (eE) {
        const GatherP gp = KARG(FusedFwdP, e_gp);
        const int64_t* const eids = KARG(FusedFwdP, e_ids);
        const float* const epos = KARG(FusedFwdP, e_pos);
        const int V = KARG(FusedFwdP, e_V);
        const float eeps = KARG(FusedFwdP, eps);
        long src = 0;
        if (gp.table) {
            src = *(const AS_GLOBAL long long*)gp.cursor + b;
            src = src < gp.n ? (long)*(const AS_GLOBAL int64_t*)(gp.perm + src) : 0;
            if (tid == 0) *(AS_GLOBAL int64_t*)(gp.ans_out + b) = *(const AS_GLOBAL int64_t*)(gp.ans_table + src);
        }
        const f32x4 eg = gld4(KARG(FusedFwdP, e_g) + ((tid & 15) << 2)), eb = gld4(KARG(FusedFwdP, e_b) + ((tid & 15) << 2));
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int idx = tid + p * 512, r = idx >> 4, c4 = (idx & 15) << 2;
            const bool ok = r < L;
            const long e = (tok0 + r) * 64 + c4;
            f32x4 v = {0, 0, 0, 0};
            int id = 0;
            if (ok) {
                int64_t id64;
                if (gp.table) {
                    id64 = *(const AS_GLOBAL int64_t*)(gp.table + src * L + r);
                    if (c4 == 0) *(AS_GLOBAL int64_t*)(gp.ids_out + tok0 + r) = id64;
                } else id64 = *(const AS_GLOBAL int64_t*)(eids + tok0 + r);
                id = (int)id64;
                id = id < 0 ? 0 : (id >= V ? V - 1 : id);     // defensive clamp: never read outside the table
                v = gld4(eE + (long)id * 64 + c4) + gld4(epos + (long)r * 64 + c4);
            }
            if (c4 == 0) { sIds[r] = id; if (ok) *(AS_GLOBAL int*)(KARG(FusedFwdP, e_ids32) + tok0 + r) = id; }
            const float mean = group_sum<16>(v.x + v.y + v.z + v.w) * (1.0f / 64.0f);
            f32x4 dl = {0, 0, 0, 0};
            if (ok) dl = v - mean;
            const float var = group_sum<16>(dl.x * dl.x + dl.y * dl.y + dl.z * dl.z + dl.w * dl.w) * (1.0f / 64.0f);
            const float rs = 1.0f / sqrtf(var + eeps);
            f32x4 y = {0, 0, 0, 0};
            if (ok) {
                const f32x4 xh = dl * rs;
                y = (eg * xh + eb) * drop_mult4(KARG(FusedFwdP, e_drop), dseed, (uint64_t)e >> 2);
                gst4(KARG(FusedFwdP, e_xhat) + e, xh);
                gst4(KARG(FusedFwdP, e_X0) + e, y);
                if (c4 == 0) gst(KARG(FusedFwdP, e_rstd) + tok0 + r, rs);
            }
            st4(sX + r * FS + c4, y);
        }
    } else {
        const float* const X = KARG(FusedFwdP, X);
        const int* const ids32 = KARG(FusedFwdP, ids32);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int idx = tid + p * 512, r = idx >> 4, c4 = (idx & 15) << 2;
            f32x4 v = gld4(X + (tok0 + min(r, L - 1)) * 64 + c4);
            if (r >= L) v = f32x4{0, 0, 0, 0};
            st4(sX + r * FS + c4, v);
        }
        if (tid < 64) sIds[tid] = tid < L ? gldi(ids32 + (tok0 + tid)) : 0;
    }
    build_twiddle_table(KARG(FusedFwdP, tw), L, cb, sTab);
}

// ---- weight stream -----------------------------------------------------------------------------------------------
// A fragment = 16 weight rows x 16 in-features = the A operands of 4 MFMAs for all 64 lanes = 1 KiB: lane (i, g) uses the
// 16 bytes W[16 o + i][16 c + 4 g ..+3].  Read straight from global that is 16 cache lines per quarter-wave (the first
// build of this kernel did: ~64 address-path cycles per wave instruction; the stamps showed every stage 2-3x over its MFMA
// time).  So the 8 waves stream the block's 196 KB of weights ONCE per workgroup with coalesced loads (a quarter-wave =
// 256 contiguous bytes of a weight row) into a two-slot LDS ring in fragment order, and every wave reads its fragments
// with one conflict-free ds_read_b128 each.  A unit = 16 fragments = 16 KiB = the weights of 32 MFMAs for each wave of a pair:
//   units 0..3   Wq, Wk, Wv, Wo (whole matrices)                     frag = 4 (output slab) + in-feature chunk
//   units 4 + s  feed-forward pipeline step s = 0..9:  frags 0..3 / 4..7 = W1 output slab s / 8 + s (s < 8),
//                frags 8..11 / 12..15 = W2 inner chunk s - 2 / 8 + s - 2 as 4 output slabs (s >= 2)
// Piece (i, g) of a fragment sits at 16-byte position 4 i + (g ^ f(i >> 2)), f = {0, 3, 2, 1}: the b128 reads (lane
// groups {0-3, 12-15, 20-27}, ...) are conflict-free on the LDS banks, the b128 writes (8 lanes = one weight row of two
// fragments) 2-way.
// Hand-shake: barrier_k = "unit k is complete in slot k & 1, and every wave is done with unit k - 1".  After it a wave
// reads its fragments of unit k into registers, writes its share of unit k + 1 into the other slot (global loads
// requested three units earlier), requests unit k + 4, and multiplies.
constexpr int CHAIN_UNITS = 14;
__device__ __forceinline__ int frag_pos(int i, int g) { return 4 * i + (g ^ ((4 - (i >> 2)) & 3)); }

struct ChainW { const float *wq, *wk, *wv, *wo, *w1, *w2; };

// wave w (0..7), lane l: its two 16-byte pieces of unit K -> source pointer (null: no piece) and LDS float offset in the slot
template <int K>
__device__ __forceinline__ void unit_piece(const ChainW& W, int w, int l, int k2, const float*& src, int& dstoff) {
    const int p = (2 * w + k2) * 64 + l;                             // piece index 0..1023 of the unit
    if constexpr (K < 4) {
        const float* M = K == 0 ? W.wq : (K == 1 ? W.wk : (K == 2 ? W.wv : W.wo));
        const int row = p >> 4, col4 = p & 15;                       // 64 rows x 16 pieces, row-major = 16 KiB contiguous
        src = M + (long)row * 64 + 4 * col4;
        dstoff = ((row >> 4) * 4 + (col4 >> 2)) * 256 + 4 * frag_pos(row & 15, col4 & 3);
    } else {
        constexpr int S = K - 4;
        const int q = p >> 8, pp = p & 255;                          // q is wave-uniform: waves 0,1 / 2,3 / 4,5 / 6,7
        if (q < 2) {                                                 // W1 output slab S (q = 0) or 8 + S (q = 1): 16 rows x 64, contiguous
            const int row = pp >> 4, col4 = pp & 15;
            src = S < 8 ? W.w1 + (long)(16 * (8 * q + S) + row) * 64 + 4 * col4 : nullptr;
            dstoff = (4 * q + (col4 >> 2)) * 256 + 4 * frag_pos(row, col4 & 3);
        } else {                                                     // W2 inner chunk S - 2 (q = 2) or 8 + S - 2 (q = 3): 64 rows x 64 B
            const int row = pp >> 2, g = pp & 3;
            src = S >= 2 ? W.w2 + (long)row * 256 + 16 * (8 * (q - 2) + S - 2) + 4 * g : nullptr;
            dstoff = (8 + 4 * (q - 2) + (row >> 4)) * 256 + 4 * frag_pos(row & 15, g);
        }
    }
}

template <int DH, class TAILP>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
fused_chain_fwd_kernel(const FusedFwdP P_unused, const TAILP T_unused) {
#define PTYPE FusedFwdP
    constexpr bool TAIL = IsTail<TAILP>::value;
    constexpr unsigned KOFF = (unsigned)((sizeof(FusedFwdP) + 7) & ~(size_t)7);     // kernarg offset of T_unused
    constexpr int NC = DH / 16;                 // 16-feature slabs per head
    constexpr int NH = 64 / DH;
    constexpr int NQ = NH == 1 ? 4 : 2;         // Q slabs a wave forms (one head: both waves of a pair need the whole query)
    constexpr int NHW = NH == 1 ? 1 : NH / 2;   // heads per wave
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int TS = 64 * FS;
    float* sX = sm;                             // T0: x tile [token][feature]; then the pair hand-overs; TAIL: the block's output tile
    float* sD = sm + TS;                        // T1: FrequencyLayer output (dsp); TAIL: the tail's row vectors
    float* sK = sm + 2 * TS;                    // T2: K [token][feature]
    float* sVt = sm + 3 * TS;                   // T3: V^T [feature][token]
    float* sRing = sm + 4 * TS;                 // T4..T5: weight ring, 2 slots x 4096 floats (TAIL: the tail's DFT partials later)
    static_assert(2 * 4096 <= 2 * TS, "ring must fit two tiles");
    float* sTab = sm + 6 * TS;                  // FUSED_MAX_CB * 128
    float* sSpec = sTab + FUSED_MAX_CB * 128;   // FUSED_MAX_CB * 128 (tail only)
    int* sIds = reinterpret_cast<int*>(sSpec + FUSED_MAX_CB * 128);
    float* sVec = reinterpret_cast<float*>(sIds + 64);              // the block's 12 bias / gamma / beta / sqrt_beta vectors + b1: 1024 floats
    enum { V_BQ = 0, V_BK = 64, V_BV = 128, V_BO = 192, V_B2 = 256, V_FG = 320, V_FB = 384, V_AG = 448, V_AB = 512, V_FFG = 576,
           V_FFB = 640, V_BETA = 704, V_B1 = 768 };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int half = wave >> 2, T = wave & 3;   // pair T = waves T, T + 4; half 0 finishes the row passes
    const int L = KARG(FusedFwdP, L), Lp = KARG(FusedFwdP, Lp), cb = KARG(FusedFwdP, cb);
    const int b = blockIdx.x;
    const long tok0 = (long)b * L;
    const int t = 16 * T + n;
    const bool tile_on = 16 * T < L, ok = t < L;
    const long erow = (tok0 + t) * 64 + 4 * g;                       // element offset of this lane's 4 features of slab 0
    float* const trash = KARG(FusedFwdP, trash) + 4 * lane;
    // stores of padded token rows go to a trash line instead of being predicated: a predicated store is a branch, and a
    // branch ends the scheduling region
    auto dst = [&](float* base, long off) { return ok ? base + off : trash; };

    STAMP(0);
    const DropSeed dseed = drop_seed(KARG(FusedFwdP, drop_f));
    const ChainW W = {KARG(FusedFwdP, wq), KARG(FusedFwdP, wk), KARG(FusedFwdP, wv), KARG(FusedFwdP, wo), KARG(FusedFwdP, w1), KARG(FusedFwdP, w2)};
    constexpr int D = 3;                                             // units in flight in registers
    f32x4 st[D][2];
    auto issue = [&](auto kc) {
        constexpr int K = decltype(kc)::value;
        if constexpr (K < CHAIN_UNITS) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const float* src; int off;
                unit_piece<K>(W, wave, lane, k2, src, off);
                st[K % D][k2] = src ? gld4(src) : f32x4{0, 0, 0, 0};
            }
        }
    };
    auto fill = [&](auto kc) {
        constexpr int K = decltype(kc)::value;
        if constexpr (K < CHAIN_UNITS) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const float* src; int off;
                unit_piece<K>(W, wave, lane, k2, src, off);
                st4(sRing + (K & 1) * 4096 + off, st[K % D][k2]);
            }
        }
    };
    issue(std::integral_constant<int, 0>{}); issue(std::integral_constant<int, 1>{}); issue(std::integral_constant<int, 2>{});
    float vec0, vec1;
    {   // the small per-feature vectors go through LDS once: read where they are needed, a global load would expose an L2 round
        // trip on the single chain of a wave (the stamps of the first pair build: 2-3 k cycles per LayerNorm stage)
        const float* v0 = wave == 0 ? KARG(FusedFwdP, bq) : wave == 1 ? KARG(FusedFwdP, bk) : wave == 2 ? KARG(FusedFwdP, bv) : wave == 3 ? KARG(FusedFwdP, bo) :
                          wave == 4 ? KARG(FusedFwdP, b2) : wave == 5 ? KARG(FusedFwdP, f_g) : wave == 6 ? KARG(FusedFwdP, f_b) : KARG(FusedFwdP, a_g);
        const float* v1 = wave == 0 ? KARG(FusedFwdP, a_b) : wave == 1 ? KARG(FusedFwdP, ff_g) : wave == 2 ? KARG(FusedFwdP, ff_b) : wave == 3 ? KARG(FusedFwdP, sqrt_beta) :
                          KARG(FusedFwdP, b1) + 64 * (wave - 4);
        vec0 = gld(v0 + lane); vec1 = gld(v1 + lane);               // (stored to LDS after phase 0: their round trip hides behind it)
    }
    __builtin_amdgcn_sched_barrier(0);
    chain_phase0(sX, sIds, sTab, dseed, L, cb, tok0, b);
    sVec[64 * wave + lane] = vec0;
    sVec[512 + 64 * wave + lane] = vec1;
    fill(std::integral_constant<int, 0>{});
    issue(std::integral_constant<int, 3>{});
    STAMP(1);

    const float* const myfrag = sRing + 4 * frag_pos(n, g);          // this lane's piece inside fragment 0 of slot 0
    f32x4 wf[8];
    // after barrier_K: fragments f0 + {0..3} and f1 + {0..3} of unit K -> registers; this wave's share of unit K + 1 -> the
    // other slot; unit K + 4 requested.  hipcc's scheduler sinks loads to their first use: sched_barrier pins them here.
#define UNIT_BEGIN(K, f0, f1) { \
        lds_barrier(); \
        _Pragma("unroll") for (int f = 0; f < 4; ++f) { wf[f] = ld4(myfrag + ((K) & 1) * 4096 + ((f0) + f) * 256); \
                                                          wf[4 + f] = ld4(myfrag + ((K) & 1) * 4096 + ((f1) + f) * 256); } \
        fill(std::integral_constant<int, (K) + 1>{}); issue(std::integral_constant<int, (K) + 4>{}); \
        __builtin_amdgcn_sched_barrier(0); }

    // ---- FrequencyLayer (src/model/bsarec.py:90-104) in three slices beside the Q / K / V projections: both waves of a pair
    //      form the 2 cb spectrum rows of their 4 features per lane over ALL tokens (no partial sums through LDS, no
    //      barrier), wave `half` then finishes rows 16 T + 4 (2 half + {0, 1}) + g of the tile
    const int f4 = 4 * n;
    f32x4 re[FUSED_MAX_CB], im[FUSED_MAX_CB];
    auto spectrum = [&]() {
#pragma unroll
        for (int k = 0; k < FUSED_MAX_CB; ++k) { re[k] = f32x4{0, 0, 0, 0}; im[k] = re[k]; }
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int tt = 16 * g + i;
            const f32x4 xv = ld4(sX + tt * FS + f4);                 // rows >= L are zero (and so are their twiddles)
#pragma unroll
            for (int k = 0; k < FUSED_MAX_CB; ++k)
                if (k < cb) {
                    const float c = sTab[2 * (k * 64 + tt)], sn = sTab[2 * (k * 64 + tt) + 1];
                    re[k] += xv * c; im[k] -= xv * sn;
                }
        }
#pragma unroll
        for (int k = 0; k < FUSED_MAX_CB; ++k)
            if (k < cb) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { re[k][j] = quad_rows_sum(re[k][j]); im[k][j] = quad_rows_sum(im[k][j]); }
            }
    };
    auto dsp_rows = [&](int p) {
        f32x4 b2 = ld4(sVec + V_BETA + f4);
        b2 = b2 * b2;
        const f32x4 fg = ld4(sVec + V_FG + f4), fb = ld4(sVec + V_FB + f4);
        const DropP drop_f = KARG(FusedFwdP, drop_f);
        const float eps = KARG(FusedFwdP, eps);
        float* const xhat_f = KARG(FusedFwdP, xhat_f);
        float* const rstd_f = KARG(FusedFwdP, rstd_f);
        float* const dspG = KARG(FusedFwdP, dsp);
        const float invL = 1.0f / (float)L;
        const int tt = 16 * T + 4 * p + g;
        const bool rok = tt < L;
        const long e = (tok0 + tt) * 64 + f4;
        f32x4 v = {0, 0, 0, 0};
        if (rok) {
            const f32x4 xv = ld4(sX + tt * FS + f4);
            f32x4 low = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < FUSED_MAX_CB; ++k)
                if (k < cb) {
                    const float w = (k == 0 || (2 * k == L)) ? 1.0f : 2.0f;
                    const float c = sTab[2 * (k * 64 + tt)] * w, sn = sTab[2 * (k * 64 + tt) + 1] * w;
                    low += re[k] * c - im[k] * sn;
                }
            low = low * invL;
            v = (low + b2 * (xv - low)) * drop_mult4(drop_f, dseed, (uint64_t)e >> 2) + xv;
        }
        const float mean = group_sum<16>(v.x + v.y + v.z + v.w) * (1.0f / 64.0f);
        f32x4 dl = {0, 0, 0, 0};
        if (rok) dl = v - mean;
        const float var = group_sum<16>(dl.x * dl.x + dl.y * dl.y + dl.z * dl.z + dl.w * dl.w) * (1.0f / 64.0f);
        const float rs = 1.0f / sqrtf(var + eps);
        f32x4 y = {0, 0, 0, 0};
        if (rok) {
            const f32x4 xh = dl * rs;
            y = fg * xh + fb;
            gst4(xhat_f + e, xh);
            if (dspG) gst4(dspG + e, y);
            if (n == 0) gst(rstd_f + (tok0 + tt), rs);
        }
        st4(sD + tt * FS + f4, y);
    };

    // ---- Q, K, V projections: this wave's output slabs 2 half, 2 half + 1 (one attention head: also Q's other two)
    f32x4 x[4], q[NQ];
    const int o0 = 2 * half;
    {
        float* const qG = KARG(FusedFwdP, q);
        float* const kG = KARG(FusedFwdP, k);
        float* const vG = KARG(FusedFwdP, v);
        UNIT_BEGIN(0, 8 * half, 8 * half + 4)                        // ---- barrier_0 (also: x tile, ids, twiddles, vectors, ring slot 0)
        f32x4 bia[3][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bia[0][j] = ld4(sVec + V_BQ + 16 * (o0 + j) + 4 * g); bia[1][j] = ld4(sVec + V_BK + 16 * (o0 + j) + 4 * g); bia[2][j] = ld4(sVec + V_BV + 16 * (o0 + j) + 4 * g);
        }
        f32x4 bq2[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
        if (NH == 1) { bq2[0] = ld4(sVec + V_BQ + 16 * (2 - o0) + 4 * g); bq2[1] = ld4(sVec + V_BQ + 16 * (3 - o0) + 4 * g); }
#pragma unroll
        for (int c = 0; c < 4; ++c) x[c] = tile_on ? ld4(sX + t * FS + 16 * c + 4 * g) : f32x4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            f32x4 a0 = bia[u][0], a1 = bia[u][1];
            // (running the upper wave's FrequencyLayer slice BEFORE its MFMAs -- a stagger of the two waves of a SIMD -- was
            //  measured: 20.0 k -> 21.5 k cycles for this stage, not kept)
            auto slice = [&]() {
                if (!tile_on) return;
                if (u == 0) spectrum(); else dsp_rows(2 * half + (u - 1));
            };
            if (tile_on) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        a0 = mfma16(wf[c][r], x[c][r], a0);
                        a1 = mfma16(wf[4 + c][r], x[c][r], a1);
                    }
                float* const G = u == 0 ? qG : (u == 1 ? kG : vG);
                gst4(dst(G, erow + 16 * o0), a0); gst4(dst(G, erow + 16 * (o0 + 1)), a1);
            } else { a0 = f32x4{0, 0, 0, 0}; a1 = a0; }          // no token of this tile exists: its K rows / V^T columns must still be
            if (u == 0) {                                            // finite (the other pairs' MFMAs read them; their P is exactly 0)
                q[NQ == 4 ? o0 : 0] = a0; q[NQ == 4 ? o0 + 1 : 1] = a1;
                if (NH == 1 && tile_on) {                            // one head: the pair's other two Q slabs too (the whole query is this wave's B operand)
                    f32x4 wq2[8];
#pragma unroll
                    for (int f = 0; f < 8; ++f) wq2[f] = ld4(myfrag + (8 * (1 - half) + f) * 256);
                    f32x4 c0 = bq2[0], c1 = bq2[1];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { c0 = mfma16(wq2[c][r], x[c][r], c0); c1 = mfma16(wq2[4 + c][r], x[c][r], c1); }
                    q[NQ == 4 ? 2 - o0 : 0] = c0; q[NQ == 4 ? 3 - o0 : 1] = c1;
                }
            } else if (u == 1) {
                st4(sK + t * FS + 16 * o0 + 4 * g, a0); st4(sK + t * FS + 16 * (o0 + 1) + 4 * g, a1);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sVt[(16 * o0 + 4 * g + r) * FS + t] = a0[r];
                    sVt[(16 * (o0 + 1) + 4 * g + r) * FS + t] = a1[r];
                }
            }
            slice();
            if (u == 0) UNIT_BEGIN(1, 8 * half, 8 * half + 4)
            if (u == 1) UNIT_BEGIN(2, 8 * half, 8 * half + 4)
        }
    }
    STAMP(2);
    // ---- unit 3 = Wo; barrier_3 also publishes K, V^T of every tile and dsp.  This wave's dense fragments: output slab o,
    //      context chunks 2 half + {0, 1} -> wf[2 o + cc]
    lds_barrier();
#pragma unroll
    for (int o = 0; o < 4; ++o) { wf[2 * o] = ld4(myfrag + 4096 + (4 * o + 2 * half) * 256); wf[2 * o + 1] = ld4(myfrag + 4096 + (4 * o + 2 * half + 1) * 256); }
    fill(std::integral_constant<int, 4>{}); issue(std::integral_constant<int, 7>{});
    __builtin_amdgcn_sched_barrier(0);
    STAMP(3);
    const float eps = KARG(FusedFwdP, eps);
    f32x4 a[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) a[o] = f32x4{0, 0, 0, 0};
    if (tile_on) {
        // ---- attention of this wave's heads, transposed: lane = query, registers = keys      src/model/_modules.py:118-135
        const DropP drop_p = KARG(FusedFwdP, drop_p);
        float* const probsG = KARG(FusedFwdP, probs);
        float* const ctxG = KARG(FusedFwdP, ctx);
        const float inv_sqrt = 1.0f / sqrtf((float)DH);
        f32x4 ctx[2];
#pragma unroll
        for (int hh = 0; hh < NHW; ++hh) {
            const int h = NH == 1 ? 0 : half * NHW + hh;
            f32x4 s[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) s[kt] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int cc = 0; cc < NC; ++cc) {
                f32x4 kf[4];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) kf[kt] = ld4(sK + (16 * kt + n) * FS + h * DH + 16 * cc + 4 * g);
                const f32x4 qs = q[NH == 1 ? cc : hh * NC + cc];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) s[kt] = mfma16(kf[kt][r], qs[r], s[kt]);
            }
            // scale, additive mask (-10000, fp32), softmax over keys = registers x the four lanes of this query
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const int key0 = 16 * kt + 4 * g;
                const int4 i4 = *reinterpret_cast<const int4*>(sIds + key0);
                const int idk[4] = {i4.x, i4.y, i4.z, i4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {                        // (selects, not branches)
                    const int key = key0 + r;
                    const float add = (key <= t && idk[r] > 0) ? 0.0f : -10000.0f;
                    float sv = __fadd_rn(__fmul_rn(s[kt][r], inv_sqrt), add);      // scaled, THEN masked, two roundings as the reference's two ops
                    sv = key < L ? sv : -INFINITY;
                    s[kt][r] = sv;
                    mx = fmaxf(mx, sv);
                }
            }
            mx = quad_rows_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __expf(s[kt][r] - mx);          // keys past L hold -inf: exactly 0
                    s[kt][r] = e;
                    sum += e;
                }
            sum = quad_rows_sum(sum);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const int key0 = 16 * kt + 4 * g;
                const f32x4 p = s[kt] * inv;
                const bool pok = ok && key0 < Lp && (NH > 1 || half == 0);       // (one head: both waves hold the same row, one stores it)
                const long e = (((long)b * NH + h) * L + (ok ? t : 0)) * Lp + (key0 < Lp ? key0 : 0);
                gst4(pok ? probsG + e : trash, p);
                s[kt] = p * drop_mult4(drop_p, dseed, (uint64_t)e >> 2);
            }
            // ctx^T = V^T . Drop(P)^T  (the probability accumulators are the B operand as they stand): this wave's context
            // slabs = features 32 half .. 32 half + 31
            constexpr int NF = NH == 1 ? 2 : NC;                     // context slabs produced per head pass
#pragma unroll
            for (int fc = 0; fc < NF; ++fc) ctx[NH == 1 ? fc : hh * NC + fc] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {                         // (unpopulated key tiles: K rows / V^T columns are zero, P is zero)
                f32x4 vf[NF];
#pragma unroll
                for (int fc = 0; fc < NF; ++fc) vf[fc] = ld4(sVt + ((NH == 1 ? 32 * half : h * DH) + 16 * fc + n) * FS + 16 * kt + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int fc = 0; fc < NF; ++fc) ctx[NH == 1 ? fc : hh * NC + fc] = mfma16(vf[fc][r], s[kt][r], ctx[NH == 1 ? fc : hh * NC + fc]);
            }
        }
        gst4(dst(ctxG, erow + 16 * o0), ctx[0]); gst4(dst(ctxG, erow + 16 * (o0 + 1)), ctx[1]);
        STAMP(4);
        // ---- dense, K split inside the pair: this wave's two context slabs against all four output slabs
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int o = 0; o < 4; ++o) a[o] = mfma16(wf[2 * o + cc][r], ctx[cc][r], a[o]);
        if (half == 1) {
#pragma unroll
            for (int o = 0; o < 4; ++o) st4(sX + t * FS + 16 * o + 4 * g, a[o]);       // (the x tile is dead: every wave holds its slabs)
        }
    }
    lds_barrier();                                                   // ---- hand-over 1: the upper wave's dense partial sums
    f32x4 hm[4], y[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { y[c] = f32x4{0, 0, 0, 0}; hm[c] = y[c]; }
    if (tile_on && half == 0) {
        // ---- + bias, dropout, residual, LayerNorm, alpha mix -> hmix (to the pair's other wave through LDS)
        const DropP drop_o = KARG(FusedFwdP, drop_o);
        f32x4 v[4], xh[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const f32x4 part = ld4(sX + t * FS + 16 * o + 4 * g) + ld4(sVec + V_BO + 16 * o + 4 * g);
            v[o] = (a[o] + part) * drop_mult4(drop_o, dseed, (uint64_t)(erow + 16 * o) >> 2) + x[o];
        }
        float rs;
        ln_slabs(v, eps, xh, rs);
        const float alpha = KARG(FusedFwdP, alpha), oma = KARG(FusedFwdP, oma);
        float* const xhat_a = KARG(FusedFwdP, xhat_a);
        float* const hmixG = KARG(FusedFwdP, hmix);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const f32x4 ga = ld4(sVec + V_AG + 16 * o + 4 * g), ba = ld4(sVec + V_AB + 16 * o + 4 * g);
            hm[o] = alpha * ld4(sD + t * FS + 16 * o + 4 * g) + oma * (ga * xh[o] + ba);
            gst4(dst(xhat_a, erow + 16 * o), xh[o]); gst4(dst(hmixG, erow + 16 * o), hm[o]);
            st4(sX + t * FS + 16 * o + 4 * g, hm[o]);
        }
        gst((ok && g == 0) ? KARG(FusedFwdP, rstd_a) + tok0 + t : trash, rs);
    }
    STAMP(5);
    // ---- feed-forward: this wave's 8 inner slabs j = 8 half + s as a 3-stage pipeline, one weight unit per step s = 0..9:
    //        MFMA   dense_1 slab 8 half + s  (16, one chain; s < 8)          u_s = b1 + W1[slab] . hmix^T
    //        MFMA   dense_2 inner chunk 8 half + s - 2  (16, four chains)     y  += W2[:, chunk] . gelu(u_{s-2})
    //        VALU   erf-GELU + GELU' on the accumulator of step s - 1, both stored for the backward  (1 <= s <= 8)
    {
        float* const uG = KARG(FusedFwdP, u);
        float* const gpG = KARG(FusedFwdP, gp);
        float* const udst = dst(uG, (tok0 + t) * 256 + 128 * half + 4 * g);
        float* const gdst = dst(gpG, (tok0 + t) * 256 + 128 * half + 4 * g);
        const long ustep = ok ? 16 : 0;
        if (half == 0) {
#pragma unroll
            for (int o = 0; o < 4; ++o) y[o] = ld4(sVec + V_B2 + 16 * o + 4 * g);
        }
        f32x4 ucur = {0, 0, 0, 0}, glprev = {0, 0, 0, 0};
        f32x4 bnext = ld4(sVec + V_B1 + 128 * half + 4 * g);
#pragma unroll
        for (int s = 0; s < 10; ++s) {
            // barrier_{4+s}; at s = 0 it is also hand-over 2: hmix of the tile is in LDS
            lds_barrier();
#pragma unroll
            for (int f = 0; f < 4; ++f) { wf[f] = ld4(myfrag + (s & 1) * 4096 + (4 * half + f) * 256); wf[4 + f] = ld4(myfrag + (s & 1) * 4096 + (8 + 4 * half + f) * 256); }
            if (s == 0) { fill(std::integral_constant<int, 5>{}); issue(std::integral_constant<int, 8>{}); }
            if (s == 1) { fill(std::integral_constant<int, 6>{}); issue(std::integral_constant<int, 9>{}); }
            if (s == 2) { fill(std::integral_constant<int, 7>{}); issue(std::integral_constant<int, 10>{}); }
            if (s == 3) { fill(std::integral_constant<int, 8>{}); issue(std::integral_constant<int, 11>{}); }
            if (s == 4) { fill(std::integral_constant<int, 9>{}); issue(std::integral_constant<int, 12>{}); }
            if (s == 5) { fill(std::integral_constant<int, 10>{}); issue(std::integral_constant<int, 13>{}); }
            if (s == 6) fill(std::integral_constant<int, 11>{});
            if (s == 7) fill(std::integral_constant<int, 12>{});
            if (s == 8) fill(std::integral_constant<int, 13>{});
            if (s == 0 && half == 1 && tile_on) {
#pragma unroll
                for (int o = 0; o < 4; ++o) hm[o] = ld4(sX + t * FS + 16 * o + 4 * g);
            }
            f32x4 unext = bnext;
            if (s + 1 < 8) bnext = ld4(sVec + V_B1 + 128 * half + 16 * (s + 1) + 4 * g);
            __builtin_amdgcn_sched_barrier(0);
            if (tile_on) {
                float gq[4] = {0.f, 0.f, 0.f, 0.f}, gpq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (s >= 1 && s <= 8) gelu_both(ucur[r], gq[r], gpq[r]);
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        if (s >= 2) y[o] = mfma16(wf[4 + o][r], glprev[r], y[o]);
                        if (s < 8) unext = mfma16(wf[o][r], hm[o][r], unext);          // (chunk c = o of the slab)
                    }
                }
                const f32x4 gl = {gq[0], gq[1], gq[2], gq[3]}, gd = {gpq[0], gpq[1], gpq[2], gpq[3]};
                if (s >= 1 && s <= 8) { gst4(udst + ustep * (s - 1), gl); gst4(gdst + ustep * (s - 1), gd); }
                glprev = gl;
                ucur = unext;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    STAMP(6);
    // TAIL: the top block's weight fragments are requested here, in the shadow of the hand-over and the final row pass (lane n
    // reads ITS weight row: ~50 address-path cycles per wave instruction, the wave cannot move on before they are issued)
    TopFwdRegs<false> TR;
    if constexpr (TAIL) { if (half == 0) top_fwd_prefetch<false, KOFF, true>(TR); }
    if (half == 1) {
        if (tile_on) {
#pragma unroll
            for (int o = 0; o < 4; ++o) st4(sX + t * FS + 16 * o + 4 * g, y[o]);
        }
        { long long* st_ = KARG(FusedFwdP, stamps); if (st_ && blockIdx.x == 0 && threadIdx.x == 256) st_[9] = clock64(); }
        lds_barrier();                                               // ---- hand-over 3 (upper wave's side): its feed-forward partial sums
        return;
    }
    lds_barrier();                                                   // ---- hand-over 3
    if (tile_on) {
        // ---- dropout + residual + LayerNorm -> block output
        const DropP drop_ff = KARG(FusedFwdP, drop_ff);
        f32x4 v[4], xh[4];
#pragma unroll
        for (int o = 0; o < 4; ++o)
            v[o] = (y[o] + ld4(sX + t * FS + 16 * o + 4 * g)) * drop_mult4(drop_ff, dseed, (uint64_t)(erow + 16 * o) >> 2) + hm[o];
        float rs;
        ln_slabs(v, eps, xh, rs);
        float* const xhat_ff = KARG(FusedFwdP, xhat_ff);
        float* const Xout = KARG(FusedFwdP, Xout);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const f32x4 gf = ld4(sVec + V_FFG + 16 * o + 4 * g), bf = ld4(sVec + V_FFB + 16 * o + 4 * g);
            y[o] = ok ? gf * xh[o] + bf : f32x4{0, 0, 0, 0};
            gst4(dst(xhat_ff, erow + 16 * o), xh[o]); gst4(dst(Xout, erow + 16 * o), y[o]);
        }
        gst((ok && g == 0) ? KARG(FusedFwdP, rstd_ff) + tok0 + t : trash, rs);
    } else {
#pragma unroll
        for (int o = 0; o < 4; ++o) y[o] = f32x4{0, 0, 0, 0};
    }
    STAMP(7);
    if constexpr (TAIL) {
        // the one-row top block as this kernel's tail (fused_top.h): its x tile = this block's output, rows >= L zero
#pragma unroll
        for (int o = 0; o < 4; ++o) st4(sX + t * FS + 16 * o + 4 * g, y[o]);
        lds_barrier();                                               // ---- (waves 0..3 only: waves 4..7 have exited)
        STAMP(8);
        top_fwd_rest<DH, false, KOFF, false>(TR, dseed, sX, sK, sVt, sRing, sTab, sSpec, sD, sIds);
    }
#undef UNIT_BEGIN
}
#undef PTYPE

static inline size_t fused_chain_fwd_smem_bytes(bool tail) {
    (void)tail;
    return (size_t)(6 * 64 * FS + 2 * FUSED_MAX_CB * 128 + 64 + 1024) * 4;
}
