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
// the x tile, K, V^T (attention mixes tokens) and the FrequencyLayer output.  TWO workgroup barriers instead of ~25;
// waves 0..3 = the four token tiles ("owners", one per SIMD), waves 4..7 = the FrequencyLayer of the same four tiles
// (pruned DFT: every helper wave forms the 2 cb spectrum rows of ITS 4 features per lane over all tokens itself -- no
// partial sums through LDS, no barrier) and exit.  Between the two barriers an owner never waits for another wave: its
// vector work (GELU, Philox, LayerNorm) issues in the shadow of its own dependent MFMA chains.
// Same arithmetic (fp32 MFMA = fmaf chains), same Philox stream, same saved tensors as fused_layer_fwd_kernel; parity
// against the oracle by the same tests (tests/test_gpu_parity.py runs both kernel sets).
// src/model/bsarec.py:56-104, src/model/_modules.py:22-140.
#pragma once
#include "fused_layer.h"

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// sum / max over the four lanes n, n + 16, n + 32, n + 48 (v_permlane16_swap: odd rows of the first operand <-> even rows of
// the second; v_permlane32_swap: upper half of the first <-> lower half of the second; see halves_of in common.h)
__device__ __forceinline__ void rows_of(float v, float& a_, float& b_) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    a_ = __builtin_bit_cast(float, a); b_ = __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float quad_rows_sum(float v) { float a, b; rows_of(v, a, b); return xor32_sum(a + b); }
__device__ __forceinline__ float quad_rows_max(float v) { float a, b; rows_of(v, a, b); return xor32_max(fmaxf(a, b)); }

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
// build of this kernel did: ~64 address-path cycles per wave instruction, x 4 owner waves -- the stamps showed every stage
// 2.2-3x over its MFMA time).  So waves 4..7 stream the block's 196 KB of weights ONCE per workgroup with coalesced loads
// (a quarter-wave = 256 contiguous bytes) into a two-slot LDS ring in fragment order, and the owners read their fragments
// with one conflict-free ds_read_b128 each.  A unit = 8 fragments = 8 KiB = the weights of 32 MFMAs per owner:
//   units 0..5   Wq, Wk, Wv as pairs of output slabs (32 rows x 64)         frag = 4 (slab in pair) + in-feature chunk
//   units 6..7   Wo, the same way
//   units 8 + s  feed-forward pipeline step s = 0..17: frags 0..3 = W1 output slab s (s < 16), frags 4..7 = W2 inner
//                chunk s - 2 as 4 output slabs (s >= 2)
// Piece (i, g) of a fragment sits at 16-byte position 4 i + (g ^ f(i >> 2)), f = {0, 3, 2, 1}: the owners' b128 reads
// (lane groups {0-3, 12-15, 20-27}, ...) and the loaders' b128 writes (8 lanes = one weight row of two fragments) are
// both at most 2-way on the LDS banks (reads: conflict-free).
// Hand-shake: ONE workgroup barrier per unit.  Barrier E_k: unit k is complete in slot k & 1.  Loaders write unit k
// between E_{k-1} and E_k; owners read unit k into registers between E_k and E_{k+1} and multiply with it after E_{k+1}
// (second register set), so slot k & 1 is free again when the loaders pass E_{k+1}.
constexpr int CHAIN_UNITS = 26;
__device__ __forceinline__ int frag_pos(int i, int g) { return 4 * i + (g ^ ((4 - (i >> 2)) & 3)); }

struct ChainW { const float *wq, *wk, *wv, *wo, *w1, *w2; };

// loader wave lw (0..3), lane l: its two 16-byte pieces of unit K -> source pointer (null: no piece) and LDS float offset in the slot
template <int K>
__device__ __forceinline__ void unit_piece(const ChainW& W, int lw, int l, int k2, const float*& src, int& dstoff) {
    const int p = (2 * lw + k2) * 64 + l;                            // piece index 0..511 of the unit
    if constexpr (K < 8) {
        const float* M = K < 2 ? W.wq : (K < 4 ? W.wk : (K < 6 ? W.wv : W.wo));
        const int row = p >> 4, col4 = p & 15;                       // 32 rows x 16 pieces, row-major = 8 KiB contiguous
        src = M + (long)(32 * (K & 1) + row) * 64 + 4 * col4;
        dstoff = ((row >> 4) * 4 + (col4 >> 2)) * 256 + 4 * frag_pos(row & 15, col4 & 3);
    } else {
        constexpr int S = K - 8;
        if (p < 256) {                                               // W1 output slab S: 16 rows x 64 in-features, contiguous
            const int row = p >> 4, col4 = p & 15;
            src = S < 16 ? W.w1 + (long)(16 * S + row) * 64 + 4 * col4 : nullptr;
            dstoff = (col4 >> 2) * 256 + 4 * frag_pos(row, col4 & 3);
        } else {                                                     // W2 inner chunk S - 2: 64 rows x 16 in-features (64 B per row)
            const int q = p - 256, row = q >> 2, g = q & 3;
            src = S >= 2 ? W.w2 + (long)row * 256 + 16 * (S - 2) + 4 * g : nullptr;
            dstoff = (4 + (row >> 4)) * 256 + 4 * frag_pos(row & 15, g);
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
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int TS = 64 * FS;
    float* sX = sm;                             // T0: x tile [token][feature]; TAIL: the block's output tile at the end
    float* sD = sm + TS;                        // T1: FrequencyLayer output (dsp); TAIL: the tail's row vectors
    float* sK = sm + 2 * TS;                    // T2: K [token][feature]
    float* sVt = sm + 3 * TS;                   // T3: V^T [feature][token]
    float* sRing = sm + 4 * TS;                 // T4: weight ring, 2 slots x 2048 floats (TAIL: T4..T5 = the tail's DFT partials later)
    constexpr int NT = TAIL ? 6 : 5;
    static_assert(2 * 2048 <= TS, "ring must fit one tile");
    float* sTab = sm + NT * TS;                 // FUSED_MAX_CB * 128
    float* sSpec = sTab + FUSED_MAX_CB * 128;   // FUSED_MAX_CB * 128 (tail only)
    int* sIds = reinterpret_cast<int*>(sSpec + FUSED_MAX_CB * 128);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int L = KARG(FusedFwdP, L), Lp = KARG(FusedFwdP, Lp), cb = KARG(FusedFwdP, cb);
    const int b = blockIdx.x;
    const long tok0 = (long)b * L;

    STAMP(0);
    const DropSeed dseed = drop_seed(KARG(FusedFwdP, drop_f));

    if (wave >= 4) {
        // ================= loaders (waves 4..7): weight stream + FrequencyLayer of token tile T =================
        const int lw = wave - 4, T = lw;
        const ChainW W = {KARG(FusedFwdP, wq), KARG(FusedFwdP, wk), KARG(FusedFwdP, wv), KARG(FusedFwdP, wo), KARG(FusedFwdP, w1), KARG(FusedFwdP, w2)};
        constexpr int D = 3;                                         // units in flight in registers
        f32x4 st[D][2];
        auto issue = [&](auto kc) {
            constexpr int K = decltype(kc)::value;
            if constexpr (K < CHAIN_UNITS) {
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const float* src; int off;
                    unit_piece<K>(W, lw, lane, k2, src, off);
                    st[K % D][k2] = src ? gld4(src) : f32x4{0, 0, 0, 0};
                }
            }
        };
        auto fill = [&](auto kc) {
            constexpr int K = decltype(kc)::value;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const float* src; int off;
                unit_piece<K>(W, lw, lane, k2, src, off);
                st4(sRing + (K & 1) * 2048 + off, st[K % D][k2]);
            }
        };
        issue(std::integral_constant<int, 0>{}); issue(std::integral_constant<int, 1>{}); issue(std::integral_constant<int, 2>{});
        __builtin_amdgcn_sched_barrier(0);
        chain_phase0(sX, sIds, sTab, dseed, L, cb, tok0, b);
        lds_barrier();                                               // ---- B0: x tile, ids, twiddles
        // FrequencyLayer (src/model/bsarec.py:90-104) in slices between the unit barriers: spectrum, then 4 rows per slice
        const int f4 = 4 * n;                                        // this lane's 4 features; g = row quarter
        const bool tile_on = 16 * T < L;
        f32x4 re[FUSED_MAX_CB], im[FUSED_MAX_CB];
        auto spectrum = [&]() {
#pragma unroll
            for (int k = 0; k < FUSED_MAX_CB; ++k) { re[k] = f32x4{0, 0, 0, 0}; im[k] = re[k]; }
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const int t = 16 * g + i;
                const f32x4 xv = ld4(sX + t * FS + f4);              // rows >= L are zero (and so are their twiddles)
#pragma unroll
                for (int k = 0; k < FUSED_MAX_CB; ++k)
                    if (k < cb) {
                        const float c = sTab[2 * (k * 64 + t)], sn = sTab[2 * (k * 64 + t) + 1];
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
            const float* const sqrt_beta = KARG(FusedFwdP, sqrt_beta);
            f32x4 b2 = gld4(sqrt_beta + f4);
            b2 = b2 * b2;
            const f32x4 fg = gld4(KARG(FusedFwdP, f_g) + f4), fb = gld4(KARG(FusedFwdP, f_b) + f4);
            const DropP drop_f = KARG(FusedFwdP, drop_f);
            const float eps = KARG(FusedFwdP, eps);
            float* const xhat_f = KARG(FusedFwdP, xhat_f);
            float* const rstd_f = KARG(FusedFwdP, rstd_f);
            float* const dspG = KARG(FusedFwdP, dsp);
            const float invL = 1.0f / (float)L;
            const int t = 16 * T + 4 * p + g;
            const bool ok = t < L;
            const long e = (tok0 + t) * 64 + f4;
            f32x4 v = {0, 0, 0, 0};
            if (ok) {
                const f32x4 xv = ld4(sX + t * FS + f4);
                f32x4 low = {0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < FUSED_MAX_CB; ++k)
                    if (k < cb) {
                        const float w = (k == 0 || (2 * k == L)) ? 1.0f : 2.0f;
                        const float c = sTab[2 * (k * 64 + t)] * w, sn = sTab[2 * (k * 64 + t) + 1] * w;
                        low += re[k] * c - im[k] * sn;
                    }
                low = low * invL;
                v = (low + b2 * (xv - low)) * drop_mult4(drop_f, dseed, (uint64_t)e >> 2) + xv;
            }
            const float mean = group_sum<16>(v.x + v.y + v.z + v.w) * (1.0f / 64.0f);
            f32x4 dl = {0, 0, 0, 0};
            if (ok) dl = v - mean;
            const float var = group_sum<16>(dl.x * dl.x + dl.y * dl.y + dl.z * dl.z + dl.w * dl.w) * (1.0f / 64.0f);
            const float rs = 1.0f / sqrtf(var + eps);
            f32x4 y = {0, 0, 0, 0};
            if (ok) {
                const f32x4 xh = dl * rs;
                y = fg * xh + fb;
                gst4(xhat_f + e, xh);
                if (dspG) gst4(dspG + e, y);
                if (n == 0) gst(rstd_f + (tok0 + t), rs);
            }
            st4(sD + t * FS + f4, y);
        };
        // unit loop: write unit K to its slot, request unit K + D, one slice of the FrequencyLayer, barrier E_K
#define LOADER_STEP(K, EXTRA) { fill(std::integral_constant<int, K>{}); issue(std::integral_constant<int, K + D>{}); EXTRA; lds_barrier(); }
        LOADER_STEP(0, )
        LOADER_STEP(1, if (tile_on) spectrum())
        LOADER_STEP(2, if (tile_on) dsp_rows(0))
        LOADER_STEP(3, if (tile_on) dsp_rows(1))
        LOADER_STEP(4, if (tile_on) dsp_rows(2))
        LOADER_STEP(5, if (tile_on) dsp_rows(3))
        LOADER_STEP(6, { long long* st_ = KARG(FusedFwdP, stamps); if (st_ && blockIdx.x == 0 && threadIdx.x == 256) st_[9] = clock64(); })
        LOADER_STEP(7, )                                             // E_7 = the K / V^T / dsp exchange barrier of the owners
        LOADER_STEP(8, ) LOADER_STEP(9, ) LOADER_STEP(10, ) LOADER_STEP(11, ) LOADER_STEP(12, ) LOADER_STEP(13, )
        LOADER_STEP(14, ) LOADER_STEP(15, ) LOADER_STEP(16, ) LOADER_STEP(17, ) LOADER_STEP(18, ) LOADER_STEP(19, )
        LOADER_STEP(20, ) LOADER_STEP(21, ) LOADER_STEP(22, ) LOADER_STEP(23, ) LOADER_STEP(24, ) LOADER_STEP(25, )
#undef LOADER_STEP
        return;
    }

    // ================= owners (waves 0..3): token tile T through attention branch + feed-forward, in registers =================
    // Scheduling notes.  (1) hipcc's machine scheduler sinks loads down to their first use (fewer live registers, no
    // prefetch left) -- __builtin_amdgcn_sched_barrier(0) behind each fragment read pins it.  (2) Stores of padded token rows
    // go to a trash line instead of being predicated: a predicated store is a branch, and a branch ends the scheduling
    // region (MFMA / VALU interleaving stops at it).
    chain_phase0(sX, sIds, sTab, dseed, L, cb, tok0, b);
    lds_barrier();                                                   // ---- B0: x tile, ids, twiddles
    STAMP(1);
    const int T = wave, t = 16 * T + n;
    const bool tile_on = 16 * T < L, ok = t < L;
    const long erow = (tok0 + t) * 64 + 4 * g;                       // element offset of this lane's 4 features of slab 0
    float* const trash = KARG(FusedFwdP, trash) + 4 * lane;
    auto dst = [&](float* base, long off) { return ok ? base + off : trash; };
    const float* const myfrag = sRing + 4 * frag_pos(n, g);          // this lane's piece inside fragment 0 of slot 0
    f32x4 ws[2][8];                                                  // two register sets of 8 fragments
    auto rd_unit = [&](int k, f32x4 (&wd)[8]) {
#pragma unroll
        for (int f = 0; f < 8; ++f) wd[f] = ld4(myfrag + (k & 1) * 2048 + f * 256);
    };
    f32x4 x[4], q[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { x[c] = f32x4{0, 0, 0, 0}; q[c] = x[c]; }
    if (tile_on) {
#pragma unroll
        for (int c = 0; c < 4; ++c) x[c] = ld4(sX + t * FS + 16 * c + 4 * g);
    }
    // ---- Q, K, V projections: unit u = output slabs 2 (u & 1), +1 of [Q | K | V]: two independent accumulator chains
    {
        const float* const bq = KARG(FusedFwdP, bq);
        const float* const bk = KARG(FusedFwdP, bk);
        const float* const bv = KARG(FusedFwdP, bv);
        float* const qG = KARG(FusedFwdP, q);
        float* const kG = KARG(FusedFwdP, k);
        float* const vG = KARG(FusedFwdP, v);
        lds_barrier();                                               // ---- E_0
        rd_unit(0, ws[0]);
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            lds_barrier();                                           // ---- E_{u+1}
            rd_unit(u + 1, ws[(u + 1) & 1]);
            const float* B = u < 2 ? bq : (u < 4 ? bk : bv);
            const int o0 = 2 * (u & 1);
            f32x4 a0 = gld4(B + 16 * o0 + 4 * g), a1 = gld4(B + 16 * (o0 + 1) + 4 * g);
            __builtin_amdgcn_sched_barrier(0);
            if (tile_on) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        a0 = mfma16(ws[u & 1][c][r], x[c][r], a0);
                        a1 = mfma16(ws[u & 1][4 + c][r], x[c][r], a1);
                    }
                float* const G = u < 2 ? qG : (u < 4 ? kG : vG);
                gst4(dst(G, erow + 16 * o0), a0); gst4(dst(G, erow + 16 * (o0 + 1)), a1);
            } else { a0 = f32x4{0, 0, 0, 0}; a1 = a0; }          // no token of this tile exists: K rows / V^T columns must still be
            if (u < 2) {                                             // finite (the other owners' MFMAs read them; their P is exactly 0)
                q[o0] = a0; q[o0 + 1] = a1;
            } else if (u < 4) {
                st4(sK + t * FS + 16 * o0 + 4 * g, a0); st4(sK + t * FS + 16 * (o0 + 1) + 4 * g, a1);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sVt[(16 * o0 + 4 * g + r) * FS + t] = a0[r];
                    sVt[(16 * (o0 + 1) + 4 * g + r) * FS + t] = a1[r];
                }
            }
        }
    }
    STAMP(2);
    lds_barrier();                                                   // ---- E_7: Wo's second half; K, V^T of every tile and dsp are in LDS
    rd_unit(7, ws[1]);                                               // (ws[0] holds unit 6 = Wo output slabs 0, 1)
    __builtin_amdgcn_sched_barrier(0);
    STAMP(3);
    f32x4 y[4], hm[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { y[c] = f32x4{0, 0, 0, 0}; hm[c] = y[c]; }
    const float eps = KARG(FusedFwdP, eps);
    if (tile_on) {
        // ---- attention, transposed: lane = query, registers = keys            src/model/_modules.py:118-135
        const DropP drop_p = KARG(FusedFwdP, drop_p);
        float* const probsG = KARG(FusedFwdP, probs);
        float* const ctxG = KARG(FusedFwdP, ctx);
        const float inv_sqrt = 1.0f / sqrtf((float)DH);
        f32x4 ctx[4];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            f32x4 s[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) s[kt] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int cc = 0; cc < NC; ++cc) {
                f32x4 kf[4];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) kf[kt] = ld4(sK + (16 * kt + n) * FS + h * DH + 16 * cc + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) s[kt] = mfma16(kf[kt][r], q[h * NC + cc][r], s[kt]);
            }
            // scale, additive mask (-10000, fp32), softmax over keys = registers x the four lanes of this query
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const int key0 = 16 * kt + 4 * g;
                const int4 i4 = *reinterpret_cast<const int4*>(sIds + key0);
                const int idk[4] = {i4.x, i4.y, i4.z, i4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {                        // (selects, not branches: a branch would end the scheduling region)
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
                const bool pok = ok && key0 < Lp;                    // (elsewhere p is 0 or the row is not stored: any mask will do)
                const long e = (((long)b * NH + h) * L + (ok ? t : 0)) * Lp + (key0 < Lp ? key0 : 0);
                gst4(pok ? probsG + e : trash, p);
                s[kt] = p * drop_mult4(drop_p, dseed, (uint64_t)e >> 2);
            }
            // ctx^T = V^T . Drop(P)^T  (the probability accumulators are the B operand as they stand)
#pragma unroll
            for (int fc = 0; fc < NC; ++fc) ctx[h * NC + fc] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {                         // (unpopulated key tiles: K rows / V^T columns are zero, P is zero)
                f32x4 vf[NC];
#pragma unroll
                for (int fc = 0; fc < NC; ++fc) vf[fc] = ld4(sVt + (h * DH + 16 * fc + n) * FS + 16 * kt + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int fc = 0; fc < NC; ++fc) ctx[h * NC + fc] = mfma16(vf[fc][r], s[kt][r], ctx[h * NC + fc]);
            }
#pragma unroll
            for (int fc = 0; fc < NC; ++fc) gst4(dst(ctxG, erow + 16 * (h * NC + fc)), ctx[h * NC + fc]);
        }
        STAMP(4);
        // ---- dense (units 6, 7 in the two register sets) + dropout + residual + LayerNorm + alpha mix
        f32x4 a[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) a[o] = gld4(KARG(FusedFwdP, bo) + 16 * o + 4 * g);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int o = 0; o < 4; ++o) a[o] = mfma16(ws[o >> 1][4 * (o & 1) + c][r], ctx[c][r], a[o]);
        const DropP drop_o = KARG(FusedFwdP, drop_o);
        {
            f32x4 v[4], xh[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) v[o] = a[o] * drop_mult4(drop_o, dseed, (uint64_t)(erow + 16 * o) >> 2) + x[o];
            float rs;
            ln_slabs(v, eps, xh, rs);
            const float alpha = KARG(FusedFwdP, alpha), oma = KARG(FusedFwdP, oma);
            float* const xhat_a = KARG(FusedFwdP, xhat_a);
            float* const hmixG = KARG(FusedFwdP, hmix);
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const f32x4 ga = gld4(KARG(FusedFwdP, a_g) + 16 * o + 4 * g), ba = gld4(KARG(FusedFwdP, a_b) + 16 * o + 4 * g);
                hm[o] = alpha * ld4(sD + t * FS + 16 * o + 4 * g) + oma * (ga * xh[o] + ba);
                gst4(dst(xhat_a, erow + 16 * o), xh[o]); gst4(dst(hmixG, erow + 16 * o), hm[o]);
            }
            gst((ok && g == 0) ? KARG(FusedFwdP, rstd_a) + tok0 + t : trash, rs);
        }
    }
    STAMP(5);
    // ---- feed-forward as a 3-stage pipeline over the 16 inner slabs, one weight unit per step s = 0..17:
    //        MFMA   dense_1 slab s  (16, one chain; s < 16)              u_s = b1 + W1[slab s] . hmix^T
    //        MFMA   dense_2 inner chunk s - 2  (16, four chains; s >= 2)  y  += W2[:, chunk s-2] . gelu(u_{s-2})
    //        VALU   erf-GELU + GELU' on the accumulator of slab s - 1, both stored for the backward  (1 <= s <= 16)
    //      so the ~200 vector instructions of a slab issue in the shadow of 32 MFMAs that do not wait for them.
    {
        float* const uG = KARG(FusedFwdP, u);
        float* const gpG = KARG(FusedFwdP, gp);
        const float* const b1 = KARG(FusedFwdP, b1);
        float* const udst = dst(uG, (tok0 + t) * 256 + 4 * g);
        float* const gdst = dst(gpG, (tok0 + t) * 256 + 4 * g);
        const long ustep = ok ? 16 : 0;
#pragma unroll
        for (int o = 0; o < 4; ++o) y[o] = gld4(KARG(FusedFwdP, b2) + 16 * o + 4 * g);
        lds_barrier();                                               // ---- E_8
        rd_unit(8, ws[0]);
        f32x4 ucur = {0, 0, 0, 0}, glprev = {0, 0, 0, 0};
        f32x4 bnext = gld4(b1 + 4 * g);
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            if (s + 1 < 18) { lds_barrier(); rd_unit(8 + s + 1, ws[(s + 1) & 1]); }      // ---- E_{9+s}
            f32x4 unext = bnext;
            if (s + 1 < 16) bnext = gld4(b1 + 16 * (s + 1) + 4 * g);
            __builtin_amdgcn_sched_barrier(0);
            if (tile_on) {
                float gq[4] = {0.f, 0.f, 0.f, 0.f}, gpq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (s >= 1 && s <= 16) gelu_both(ucur[r], gq[r], gpq[r]);
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        if (s >= 2) y[o] = mfma16(ws[s & 1][4 + o][r], glprev[r], y[o]);
                        if (s < 16) unext = mfma16(ws[s & 1][o][r], hm[o][r], unext);      // (chunk c = o of slab s)
                    }
                }
                const f32x4 gl = {gq[0], gq[1], gq[2], gq[3]}, gd = {gpq[0], gpq[1], gpq[2], gpq[3]};
                if (s >= 1 && s <= 16) { gst4(udst + ustep * (s - 1), gl); gst4(gdst + ustep * (s - 1), gd); }
                glprev = gl;
                ucur = unext;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    STAMP(6);
    if (tile_on) {
        // ---- dropout + residual + LayerNorm -> block output
        const DropP drop_ff = KARG(FusedFwdP, drop_ff);
        f32x4 v[4], xh[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) v[o] = y[o] * drop_mult4(drop_ff, dseed, (uint64_t)(erow + 16 * o) >> 2) + hm[o];
        float rs;
        ln_slabs(v, eps, xh, rs);
        float* const xhat_ff = KARG(FusedFwdP, xhat_ff);
        float* const Xout = KARG(FusedFwdP, Xout);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const f32x4 gf = gld4(KARG(FusedFwdP, ff_g) + 16 * o + 4 * g), bf = gld4(KARG(FusedFwdP, ff_b) + 16 * o + 4 * g);
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
        TopFwdRegs<false> TR;
        top_fwd_prefetch<false, KOFF>(TR);
#pragma unroll
        for (int o = 0; o < 4; ++o) st4(sX + t * FS + 16 * o + 4 * g, y[o]);
        lds_barrier();                                               // ---- (owners only: waves 4..7 have exited)
        STAMP(8);
        top_fwd_rest<DH, false, KOFF>(TR, dseed, sX, sK, sVt, sm + 4 * TS, sTab, sSpec, sD, sIds);
    }
}
#undef PTYPE

static inline size_t fused_chain_fwd_smem_bytes(bool tail) {
    return (size_t)((tail ? 6 : 5) * 64 * FS + 2 * FUSED_MAX_CB * 128 + 64) * 4;
}
