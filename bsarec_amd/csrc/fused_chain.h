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

// 16 bytes of weight row (16 o + i) at in-features 16 c + 4 g .. +3 (ldw = row length in floats)
__device__ __forceinline__ f32x4 wfrag(const float* __restrict__ W, int ldw, int o, int c, int i, int g) {
    return gld4(W + (long)(16 * o + i) * ldw + 16 * c + 4 * g);
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
    constexpr int NT = TAIL ? 6 : 4;            // TAIL: T4..T5 = the tail's DFT partials
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
    chain_phase0(sX, sIds, sTab, dseed, L, cb, tok0, b);
    lds_barrier();                                                   // ---- B0: x tile, ids, twiddles
    STAMP(1);

    if (wave >= 4) {
        // ================= helpers: FrequencyLayer of token tile T (src/model/bsarec.py:90-104) =================
        const int T = wave - 4;
        if (16 * T < L) {
            const int f4 = 4 * n;                                    // this lane's 4 features; g = row quarter
            f32x4 re[FUSED_MAX_CB], im[FUSED_MAX_CB];
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
#pragma unroll
            for (int p = 0; p < 4; ++p) {
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
            }
        }
        lds_barrier();                                               // ---- B1 (helpers' side): dsp complete
        return;
    }

    // ================= owners: token tile T through attention branch + feed-forward, in registers =================
    // Scheduling notes.  (1) Weight fragments are requested one unit of work ahead into a second register set; hipcc's
    // machine scheduler sinks such loads down to their first use (fewer live registers, no prefetch left: the first build
    // of this kernel waited vmcnt(0) in front of every MFMA group) -- __builtin_amdgcn_sched_barrier(0) behind each
    // request pins it.  (2) Stores of padded token rows go to a trash line instead of being predicated: a predicated store
    // is a branch, and a branch ends the scheduling region (MFMA / VALU interleaving stops at it).
    const int T = wave, t = 16 * T + n;
    const bool tile_on = 16 * T < L, ok = t < L;
    const long erow = (tok0 + t) * 64 + 4 * g;                       // element offset of this lane's 4 features of slab 0
    float* const trash = KARG(FusedFwdP, trash) + 4 * lane;
    auto dst = [&](float* base, long off) { return ok ? base + off : trash; };
    f32x4 x[4], q[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { x[c] = f32x4{0, 0, 0, 0}; q[c] = x[c]; }
    if (tile_on) {
        // ---- Q, K, V projections: 12 output slabs, two at a time (two independent accumulator chains), weights one pair ahead
        const float* const Wq = KARG(FusedFwdP, wq);
        const float* const Wk = KARG(FusedFwdP, wk);
        const float* const Wv = KARG(FusedFwdP, wv);
        const float* const bq = KARG(FusedFwdP, bq);
        const float* const bk = KARG(FusedFwdP, bk);
        const float* const bv = KARG(FusedFwdP, bv);
        float* const qG = KARG(FusedFwdP, q);
        float* const kG = KARG(FusedFwdP, k);
        float* const vG = KARG(FusedFwdP, v);
        f32x4 w[2][8];
        f32x4 bias[2][2];
        auto issue = [&](int u, f32x4 (&wd)[8], f32x4 (&bd)[2]) {   // pair u = output slabs 2u, 2u+1 of [Q | K | V]
            const float* W = u < 2 ? Wq : (u < 4 ? Wk : Wv);
            const float* B = u < 2 ? bq : (u < 4 ? bk : bv);
            const int o0 = 2 * (u & 1);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int c = 0; c < 4; ++c) wd[4 * j + c] = wfrag(W, 64, o0 + j, c, n, g);
                bd[j] = gld4(B + 16 * (o0 + j) + 4 * g);
            }
        };
        issue(0, w[0], bias[0]);
#pragma unroll
        for (int c = 0; c < 4; ++c) x[c] = ld4(sX + t * FS + 16 * c + 4 * g);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            if (u + 1 < 6) issue(u + 1, w[(u + 1) & 1], bias[(u + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 a0 = bias[u & 1][0], a1 = bias[u & 1][1];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a0 = mfma16(w[u & 1][c][r], x[c][r], a0);
                    a1 = mfma16(w[u & 1][4 + c][r], x[c][r], a1);
                }
            const int o0 = 2 * (u & 1);
            float* const G = u < 2 ? qG : (u < 4 ? kG : vG);
            gst4(dst(G, erow + 16 * o0), a0); gst4(dst(G, erow + 16 * (o0 + 1)), a1);
            if (u < 2) {
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
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    else {
        // no token of this tile exists (L <= 16 T): its K rows and V^T columns are read by the other owners' MFMAs (the
        // probabilities of those keys are exactly 0, but 0 x uninitialised LDS could be NaN)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            st4(sK + t * FS + 16 * c + 4 * g, f32x4{0, 0, 0, 0});
#pragma unroll
            for (int r = 0; r < 4; ++r) sVt[(16 * c + 4 * g + r) * FS + t] = 0.f;
        }
    }
    lds_barrier();                                                   // ---- B1: K, V^T of every tile and dsp are in LDS
    STAMP(3);
    f32x4 y[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) y[c] = f32x4{0, 0, 0, 0};
    if (tile_on) {
        const float* const Wo = KARG(FusedFwdP, wo);
        f32x4 wo[16], bo4[4];
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
            if (h == NH - 1) {          // dense weights: requested under the last head's softmax, needed right after the attention
#pragma unroll
                for (int o = 0; o < 4; ++o) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) wo[4 * o + c] = wfrag(Wo, 64, o, c, n, g);
                    bo4[o] = gld4(KARG(FusedFwdP, bo) + 16 * o + 4 * g);
                }
                __builtin_amdgcn_sched_barrier(0);
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
        // ---- dense + dropout + residual + LayerNorm + alpha mix
        f32x4 a[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) a[o] = bo4[o];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int o = 0; o < 4; ++o) a[o] = mfma16(wo[4 * o + c][r], ctx[c][r], a[o]);
        // first feed-forward weights requested before the LayerNorm
        const float* const W1 = KARG(FusedFwdP, w1);
        const float* const W2 = KARG(FusedFwdP, w2);
        const float* const b1 = KARG(FusedFwdP, b1);
        f32x4 w1f[2][4], w2f[2][4], fb1[2];
        auto issue_w1 = [&](int j, f32x4 (&wd)[4], f32x4& bd) {     // dense_1 output slab j: 4 in-feature chunks + its bias
#pragma unroll
            for (int c = 0; c < 4; ++c) wd[c] = wfrag(W1, 64, j, c, n, g);
            bd = gld4(b1 + 16 * j + 4 * g);
        };
        auto issue_w2 = [&](int j, f32x4 (&wd)[4]) {                // dense_2 inner chunk j: 4 output slabs
#pragma unroll
            for (int o = 0; o < 4; ++o) wd[o] = wfrag(W2, 256, o, j, n, g);
        };
        issue_w1(0, w1f[0], fb1[0]);
        issue_w1(1, w1f[1], fb1[1]);
        __builtin_amdgcn_sched_barrier(0);
        const DropP drop_o = KARG(FusedFwdP, drop_o);
        const float eps = KARG(FusedFwdP, eps);
        f32x4 hm[4];
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
        STAMP(5);
        // ---- feed-forward as a 3-stage pipeline over the 16 inner slabs.  Step j issues, independent of each other:
        //        MFMA   dense_1 slab j + 1  (16, one chain)            u_{j+1} = b1 + W1[slab j+1] . hmix^T
        //        MFMA   dense_2 inner chunk j - 1  (16, four chains)   y      += W2[:, chunk j-1] . gelu(u_{j-1})
        //        VALU   erf-GELU + GELU' on the accumulator of slab j, both stored for the backward
        //      so the ~200 vector instructions of a slab issue in the shadow of 32 MFMAs that do not wait for them.
        float* const uG = KARG(FusedFwdP, u);
        float* const gpG = KARG(FusedFwdP, gp);
        float* const udst = dst(uG, (tok0 + t) * 256 + 4 * g);
        float* const gdst = dst(gpG, (tok0 + t) * 256 + 4 * g);
        const long ustep = ok ? 16 : 0;
#pragma unroll
        for (int o = 0; o < 4; ++o) y[o] = gld4(KARG(FusedFwdP, b2) + 16 * o + 4 * g);
        f32x4 ucur = fb1[0];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) ucur = mfma16(w1f[0][c][r], hm[c][r], ucur);
        f32x4 glprev = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            // requests: W2 chunk j and W1 slab j + 2 (both used at step j + 1) into the register sets consumed at step j - 1
            issue_w2(j, w2f[j & 1]);
            if (j + 2 < 16) issue_w1(j + 2, w1f[j & 1], fb1[j & 1]);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 unext = fb1[(j + 1) & 1];
            float gq[4], gpq[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                gelu_both(ucur[r], gq[r], gpq[r]);
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    if (j > 0) y[o] = mfma16(w2f[(j - 1) & 1][o][r], glprev[r], y[o]);
                    if (j + 1 < 16) unext = mfma16(w1f[(j + 1) & 1][o][r], hm[o][r], unext);      // (chunk c = o of slab j + 1)
                }
            }
            const f32x4 gl = {gq[0], gq[1], gq[2], gq[3]}, gd = {gpq[0], gpq[1], gpq[2], gpq[3]};
            gst4(udst + ustep * j, gl); gst4(gdst + ustep * j, gd);
            glprev = gl;
            ucur = unext;
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int o = 0; o < 4; ++o) y[o] = mfma16(w2f[15 & 1][o][r], glprev[r], y[o]);
        // ---- dropout + residual + LayerNorm -> block output
        {
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
        }
    }
    STAMP(7);
    if constexpr (TAIL) {
        // the one-row top block as this kernel's tail (fused_top.h): its x tile = this block's output, rows >= L zero
        TopFwdRegs<false> TR;
        top_fwd_prefetch<false, KOFF>(TR);
#pragma unroll
        for (int o = 0; o < 4; ++o) st4(sX + t * FS + 16 * o + 4 * g, y[o]);
        lds_barrier();                                               // ---- B2 (owners only: waves 4..7 have exited)
        STAMP(8);
        top_fwd_rest<DH, false, KOFF>(TR, dseed, sX, sK, sVt, sm + 4 * TS, sTab, sSpec, sD, sIds);
    }
}
#undef PTYPE

static inline size_t fused_chain_fwd_smem_bytes(bool tail) {
    return (size_t)((tail ? 6 : 4) * 64 * FS + 2 * FUSED_MAX_CB * 128 + 64) * 4;
}
