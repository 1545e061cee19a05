// Epilogues of the generic GEMM: all read the C tile from its LDS image (row stride BN+4),
// 4 consecutive columns per lane, so every global access is a 16-byte vector.
#pragma once
#include "gemm.h"

// ---------------------------------------------------------------------------------------------
// out = acc [+ bias[n]] [* gelu'(U[m,n])] [+ R[m,n]]       (projections, input gradients, slabs)
// ---------------------------------------------------------------------------------------------
template <bool BIAS, bool ADD, bool GGRAD>
struct EpiLinear {
    float* C[3];
    long ldc, c_sb, c_sh, c_split;     // row stride, batch strides, split-K slab stride
    const float* bias[3];
    const float* R; long ldr;
    const float* U; long ldu;
    int act;                           // GGRAD: which hidden_act's derivative (0 = gelu)

    template <int BM, int BN, int NT = GEMM_THREADS>
    __device__ __forceinline__ void run(const float* Cs, const TileCtx& c) const {
        constexpr int LDC = BN + 4, CV = BN / 4;
        float* out = C[c.prob] + (long)c.b * c_sb + (long)c.hh * c_sh + (long)c.split * c_split;
        for (int idx = threadIdx.x; idx < BM * CV; idx += NT) {
            const int r = idx / CV, lc = (idx % CV) << 2;
            const int m = c.m0 + r, n = c.n0 + lc;
            if (m >= c.M || n >= c.N) continue;
            f32x4 v = ld4(Cs + r * LDC + lc);
            if (BIAS) v += ld4(bias[c.prob] + n);
            if (GGRAD) {
                const f32x4 u = ld4(U + (long)m * ldu + n);
                v.x *= act_grad_f(u.x, act); v.y *= act_grad_f(u.y, act); v.z *= act_grad_f(u.z, act); v.w *= act_grad_f(u.w, act);
            }
            if (ADD) v += ld4(R + (long)m * ldr + n);
            st4(out + (long)m * ldc + n, v);
        }
    }
};

// ---------------------------------------------------------------------------------------------
// y = LN(Drop(acc + bias) + R) ; MIX: out = alpha*dsp + (1-alpha)*y      (needs BN >= N = d)
// reference: src/model/_modules.py:65-67 (FeedForward), :136-138 (MultiHeadAttention),
//            src/model/bsarec.py:78 (alpha mix)
// ---------------------------------------------------------------------------------------------
template <bool MIX>
struct EpiLN {
    const float* bias; const float* R; DropP drop;
    const float* gamma; const float* beta; float eps;
    float* Y; float* xhat; float* rstd;
    const float* dsp; float alpha, oma;

    template <int BM, int BN, int NT = GEMM_THREADS>
    __device__ __forceinline__ void run(const float* Cs, const TileCtx& c) const {
        constexpr int LDC = BN + 4, LPR = BN / 4, RPP = NT / LPR;
        const int N = c.N;
        const int lr = threadIdx.x / LPR, lc = (threadIdx.x % LPR) << 2;
        const bool colok = lc < N;
        f32x4 bi = {0, 0, 0, 0}, g = bi, be = bi;
        if (colok) { bi = ld4(bias + lc); g = ld4(gamma + lc); be = ld4(beta + lc); }
        const float invn = 1.0f / (float)N;
        for (int r = lr; r < BM; r += RPP) {
            const int m = c.m0 + r;
            const bool ok = colok && m < c.M;
            f32x4 v = {0, 0, 0, 0};
            if (ok) {
                const long e = (long)m * N + lc;
                v = (ld4(Cs + r * LDC + lc) + bi) * drop_mult4(drop, (uint64_t)e >> 2) + ld4(R + e);
            }
            const float mean = group_sum<LPR>(v.x + v.y + v.z + v.w) * invn;
            f32x4 dl = {0, 0, 0, 0};
            if (ok) dl = v - mean;
            const float var = group_sum<LPR>(dl.x * dl.x + dl.y * dl.y + dl.z * dl.z + dl.w * dl.w) * invn;
            const float rs = 1.0f / sqrtf(var + eps);
            if (ok) {
                const long e = (long)m * N + lc;
                const f32x4 xh = dl * rs;
                f32x4 y = g * xh + be;
                if (MIX) y = alpha * ld4(dsp + e) + oma * y;
                st4(xhat + e, xh);
                st4(Y + e, y);
                if (lc == 0) rstd[m] = rs;
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------
// attention probabilities: P = softmax_k(acc / sqrt(dh) + mask)      (needs BN >= L keys)
// mask(q,k) = 0 if (k <= q and ids[b,k] > 0) else -10000, added in fp32 before the row max, as
// the reference does (src/model/_abstract_model.py:53-69, src/model/_modules.py:118-128)
// ---------------------------------------------------------------------------------------------
struct EpiSoftmax {
    const int* ids; int L, Lp; float sqrt_dh; float* P;

    template <int BM, int BN, int NT = GEMM_THREADS>
    __device__ __forceinline__ void run(const float* Cs, const TileCtx& c) const {
        constexpr int LDC = BN + 4, LPR = BN / 4, RPP = NT / LPR;
        const int lr = threadIdx.x / LPR, lc = (threadIdx.x % LPR) << 2;
        bool kvalid[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) kvalid[j] = (lc + j < L) && ids[c.b * L + lc + j] > 0;
        for (int r = lr; r < BM; r += RPP) {
            const int q = c.m0 + r;
            const f32x4 a = ld4(Cs + r * LDC + lc);
            float s[4];
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = lc + j;
                s[j] = a[j] / sqrt_dh + ((kvalid[j] && k <= q) ? 0.0f : -10000.0f);
                if (k >= L) s[j] = -INFINITY;
                mx = fmaxf(mx, s[j]);
            }
            mx = group_max<LPR>(mx);
            float e[4], sum = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { e[j] = (lc + j < L) ? expf(s[j] - mx) : 0.f; sum += e[j]; }
            sum = group_sum<LPR>(sum);
            if (q < L && lc < Lp) {
                f32x4 p = {e[0] / sum, e[1] / sum, e[2] / sum, e[3] / sum};
                st4(P + ((long)c.zb * L + q) * Lp + lc, p);
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------
// attention backward: acc = dC . V^T (gradient w.r.t. the dropped-out probabilities);
// dS = P * (dA - rowsum(dA * P)) / sqrt(dh), dA = acc * keep / (1-p)     (needs BN >= L keys)
// ---------------------------------------------------------------------------------------------
struct EpiDS {
    const float* P; DropP drop; int L, Lp; float sqrt_dh; float* dS;

    template <int BM, int BN, int NT = GEMM_THREADS>
    __device__ __forceinline__ void run(const float* Cs, const TileCtx& c) const {
        constexpr int LDC = BN + 4, LPR = BN / 4, RPP = NT / LPR;
        const int lr = threadIdx.x / LPR, lc = (threadIdx.x % LPR) << 2;
        for (int r = lr; r < BM; r += RPP) {
            const int q = c.m0 + r;
            const bool ok = q < L && lc < Lp;
            f32x4 p = {0, 0, 0, 0}, da = p;
            const long e = ((long)c.zb * L + q) * Lp + lc;
            if (ok) {
                p = ld4(P + e);
                da = ld4(Cs + r * LDC + lc) * drop_mult4(drop, (uint64_t)e >> 2);
            }
            const float delta = group_sum<LPR>(da.x * p.x + da.y * p.y + da.z * p.z + da.w * p.w);
            if (ok) {
                f32x4 ds = p * (da - delta);
                ds.x /= sqrt_dh; ds.y /= sqrt_dh; ds.z /= sqrt_dh; ds.w /= sqrt_dh;
                st4(dS + e, ds);
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------
// Grouped weight-gradient products: up to 6 independent TN split-K problems of different shapes in
// ONE launch (dWq dWk dWv dWo dW1 dW2 of a BSARecBlock + their bias gradients).  blockIdx.x walks the
// concatenated 64x64 output tiles, blockIdx.y is the split-K slice.
// ---------------------------------------------------------------------------------------------
#define GROUP_MAX 6
struct GroupedTN {
    GemmP P[GROUP_MAX];
    EpiLinear<false, false, false> E[GROUP_MAX];
    float* bgrad[GROUP_MAX];
    int tile0[GROUP_MAX + 1];
    int tiles_n[GROUP_MAX];
    int b_gelu[GROUP_MAX];          // apply the hidden_act to the B operand while loading (dW2 = dT^T . act(U))
    int nprob;
    int act;                        // which hidden_act (0 = gelu)
};

// BF: bf16 products (gemm.h), cfg.storage = 1 outside the fused shape class.  TS: tile size -- 128 at hidden >= 128 (the host
// counts the tiles with the same TS): half the fragment reads and staging writes per MFMA of the 64 x 64 form
template <bool BF, int TS>
__global__ void __launch_bounds__(GEMM_THREADS, (BF && TS == 128) ? 2 : 1)
gemm_grouped_tn_kernel(const GroupedTN G) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int p = 0;
    while (p + 1 < G.nprob && (int)blockIdx.x >= G.tile0[p + 1]) ++p;
    const int local = blockIdx.x - G.tile0[p];
    const int bx = local / G.tiles_n[p], by = local % G.tiles_n[p];
    XformP X;
    X.L = 0; X.Lp = 0; X.drop.thresh = 0; X.drop.scale = 1.f; X.drop.rng = nullptr; X.drop.site = 0; X.act = G.act;
    if (G.b_gelu[p])
        gemm_body<TS, TS, 2, 2, true, true, XF_NONE, XF_GELU, true, BF>(G.P[p], X, G.E[p], G.bgrad[p], bx, by, blockIdx.y, smem);
    else
        gemm_body<TS, TS, 2, 2, true, true, XF_NONE, XF_NONE, true, BF>(G.P[p], X, G.E[p], G.bgrad[p], bx, by, blockIdx.y, smem);
}

// ---------------------------------------------------------------------------------------------
// The two products of the logits backward in ONE launch: dE = dlogits^T . h_last (TN, [V, d]) and the split-K
// slabs of d(h_last) = dlogits . E (NN).  blockIdx.x < tilesA -> problem A, else problem B.
// ---------------------------------------------------------------------------------------------
struct PairP {
    GemmP A, B;
    EpiLinear<false, false, false> EA, EB;
    int tilesA, tilesB_m;          // B: blocks = tilesB_m (m tiles) x B.nsplit
};

__global__ void __launch_bounds__(GEMM_THREADS)
gemm_logits_bwd_kernel(const PairP G) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    XformP X;
    X.L = 0; X.Lp = 0; X.drop.thresh = 0; X.drop.scale = 1.f; X.drop.rng = nullptr; X.drop.site = 0;
    const int bx = blockIdx.x;
    if (bx < G.tilesA)
        gemm_body<64, 64, 2, 2, true, true, XF_NONE, XF_NONE, false>(G.A, X, G.EA, nullptr, bx, 0, 0, smem);
    else {
        const int r = bx - G.tilesA;
        gemm_body<64, 64, 2, 2, false, true, XF_NONE, XF_NONE, false>(G.B, X, G.EB, nullptr, r % G.tilesB_m, 0, r / G.tilesB_m, smem);
    }
}
