// Row-parallel (HBM/L2-bound) kernels of the BSARec path: embedding front-end, FrequencyLayer
// (pruned DFT in LDS), LayerNorm backward, fused cross-entropy, embedding-gradient scatter,
// deterministic partial reductions and the fused Adam.  Common shape: a row of d floats is held by
// LPR = {16,32,64} consecutive lanes, 4 consecutive channels per lane (16-byte accesses), so one
// wave covers 64/LPR rows and row statistics are intra-wave shuffles.
#pragma once
#include "common.h"

#define ROW_THREADS 256

// =============================================================================================
// K1 embedding front-end: X0 = Drop(LN(E[ids] + Pos[t]))      src/model/_abstract_model.py:14-24
// =============================================================================================
// Optional batch assembly folded into the embedding kernel (the device-resident sample table of bsarec_amd.data):
// ids[b,:] = table[perm[cursor + b], :], answers[b] = ans_table[perm[cursor + b]]
// (replaces RandomSampler + DataLoader collation, src/dataset.py:207-211).  table == nullptr: ids come from `ids`.
struct GatherP {
    const int64_t* table; const int64_t* ans_table; const int64_t* perm; long n;
    const long long* cursor; int64_t* ids_out; int64_t* ans_out;
};

template <int LPR>
__global__ void __launch_bounds__(ROW_THREADS)
embed_fwd_kernel(const int64_t* __restrict__ ids, GatherP gp, const float* __restrict__ E, const float* __restrict__ Pos,
                 const float* __restrict__ gamma, const float* __restrict__ beta, float eps, DropP drop,
                 int T, int L, int d, int V, float* __restrict__ X0, float* __restrict__ xhat,
                 float* __restrict__ rstd, int* __restrict__ ids32) {
    constexpr int RPB = ROW_THREADS / LPR;
    const int lr = threadIdx.x / LPR, lc = (threadIdx.x % LPR) << 2;
    const int tok = blockIdx.x * RPB + lr;
    const bool ok = tok < T && lc < d;
    f32x4 v = {0, 0, 0, 0};
    if (ok) {
        int64_t id64;
        if (gp.table) {
            const int b = tok / L, t = tok - b * L;
            long src = *gp.cursor + b;
            src = src < gp.n ? gp.perm[src] : 0;
            id64 = gp.table[src * L + t];
            if (lc == 0) {
                gp.ids_out[tok] = id64;
                if (t == 0) gp.ans_out[b] = gp.ans_table[src];
            }
        } else id64 = ids[tok];
        int id = (int)id64;
        id = id < 0 ? 0 : (id >= V ? V - 1 : id);     // defensive clamp: never read outside the table
        if (lc == 0) ids32[tok] = id;
        v = ld4(E + (long)id * d + lc) + ld4(Pos + (long)(tok % L) * d + lc);
    }
    const float invd = 1.0f / (float)d;
    const float mean = group_sum<LPR>(v.x + v.y + v.z + v.w) * invd;
    f32x4 dl = {0, 0, 0, 0};
    if (ok) dl = v - mean;
    const float var = group_sum<LPR>(dl.x * dl.x + dl.y * dl.y + dl.z * dl.z + dl.w * dl.w) * invd;
    const float rs = 1.0f / sqrtf(var + eps);
    if (ok) {
        const long e = (long)tok * d + lc;
        const f32x4 xh = dl * rs;
        st4(xhat + e, xh);
        st4(X0 + e, (ld4(gamma + lc) * xh + ld4(beta + lc)) * drop_mult4(drop, (uint64_t)e >> 2));
        if (lc == 0) rstd[tok] = rs;
    }
}

// =============================================================================================
// LayerNorm backward (row op) with the dropout that sits before / after it, plus per-block
// partial sums of d(gamma), d(beta).   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy*gamma
// =============================================================================================
struct LnBranch {
    const float* xhat; const float* rstd; const float* gamma;
    float in_scale;            // alpha / (1 - alpha) of the BSARec mix, else 1
    DropP drop;
    float* dT;                 // POST-drop output: dz * keep/(1-p)   (gradient of the dropped-out tensor)
    float* pgamma; float* pbeta;   // [gridDim.x][d] partials
};

// MODE 0: one branch, dropout AFTER LN in backward order (y = LN(Drop(z) + res)):  dz -> dZ, dT
// MODE 1: two branches on the same upstream gradient (attention LN and filter LN):
//         dXacc = dzA + dzF ; dT of each branch
// MODE 2: embedding: y = Drop(LN(e)):  de = LNbwd(dy * keep/(1-p)) -> dZ
template <int LPR, int MODE>
__global__ void __launch_bounds__(ROW_THREADS)
ln_bwd_kernel(const float* __restrict__ dY, LnBranch a, LnBranch f, float* __restrict__ dZ, int T, int d,
              int rows_per_block) {
    constexpr int RPP = ROW_THREADS / LPR;
    __shared__ float red[4][RPP][LPR * 4];
    const int lr = threadIdx.x / LPR, lc = (threadIdx.x % LPR) << 2;
    const bool colok = lc < d;
    const float invd = 1.0f / (float)d;
    f32x4 ga = {0, 0, 0, 0}, gf = ga;
    if (colok) { ga = ld4(a.gamma + lc); if (MODE == 1) gf = ld4(f.gamma + lc); }
    f32x4 sga = {0, 0, 0, 0}, sba = sga, sgf = sga, sbf = sga;
    const int row0 = blockIdx.x * rows_per_block;
    for (int r = lr; r < rows_per_block; r += RPP) {
        const int tok = row0 + r;
        const bool ok = colok && tok < T;
        const long e = (long)tok * d + lc;
        f32x4 dy = {0, 0, 0, 0}, xa = dy, xf = dy;
        float ra = 0.f, rf = 0.f;
        if (ok) {
            dy = ld4(dY + e);
            if (MODE == 2) dy = dy * drop_mult4(a.drop, (uint64_t)e >> 2);
            xa = ld4(a.xhat + e); ra = a.rstd[tok];
            if (MODE == 1) { xf = ld4(f.xhat + e); rf = f.rstd[tok]; }
        }
        // branch a
        const f32x4 dya = dy * a.in_scale;
        const f32x4 g = dya * ga;
        const float m1 = group_sum<LPR>(g.x + g.y + g.z + g.w) * invd;
        const float m2 = group_sum<LPR>(g.x * xa.x + g.y * xa.y + g.z * xa.z + g.w * xa.w) * invd;
        const f32x4 dza = ra * (g - m1 - xa * m2);
        sga += dya * xa; sba += dya;
        f32x4 dzsum = dza;
        if (MODE == 1) {
            const f32x4 dyf = dy * f.in_scale;
            const f32x4 g2 = dyf * gf;
            const float n1 = group_sum<LPR>(g2.x + g2.y + g2.z + g2.w) * invd;
            const float n2 = group_sum<LPR>(g2.x * xf.x + g2.y * xf.y + g2.z * xf.z + g2.w * xf.w) * invd;
            const f32x4 dzf = rf * (g2 - n1 - xf * n2);
            sgf += dyf * xf; sbf += dyf;
            dzsum += dzf;
            if (ok) st4(f.dT + e, dzf * drop_mult4(f.drop, (uint64_t)e >> 2));
        }
        if (ok) {
            st4(dZ + e, dzsum);
            if (MODE != 2) st4(a.dT + e, dza * drop_mult4(a.drop, (uint64_t)e >> 2));
        }
    }
    // block-level partials: sum the RPP row groups through LDS, fixed order -> deterministic
    st4(&red[0][lr][lc], sga); st4(&red[1][lr][lc], sba);
    if (MODE == 1) { st4(&red[2][lr][lc], sgf); st4(&red[3][lr][lc], sbf); }
    __syncthreads();
    constexpr int NV = (MODE == 1) ? 4 : 2;
    for (int i = threadIdx.x; i < NV * LPR * 4; i += ROW_THREADS) {
        const int which = i / (LPR * 4), c = i % (LPR * 4);
        if (c >= d) continue;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < RPP; ++k) s += red[which][k][c];
        float* dst = which == 0 ? a.pgamma : which == 1 ? a.pbeta : which == 2 ? f.pgamma : f.pbeta;
        dst[(long)blockIdx.x * d + c] = s;
    }
}

// =============================================================================================
// K2 FrequencyLayer: low = irfft(trunc_cb(rfft(x))) as a pruned DFT held in LDS.
//   X_k[c] = sum_t x[t,c] e^{-2 pi i k t / L}, k < cb;   low[t,c] = (1/L) sum_k w_k Re(X_k e^{+2 pi i k t/L})
//   (the 1/sqrt(L) of the two 'ortho' transforms combine to 1/L).
// One block per sequence; a row of d channels = LPR lanes x 4; the spectrum lives in LDS as
// spec[k][re|im][d]; twiddle[j] = (cos, sin)(2 pi j / L) is a host-built table (fp64 -> fp32).
//   forward : DSP = LN(Drop(low + beta^2 (x - low)) + x)                 src/model/bsarec.py:90-104
//   backward: dX  = dXin + beta^2 dF + lowpass((1 - beta^2) dF),  dbeta = 2 beta sum dF (x - low)
// =============================================================================================
#define FREQ_KC 4
template <int LPR, int NSRC, class Src>
__device__ __forceinline__ void dft_spectrum(const Src& src, int L, int d, int cb, const float* __restrict__ tw,
                                             float* __restrict__ spec /* [NSRC][cb][2][d] */,
                                             float* __restrict__ part /* [RPP][NSRC][KC][2][LPR*4] */) {
    constexpr int RPP = ROW_THREADS / LPR, W = LPR * 4;
    const int lr = threadIdx.x / LPR, lc = (threadIdx.x % LPR) << 2;
    const bool colok = lc < d;
    for (int k0 = 0; k0 < cb; k0 += FREQ_KC) {
        f32x4 re[NSRC][FREQ_KC], im[NSRC][FREQ_KC];
#pragma unroll
        for (int s = 0; s < NSRC; ++s)
#pragma unroll
            for (int j = 0; j < FREQ_KC; ++j) { re[s][j] = f32x4{0, 0, 0, 0}; im[s][j] = f32x4{0, 0, 0, 0}; }
        for (int t = lr; t < L; t += RPP) {
            f32x4 x[NSRC];
#pragma unroll
            for (int s = 0; s < NSRC; ++s) x[s] = colok ? src(s, t, lc) : f32x4{0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < FREQ_KC; ++j) {
                const int k = k0 + j;
                if (k < cb) {
                    const int a = (int)((unsigned)(k * t) % (unsigned)L);
                    const float c = tw[2 * a], sn = tw[2 * a + 1];
#pragma unroll
                    for (int s = 0; s < NSRC; ++s) { re[s][j] += x[s] * c; im[s][j] -= x[s] * sn; }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < NSRC; ++s)
#pragma unroll
            for (int j = 0; j < FREQ_KC; ++j) {
                st4(part + (((lr * NSRC + s) * FREQ_KC + j) * 2 + 0) * W + lc, re[s][j]);
                st4(part + (((lr * NSRC + s) * FREQ_KC + j) * 2 + 1) * W + lc, im[s][j]);
            }
        __syncthreads();
        for (int i = threadIdx.x; i < NSRC * FREQ_KC * 2 * W; i += ROW_THREADS) {
            const int c = i % W, ri = (i / W) & 1, j = (i / (2 * W)) % FREQ_KC, s = i / (2 * W * FREQ_KC);
            if (c < d && k0 + j < cb) {
                float acc = 0.f;
#pragma unroll
                for (int g = 0; g < RPP; ++g) acc += part[(((g * NSRC + s) * FREQ_KC + j) * 2 + ri) * W + c];
                spec[((long)(s * cb + k0 + j) * 2 + ri) * d + c] = acc;
            }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ f32x4 dft_lowpass_at(const float* __restrict__ spec, int t, int lc, int L, int d, int cb,
                                                const float* __restrict__ tw) {
    f32x4 low = {0, 0, 0, 0};
    for (int k = 0; k < cb; ++k) {
        const int a = (int)((unsigned)(k * t) % (unsigned)L);
        const float w = (k == 0 || (2 * k == L)) ? 1.0f : 2.0f;
        const float c = tw[2 * a] * w, sn = tw[2 * a + 1] * w;
        low += ld4(spec + ((long)k * 2 + 0) * d + lc) * c - ld4(spec + ((long)k * 2 + 1) * d + lc) * sn;
    }
    return low * (1.0f / (float)L);
}

template <int LPR>
__global__ void __launch_bounds__(ROW_THREADS)
freq_fwd_kernel(const float* __restrict__ X, const float* __restrict__ sqrt_beta, const float* __restrict__ gamma,
                const float* __restrict__ beta, float eps, DropP drop, const float* __restrict__ twg, int L, int d,
                int cb, float* __restrict__ DSP, float* __restrict__ xhat, float* __restrict__ rstd,
                const float* __restrict__ cw /* FMLPRec: complex_weight [cb][d][2], else null */) {
    constexpr int RPP = ROW_THREADS / LPR, W = LPR * 4;
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    float* tw = fsm;                               // [2L] rounded up to 4
    float* spec = tw + ((2 * L + 3) & ~3);         // [cb][2][d]
    float* part = spec + (long)cb * 2 * d;         // [RPP][KC][2][W]
    const int b = blockIdx.x;
    const float* x = X + (long)b * L * d;
    for (int i = threadIdx.x; i < 2 * L; i += ROW_THREADS) tw[i] = twg[i];
    __syncthreads();
    auto src = [&](int, int t, int lc) { return ld4(x + (long)t * d + lc); };
    dft_spectrum<LPR, 1>(src, L, d, cb, tw, spec, part);
    if (cw) {        // sibling model FMLPRec (src/model/fmlprec.py:103-108): Y_k = X_k W_k, then the same inverse transform
        for (int i = threadIdx.x; i < cb * d; i += ROW_THREADS) {
            const int k = i / d, c = i - k * d;
            const float xr = spec[((long)k * 2 + 0) * d + c], xi = spec[((long)k * 2 + 1) * d + c];
            const float wr = cw[(long)i * 2], wi = cw[(long)i * 2 + 1];
            spec[((long)k * 2 + 0) * d + c] = xr * wr - xi * wi;
            spec[((long)k * 2 + 1) * d + c] = xr * wi + xi * wr;
        }
        __syncthreads();
    }

    const int lr = threadIdx.x / LPR, lc = (threadIdx.x % LPR) << 2;
    const bool colok = lc < d;
    f32x4 b2 = {0, 0, 0, 0}, g = b2, be = b2;
    if (colok) { b2 = ld4(sqrt_beta + lc); b2 = b2 * b2; g = ld4(gamma + lc); be = ld4(beta + lc); }
    if (cw) b2 = f32x4{0, 0, 0, 0};               // f = the filtered signal itself (no high-pass remainder)
    const float invd = 1.0f / (float)d;
    for (int t0 = 0; t0 < L; t0 += RPP) {
        const int t = t0 + lr;
        const bool ok = colok && t < L;
        f32x4 v = {0, 0, 0, 0};
        const long e = ((long)b * L + t) * d + lc;
        if (ok) {
            const f32x4 xv = ld4(x + (long)t * d + lc);
            const f32x4 low = dft_lowpass_at(spec, t, lc, L, d, cb, tw);
            const f32x4 fl = low + b2 * (xv - low);
            v = fl * drop_mult4(drop, (uint64_t)e >> 2) + xv;
        }
        const float mean = group_sum<LPR>(v.x + v.y + v.z + v.w) * invd;
        f32x4 dl = {0, 0, 0, 0};
        if (ok) dl = v - mean;
        const float var = group_sum<LPR>(dl.x * dl.x + dl.y * dl.y + dl.z * dl.z + dl.w * dl.w) * invd;
        const float rs = 1.0f / sqrtf(var + eps);
        if (ok) {
            const f32x4 xh = dl * rs;
            st4(xhat + e, xh);
            st4(DSP + e, g * xh + be);
            if (lc == 0) rstd[(long)b * L + t] = rs;
        }
    }
}

template <int LPR>
__global__ void __launch_bounds__(ROW_THREADS)
freq_bwd_kernel(const float* __restrict__ X, const float* __restrict__ dF, const float* __restrict__ dXin,
                const float* __restrict__ sqrt_beta, const float* __restrict__ twg, int L, int d, int cb,
                float* __restrict__ dX, float* __restrict__ pbeta /* [B][d] */,
                const float* __restrict__ cw /* FMLPRec: complex_weight [cb][d][2], else null */,
                float* __restrict__ pcw /* FMLPRec: per-sequence d(complex_weight) [B][cb][d][2] */) {
    constexpr int RPP = ROW_THREADS / LPR, W = LPR * 4;
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    float* tw = fsm;
    float* spec = tw + ((2 * L + 3) & ~3);         // [2][cb][2][d]: source 0 = x, source 1 = (1-beta^2) dF
    float* part = spec + (long)2 * cb * 2 * d;     // [RPP][2][KC][2][W]
    const int b = blockIdx.x;
    const long base = (long)b * L * d;
    for (int i = threadIdx.x; i < 2 * L; i += ROW_THREADS) tw[i] = twg[i];
    __syncthreads();
    const int lr = threadIdx.x / LPR, lc = (threadIdx.x % LPR) << 2;
    const bool colok = lc < d;
    f32x4 bt = {0, 0, 0, 0};
    if (colok) bt = ld4(sqrt_beta + lc);
    f32x4 b2 = bt * bt, omb2 = 1.0f - b2;
    if (cw) { b2 = f32x4{0, 0, 0, 0}; omb2 = f32x4{1, 1, 1, 1}; }
    auto src = [&](int s, int t, int c) {
        const f32x4 v = ld4((s == 0 ? X : dF) + base + (long)t * d + c);
        return s == 0 ? v : v * omb2;
    };
    dft_spectrum<LPR, 2>(src, L, d, cb, tw, spec, part);
    if (cw) {
        // sibling model FMLPRec: with D = spectrum of dF, dX = inverse transform of D conj(W) (the same inverse as the
        // forward's) and d(W)_k = (w_k / L) conj(X_k) D_k per sequence (w_k = 1 for DC / Nyquist, else 2)
        float* sD = spec + (long)cb * 2 * d;
        for (int i = threadIdx.x; i < cb * d; i += ROW_THREADS) {
            const int k = i / d, c = i - k * d;
            const float xr = spec[((long)k * 2 + 0) * d + c], xi = spec[((long)k * 2 + 1) * d + c];
            const float dr = sD[((long)k * 2 + 0) * d + c], di = sD[((long)k * 2 + 1) * d + c];
            const float wr = cw[(long)i * 2], wi = cw[(long)i * 2 + 1];
            const float wl = ((k == 0 || 2 * k == L) ? 1.0f : 2.0f) / (float)L;
            pcw[((long)b * cb * d + i) * 2 + 0] = wl * (dr * xr + di * xi);
            pcw[((long)b * cb * d + i) * 2 + 1] = wl * (di * xr - dr * xi);
            sD[((long)k * 2 + 0) * d + c] = dr * wr + di * wi;
            sD[((long)k * 2 + 1) * d + c] = di * wr - dr * wi;
        }
        __syncthreads();
    }
    f32x4 sb = {0, 0, 0, 0};
    for (int t = lr; t < L; t += RPP) {
        if (!colok) continue;
        const long e = base + (long)t * d + lc;
        const f32x4 xv = ld4(X + e), df = ld4(dF + e);
        const f32x4 lowx = dft_lowpass_at(spec, t, lc, L, d, cb, tw);
        const f32x4 lowg = dft_lowpass_at(spec + (long)cb * 2 * d, t, lc, L, d, cb, tw);
        st4(dX + e, ld4(dXin + e) + b2 * df + lowg);
        sb += df * (xv - lowx);
    }
    __syncthreads();
    st4(part + lr * W + lc, sb);
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += ROW_THREADS) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < RPP; ++g) s += part[g * W + c];
        pbeta[(long)b * d + c] = cw ? 0.f : 2.0f * sqrt_beta[c] * s;        // FMLPRec has no sqrt_beta
    }
}

// =============================================================================================
// K7 fused cross-entropy over the materialised logits row: lse, per-row loss, dlogits
//   loss = mean_b (lse_b - logits[b, answer_b])                              src/model/bsarec.py:33-35
// =============================================================================================
#define CE_MAX_PER_THREAD 16           // rows up to 4096 classes stay in registers; longer rows re-read memory
__global__ void __launch_bounds__(ROW_THREADS)
ce_rows_kernel(const float* __restrict__ logits, const int64_t* __restrict__ answers, int V, int Vp, float inv_b,
               float* __restrict__ dlogits, float* __restrict__ loss_rows) {
    __shared__ float red[ROW_THREADS / 64];
    __shared__ float bc;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* row = logits + (long)b * Vp;
    const bool inreg = V <= CE_MAX_PER_THREAD * ROW_THREADS;
    float x[CE_MAX_PER_THREAD];
    float mx = -INFINITY;
    if (inreg) {
#pragma unroll
        for (int k = 0; k < CE_MAX_PER_THREAD; ++k) {
            const int v = tid + k * ROW_THREADS;
            x[k] = v < V ? row[v] : -INFINITY;
        }
#pragma unroll
        for (int k = 0; k < CE_MAX_PER_THREAD; ++k) mx = fmaxf(mx, x[k]);
    } else {
        for (int v = tid; v < V; v += ROW_THREADS) mx = fmaxf(mx, row[v]);
    }
    mx = group_max<64>(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) { float m = red[0]; for (int i = 1; i < ROW_THREADS / 64; ++i) m = fmaxf(m, red[i]); bc = m; }
    __syncthreads();
    mx = bc;
    float s = 0.f;
    if (inreg) {
#pragma unroll
        for (int k = 0; k < CE_MAX_PER_THREAD; ++k) s += expf(x[k] - mx);      // exp(-inf) = 0 for the tail
    } else {
        for (int v = tid; v < V; v += ROW_THREADS) s += expf(row[v] - mx);
    }
    s = group_sum<64>(s);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) { float t = 0.f; for (int i = 0; i < ROW_THREADS / 64; ++i) t += red[i]; bc = mx + logf(t); }
    __syncthreads();
    const float lse = bc;
    int ans = (int)answers[b];
    ans = ans < 0 ? 0 : (ans >= V ? V - 1 : ans);
    if (inreg) {
#pragma unroll
        for (int k = 0; k < CE_MAX_PER_THREAD; ++k) {
            const int v = tid + k * ROW_THREADS;
            if (v < Vp) dlogits[(long)b * Vp + v] = v < V ? (expf(x[k] - lse) - (v == ans ? 1.0f : 0.0f)) * inv_b : 0.f;
        }
    } else {
        for (int v = tid; v < Vp; v += ROW_THREADS) {
            float g = 0.f;
            if (v < V) g = (expf(row[v] - lse) - (v == ans ? 1.0f : 0.0f)) * inv_b;
            dlogits[(long)b * Vp + v] = g;
        }
    }
    if (tid == 0) loss_rows[b] = lse - row[ans];
}

// The same for 4,096 < V <= 24,576 (Beauty: 12,102, Yelp: 20,034 -- configs C2 / C4): the row still fits the registers of one
// workgroup as NV4 float4 per thread, all of them requested before the first use (ONE round trip instead of three passes of
// dependent 4-byte loads over 48 ... 80 KB: the streaming form above took 35.7 us per step at V = 12,102, 22 % of C2's step).
// Element v = 4 (tid + 256 k) + j; rows start 16-byte aligned (Vp is a multiple of 4), pad columns V .. Vp get gradient 0.
template <int NV4>
__global__ void __launch_bounds__(ROW_THREADS)
ce_rows_vec_kernel(const float* __restrict__ logits, const int64_t* __restrict__ answers, int V, int Vp, float inv_b,
                   float* __restrict__ dlogits, float* __restrict__ loss_rows) {
    __shared__ float red[ROW_THREADS / 64];
    __shared__ float bc;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* row = logits + (long)b * Vp;
    f32x4 x[NV4];
#pragma unroll
    for (int k = 0; k < NV4; ++k) {
        const int v = 4 * (tid + k * ROW_THREADS);
        f32x4 t = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (v < Vp) t = ld4(row + v);
        x[k] = t;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < NV4; ++k) {
        const int v = 4 * (tid + k * ROW_THREADS);
#pragma unroll
        for (int j = 0; j < 4; ++j) { if (v + j >= V) x[k][j] = -INFINITY; mx = fmaxf(mx, x[k][j]); }
    }
    mx = group_max<64>(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) { float m = red[0]; for (int i = 1; i < ROW_THREADS / 64; ++i) m = fmaxf(m, red[i]); bc = m; }
    __syncthreads();
    mx = bc;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += expf(x[k][j] - mx);                  // exp(-inf) = 0 for the tail
    s = group_sum<64>(s);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) { float t = 0.f; for (int i = 0; i < ROW_THREADS / 64; ++i) t += red[i]; bc = mx + logf(t); }
    __syncthreads();
    const float lse = bc;
    int ans = (int)answers[b];
    ans = ans < 0 ? 0 : (ans >= V ? V - 1 : ans);
#pragma unroll
    for (int k = 0; k < NV4; ++k) {
        const int v = 4 * (tid + k * ROW_THREADS);
        if (v < Vp) {
            f32x4 g;
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = v + j < V ? (expf(x[k][j] - lse) - (v + j == ans ? 1.0f : 0.0f)) * inv_b : 0.f;
            st4(dlogits + (long)b * Vp + v, g);
        }
    }
    if (tid == 0) loss_rows[b] = lse - row[ans];
}

// gradient of the last layer's output: zero everywhere except position L-1 of every sequence, where
// it is the split-K sum of dlogits . E                                   (src/model/bsarec.py:32)
__global__ void __launch_bounds__(ROW_THREADS)
dlast_kernel(const float* __restrict__ slabs, int nsplit, long slab_stride, int T, int L, int d,
             float* __restrict__ dX) {
    const long i = (long)blockIdx.x * ROW_THREADS + threadIdx.x;       // float4 index
    const long n4 = (long)T * d / 4;
    if (i >= n4) return;
    const long e = i * 4;
    const int tok = (int)(e / d), c = (int)(e % d);
    f32x4 v = {0, 0, 0, 0};
    if (tok % L == L - 1) {
        const int b = tok / L;
        const float* p = slabs + (long)b * d + c;
        f32x4 a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = f32x4{0, 0, 0, 0};
        int s = 0;
        for (; s + 7 < nsplit; s += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += ld4(p + (long)(s + k) * slab_stride);
        }
        for (; s < nsplit; ++s) a[0] += ld4(p + (long)s * slab_stride);
        v = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    st4(dX + e, v);
}

// =============================================================================================
// embedding-gradient tail: dPos[t] = sum_b de[b,t]  (deterministic), dE[ids] += de for ids != 0
// (padding_idx = 0 suppresses only the lookup gradient, src/model/_abstract_model.py:10).  The dense
// logits-path dE is already in place; rows are added with full-row (>= 64 B contiguous) f32 atomics.
// =============================================================================================
#ifndef SCATTER_FLOATS
#define SCATTER_FLOATS 4096           // LDS row accumulators per block: chunk = 4096 / (4*LPR) tokens (64 at d = 64:
                                      // 200 blocks at C1; 128-token chunks measured 18.2 us, 64- and 32-token chunks 11.6 us)
#endif
// One scatter block: CHUNK tokens starting at chunk * CHUNK.  esm: SCATTER_FLOATS floats + 2 * CHUNK ints of LDS.
template <int LPR>
__device__ __forceinline__ void embed_scatter_block(const float* __restrict__ de, const int* __restrict__ ids32, int T, int d,
                                                    float* __restrict__ dE, int chunk, float* __restrict__ esm) {
    constexpr int RPP = ROW_THREADS / LPR, W = LPR * 4, CHUNK = SCATTER_FLOATS / W;
    float* acc = esm;                                   // [CHUNK][W]
    int* sid = reinterpret_cast<int*>(esm + SCATTER_FLOATS);      // [CHUNK]
    int* lead = sid + CHUNK;                                      // [CHUNK]
    const int lr = threadIdx.x / LPR, lc = (threadIdx.x % LPR) << 2;
    const bool colok = lc < d;
    {
        // Popular items (Zipf) would serialise hundreds of global float atomics on one row.  Each block
        // owns CHUNK tokens: every token finds the first occurrence of its id in the chunk (its leader),
        // all rows are folded into the leader's LDS row with LDS atomics, and each leader then issues ONE
        // global atomic row add (>= 64 B contiguous per row).
        const int t0 = chunk * CHUNK;
        const int n = min(CHUNK, T - t0);
        for (int i = threadIdx.x; i < CHUNK; i += ROW_THREADS) sid[i] = i < n ? ids32[t0 + i] : 0;
        for (int i = threadIdx.x; i < SCATTER_FLOATS / 4; i += ROW_THREADS) st4(acc + 4 * i, f32x4{0, 0, 0, 0});
        __syncthreads();
        constexpr int NP = CHUNK / RPP;                   // rows per thread group, loads issued together
        f32x4 g[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int j = lr + p * RPP;
            g[p] = f32x4{0, 0, 0, 0};
            if (colok && j < n && sid[j] != 0) g[p] = ld4(de + (long)(t0 + j) * d + lc);
        }
        for (int j = threadIdx.x; j < CHUNK; j += ROW_THREADS) {      // first occurrence of sid[j]: 4 ids per LDS read,
            const int id = sid[j];                                    // no early exit (independent, pipelined reads)
            int l = j;
            for (int i = CHUNK - 4; i >= 0; i -= 4) {
                const int4 q = *reinterpret_cast<const int4*>(sid + i);
                if (q.w == id && i + 3 < l) l = i + 3;
                if (q.z == id && i + 2 < l) l = i + 2;
                if (q.y == id && i + 1 < l) l = i + 1;
                if (q.x == id && i < l) l = i;
            }
            lead[j] = l;
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int j = lr + p * RPP;
            if (colok && j < n && sid[j] != 0) {          // padding_idx = 0: no lookup gradient for id 0
                float* a = acc + lead[j] * W + lc;
                atomicAdd(a + 0, g[p].x); atomicAdd(a + 1, g[p].y); atomicAdd(a + 2, g[p].z); atomicAdd(a + 3, g[p].w);
            }
        }
        __syncthreads();
        // one leader row per wave-instruction: lane = column, so every global_atomic_add_f32 covers 256 contiguous
        // bytes (the full-rate shape; 4-byte pieces at a 16-byte stride run ~17x slower)
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        constexpr int PW = CHUNK / (ROW_THREADS / 64);      // tokens owned by one wave (<= 64)
        const int j0 = wv * PW, myj = j0 + lane;
        const bool is_leader = lane < PW && myj < n && sid[myj] != 0 && lead[myj] == myj;
        unsigned long long todo = __ballot(is_leader);      // wave-uniform work list: no per-token LDS round trips
        while (todo) {
            const int j = j0 + __builtin_ctzll(todo);
            todo &= todo - 1;
            const long row = (long)sid[j] * d;
            for (int c0 = 0; c0 < d; c0 += 64)
                if (c0 + lane < d) unsafeAtomicAdd(dE + row + c0 + lane, acc[j * W + c0 + lane]);
        }
    }
}

template <int LPR>
__global__ void __launch_bounds__(ROW_THREADS)
embed_bwd_kernel(const float* __restrict__ de, const int* __restrict__ ids32, int B, int L, int d,
                 float* __restrict__ dE, float* __restrict__ dPos, int scatter_blocks) {
    constexpr int RPP = ROW_THREADS / LPR, W = LPR * 4;
    extern __shared__ __attribute__((aligned(16))) float esm[];
    const int lr = threadIdx.x / LPR, lc = (threadIdx.x % LPR) << 2;
    const bool colok = lc < d;
    if ((int)blockIdx.x < scatter_blocks) {
        embed_scatter_block<LPR>(de, ids32, B * L, d, dE, blockIdx.x, esm);
        return;
    }
    float (*red)[W] = reinterpret_cast<float (*)[W]>(esm);
    // position gradient: block (t, slice) sums de[b, t, :] over the 64 sequences of its slice -> dPos partial
    // [slice][L][d]; the slices are summed by multi_reduce_kernel (deterministic)
    const int pb = blockIdx.x - scatter_blocks;
    const int t = pb % L, slice = pb / L;
    const int b0 = slice * 64, b1 = min(B, b0 + 64);
    f32x4 s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
    if (colok) {
        int bb = b0 + lr;
        for (; bb + 3 * RPP < b1; bb += 4 * RPP) {
            s0 += ld4(de + ((long)bb * L + t) * d + lc);
            s1 += ld4(de + ((long)(bb + RPP) * L + t) * d + lc);
            s2 += ld4(de + ((long)(bb + 2 * RPP) * L + t) * d + lc);
            s3 += ld4(de + ((long)(bb + 3 * RPP) * L + t) * d + lc);
        }
        for (; bb < b1; bb += RPP) s0 += ld4(de + ((long)bb * L + t) * d + lc);
    }
    st4(&red[lr][lc], (s0 + s1) + (s2 + s3));
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += ROW_THREADS) {
        float a = 0.f;
#pragma unroll
        for (int g = 0; g < RPP; ++g) a += red[g][c];
        dPos[((long)slice * L + t) * d + c] = a;
    }
}

// =============================================================================================
// deterministic second-stage reductions: dst[i] = scale * sum_s src[s*stride + i]
// =============================================================================================
struct ReduceJob { const float* src; float* dst; int nsplit; int len; long stride; float scale; int pad; };

__device__ __forceinline__ void reduce_chunk(const ReduceJob& j, int chunk, float (*red)[64]) {
    const int e = threadIdx.x & 63, sg = threadIdx.x >> 6;       // element within the block, split group
    const int i = chunk * 64 + e;
    float a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = 0.f;
    if (i < j.len) {
        const float* p = j.src + i;
        int s = sg;
        for (; s + 60 < j.nsplit; s += 64) {
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] += p[(long)(s + 4 * k) * j.stride];
        }
        for (int k = 0; s < j.nsplit; s += 4, ++k) a[k & 15] += p[(long)s * j.stride];
    }
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += a[k];
    red[sg][e] = t;
    __syncthreads();
    if (sg == 0 && i < j.len) j.dst[i] = ((red[0][e] + red[1][e]) + (red[2][e] + red[3][e])) * j.scale;
}

// 2-D form with the (three) jobs in the kernarg block: blockIdx.y = job, blockIdx.x = 64-element chunk (blocks past a
// job's length exit).  Used by the stand-alone FrequencyLayer backward.
struct ReduceJobs3 { ReduceJob j[3]; };
__global__ void __launch_bounds__(ROW_THREADS)
multi_reduce3_kernel(const ReduceJobs3 J) {
    __shared__ float red[4][64];
    const ReduceJob j = J.j[blockIdx.y];
    if (blockIdx.x * 64 >= j.len) return;
    reduce_chunk(j, blockIdx.x, red);
}

// Flat form used by the training step: blockmap[b] = job << 16 | chunk for the nblocks useful blocks only (no empty
// blocks); when tick.state != null one extra block closes the optimisation step next to the reduction
// (multi_reduce_flat_kernel, below TickP).
// =============================================================================================
// device-resident step state (lets a captured hipGraph replay with fresh dropout masks / Adam t):
//   u64 state[0] = seed, [1] = forward-step counter, [2] = Adam t; f32 view of [3] = {lr/bc1, sqrt(bc2)};
//   u32 view of [4] = ticket of the Adam kernel; f64 view of [5], [6] = b1^t, b2^t
// =============================================================================================
__global__ void step_begin_kernel(uint64_t* state, long long* cursor, int advance) {
    state[1] += 1;
    if (cursor) *cursor += advance;       // batch cursor of the device-resident sample table
}

// batch assembly on the device: ids[b,:] = table[perm[cursor + b], :], answers[b] = ans_table[perm[cursor + b]]
// (replaces RandomSampler + DataLoader collation, src/dataset.py:207-211)
__global__ void __launch_bounds__(ROW_THREADS)
gather_batch_kernel(const int64_t* __restrict__ table, const int64_t* __restrict__ ans_table,
                    const int64_t* __restrict__ perm, long n, const long long* __restrict__ cursor, int B, int L,
                    int64_t* __restrict__ ids, int64_t* __restrict__ ans) {
    const long base = *cursor;
    for (int i = blockIdx.x * ROW_THREADS + threadIdx.x; i < B * L; i += gridDim.x * ROW_THREADS) {
        const int b = i / L, t = i % L;
        long src = base + b;
        src = src < n ? perm[src] : 0;
        ids[i] = table[src * L + t];
        if (t == 0) ans[b] = ans_table[src];
    }
}

// Closing an optimisation step on the device (one 256-thread block): mean loss of the batch, Adam's t and bias
// corrections (b1^t, b2^t kept as running products: no pow()), next forward-step index (fresh dropout masks), batch
// cursor.  Runs as the extra last block of the gradient reduction (fused step) or as its own launch (bsarec_adam_step).
struct TickP {
    uint64_t* state;
    double lr, b1, b2; int adam;                           // adam != 0: advance t, publish {lr/bc1, sqrt(bc2)}
    const float* loss_rows; int B; float* loss_out;        // mean loss of the batch (loss_out null: not requested)
    long long* cursor; int advance;                        // batch cursor of the device-resident sample table
    int bump_step;                                         // state[1] += 1
};

__device__ __forceinline__ void tick_body(const TickP& tk, float* red /* [ROW_THREADS] */) {
    if (tk.loss_out) {
        float s = 0.f;
        for (int k = threadIdx.x; k < tk.B; k += ROW_THREADS) s += tk.loss_rows[k];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = ROW_THREADS / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        uint64_t* state = tk.state;
        if (tk.loss_out) tk.loss_out[0] = red[0] / (float)tk.B;
        if (tk.adam) {
            double* prod = reinterpret_cast<double*>(state + 5);
            const uint64_t t = state[2];
            const double p1 = (t == 0 ? 1.0 : prod[0]) * tk.b1, p2 = (t == 0 ? 1.0 : prod[1]) * tk.b2;
            prod[0] = p1; prod[1] = p2;
            state[2] = t + 1;
            float* f = reinterpret_cast<float*>(state + 3);
            f[0] = (float)(tk.lr / (1.0 - p1));
            f[1] = (float)sqrt(1.0 - p2);
        }
        if (tk.bump_step) state[1] += 1;
        if (tk.cursor) *tk.cursor += tk.advance;
    }
}

__global__ void __launch_bounds__(ROW_THREADS) adam_tick_kernel(TickP tk) {
    __shared__ float red[ROW_THREADS];
    tick_body(tk, red);
}

__global__ void __launch_bounds__(ROW_THREADS)
multi_reduce_flat_kernel(const ReduceJob* __restrict__ jobs, const int* __restrict__ blockmap, int nblocks, TickP tk) {
    __shared__ float red[ROW_THREADS];
    if ((int)blockIdx.x >= nblocks) { tick_body(tk, red); return; }
    const int bm = blockmap[blockIdx.x];
    reduce_chunk(jobs[bm >> 16], bm & 0xFFFF, reinterpret_cast<float(*)[64]>(red));
}

// K9 fused Adam over the flat parameter arena (torch.optim.Adam semantics, src/trainers.py:27-28):
//   g += wd*w; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; w -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// shadow != null (cfg.storage = 1): the rounded parameter also goes to the bf16 mirror, for float4 groups >= shadow_from4
typedef __bf16 adam_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned adam_pk_bf16(float a, float b) {
    const adam_bf16x2 r = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, r);
}
// Gradient sources of the data-parallel step (bsarec_adam_t): g = grads (+ grads2 on the first n2 float4 groups, zeroed
// after use), or the sum over nsrc arenas in index order -- the one-shot peer-to-peer exchange: srcs[r] is rank r's
// arena, IPC-mapped over xGMI except the local one; read with system-scope (sc0 sc1) loads so that no cache of THIS GPU
// can serve a line from the previous step.
struct GradSrcs { int nsrc; const float* src[8]; float* g2; long n2_4; };
__device__ __forceinline__ f32x4 ld4_sys(const float* p) {
    // two 8-byte relaxed system-scope atomic loads (global_load_dwordx2 ... sc0 sc1): loads the compiler itself tracks
    // (an inline-asm load would need a hand-placed s_waitcnt that the scheduler is free to move the uses across)
    const uint64_t* q = reinterpret_cast<const uint64_t*>(p);
    const uint64_t a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint64_t b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return f32x4{__builtin_bit_cast(float, (uint32_t)a), __builtin_bit_cast(float, (uint32_t)(a >> 32)),
                 __builtin_bit_cast(float, (uint32_t)b), __builtin_bit_cast(float, (uint32_t)(b >> 32))};
}
__global__ void __launch_bounds__(ROW_THREADS)
adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n4,
            const uint64_t* __restrict__ state, float b1, float b2, float eps, float wd, float gscale,
            unsigned short* __restrict__ shadow, long shadow_from4, const GradSrcs S) {
    const float* f = reinterpret_cast<const float*>(state + 3);
    const float step_size = f[0], bc2s = f[1];
    for (long i = (long)blockIdx.x * ROW_THREADS + threadIdx.x; i < n4; i += (long)gridDim.x * ROW_THREADS) {
        f32x4 gi;
        if (S.nsrc > 0) {
            f32x4 part[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) if (r < S.nsrc) part[r] = ld4_sys(S.src[r] + 4 * i);     // all peers in flight
            gi = part[0];
#pragma unroll
            for (int r = 1; r < 8; ++r) if (r < S.nsrc) gi += part[r];                            // rank order on every rank
        } else {
            gi = ld4(g + 4 * i);
            if (S.g2 && i < S.n2_4) { gi += ld4(S.g2 + 4 * i); st4(S.g2 + 4 * i, f32x4{0, 0, 0, 0}); }
        }
        gi = gi * gscale;
        f32x4 wi = ld4(w + 4 * i), mi = ld4(m + 4 * i), vi = ld4(v + 4 * i);
        if (wd != 0.f) gi += wd * wi;
        mi = b1 * mi + (1.0f - b1) * gi;
        vi = b2 * vi + (1.0f - b2) * gi * gi;
#pragma unroll
        for (int k = 0; k < 4; ++k) wi[k] -= step_size * (mi[k] / (sqrtf(vi[k]) / bc2s + eps));
        st4(w + 4 * i, wi); st4(m + 4 * i, mi); st4(v + 4 * i, vi);
        if (shadow && i >= shadow_from4)
            *reinterpret_cast<uint2*>(shadow + 4 * i) = make_uint2(adam_pk_bf16(wi.x, wi.y), adam_pk_bf16(wi.z, wi.w));
    }
}

// The single-GPU step's last kernel: the final gradient reduction and Adam in ONE launch.  Blocks [0, nblocks) sum one
// 64-element chunk of a reduction job (all split-K slabs / LayerNorm / beta / position partials), store the gradient
// and update those 64 parameters on the spot; the remaining blocks update the one tensor no job produces -- the item
// table, whose gradient (dense logits part + lookup scatter) was finished by earlier kernels.  Adam's t / bias
// corrections were advanced by the step tick in the PREVIOUS kernel (dw_direct_kernel's extra block), so `state` is
// read-only here.  Same arithmetic, element for element, as adam_kernel.
struct AdamFuseP {
    float *w, *g, *m, *v; float b1, b2, eps, wd;
    unsigned short* shadow; long shadow_from;
    long item_off, item_n4;            // the item table inside the flat arena: element offset, float4 count
};
__global__ void __launch_bounds__(ROW_THREADS)
reduce_adam_kernel(const ReduceJob* __restrict__ jobs, const int* __restrict__ blockmap, int nblocks,
                   const uint64_t* __restrict__ state, const AdamFuseP A) {
    __shared__ float red[ROW_THREADS];
    const float* f = reinterpret_cast<const float*>(state + 3);
    const float step_size = f[0], bc2s = f[1];
    if ((int)blockIdx.x < nblocks) {
        const int bm = blockmap[blockIdx.x];
        const ReduceJob j = jobs[bm >> 16];
        float (*r4)[64] = reinterpret_cast<float(*)[64]>(red);
        const int e = threadIdx.x & 63, sg = threadIdx.x >> 6, i = (bm & 0xFFFF) * 64 + e;
        float a[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = 0.f;
        if (i < j.len) {
            const float* p = j.src + i;
            int s = sg;
            for (; s + 60 < j.nsplit; s += 64) {
#pragma unroll
                for (int k = 0; k < 16; ++k) a[k] += p[(long)(s + 4 * k) * j.stride];
            }
            for (int k = 0; s < j.nsplit; s += 4, ++k) a[k & 15] += p[(long)s * j.stride];
        }
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += a[k];
        r4[sg][e] = t;
        __syncthreads();
        if (sg == 0 && i < j.len) {
            const float gi0 = ((r4[0][e] + r4[1][e]) + (r4[2][e] + r4[3][e])) * j.scale;
            j.dst[i] = gi0;
            const long o = (j.dst - A.g) + i;
            float wi = A.w[o], mi = A.m[o], vi = A.v[o], gi = gi0;
            if (A.wd != 0.f) gi += A.wd * wi;
            mi = A.b1 * mi + (1.0f - A.b1) * gi;
            vi = A.b2 * vi + (1.0f - A.b2) * gi * gi;
            wi -= step_size * (mi / (sqrtf(vi) / bc2s + A.eps));
            A.w[o] = wi; A.m[o] = mi; A.v[o] = vi;
            if (A.shadow && o >= A.shadow_from) A.shadow[o] = (unsigned short)(adam_pk_bf16(wi, 0.f) & 0xFFFFu);
        }
        return;
    }
    const long nb = gridDim.x - nblocks;
    for (long i = (long)(blockIdx.x - nblocks) * ROW_THREADS + threadIdx.x; i < A.item_n4; i += nb * ROW_THREADS) {
        const long o = A.item_off + 4 * i;
        f32x4 wi = ld4(A.w + o), gi = ld4(A.g + o), mi = ld4(A.m + o), vi = ld4(A.v + o);
        if (A.wd != 0.f) gi += A.wd * wi;
        mi = A.b1 * mi + (1.0f - A.b1) * gi;
        vi = A.b2 * vi + (1.0f - A.b2) * gi * gi;
#pragma unroll
        for (int k = 0; k < 4; ++k) wi[k] -= step_size * (mi[k] / (sqrtf(vi[k]) / bc2s + A.eps));
        st4(A.w + o, wi); st4(A.m + o, mi); st4(A.v + o, vi);
        if (A.shadow && o >= A.shadow_from)
            *reinterpret_cast<uint2*>(A.shadow + o) = make_uint2(adam_pk_bf16(wi.x, wi.y), adam_pk_bf16(wi.z, wi.w));
    }
}

// fp32 -> bf16 (round to nearest even) for up to six tensors, jobs in the kernarg block: blockIdx.y = tensor
struct CastJobs6 { const float* src[6]; unsigned short* dst[6]; long n4[6]; };
__global__ void __launch_bounds__(ROW_THREADS)
cast_bf16_kernel(const CastJobs6 J) {
    const float* src = J.src[blockIdx.y];
    unsigned short* dst = J.dst[blockIdx.y];
    const long i = (long)blockIdx.x * ROW_THREADS + threadIdx.x;
    if (i >= J.n4[blockIdx.y]) return;
    const f32x4 x = ld4(src + 4 * i);
    *reinterpret_cast<uint2*>(dst + 4 * i) = make_uint2(adam_pk_bf16(x.x, x.y), adam_pk_bf16(x.z, x.w));
}

__global__ void __launch_bounds__(ROW_THREADS)
loss_mean_kernel(const float* __restrict__ rows, int B, float* __restrict__ out) {
    __shared__ float red[ROW_THREADS];
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += ROW_THREADS) s += rows[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = ROW_THREADS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] / (float)B;
}

// =============================================================================================
// Sibling model SASRec: BCE head on one positive and one negative item at the last position
// (src/model/sasrec.py:41-63).  loss = mean_{pos != 0} softplus(-x_pos) + mean_{pos != 0} softplus(x_neg),
// x = <h_last, E[item]>.  One workgroup: per-row logits, the count of kept rows, the loss and the per-row
// gradient coefficients coef[0][b] = d loss / d x_pos, coef[1][b] = d loss / d x_neg.
// =============================================================================================
__global__ void __launch_bounds__(ROW_THREADS)
bce_rows_kernel(const float* __restrict__ hlast, long hstride, const float* __restrict__ E, const int64_t* __restrict__ pos,
                const int64_t* __restrict__ neg, int B, int d, int V, float* __restrict__ coef, float* __restrict__ loss_out,
                int logsig /* 1: FMLPRec's -log(sigmoid + 1e-24) form over all rows (src/model/fmlprec.py:56-60) */) {
    __shared__ float red[2][ROW_THREADS];
    float lsum = 0.f, cnt = 0.f;
    for (int b = threadIdx.x; b < B; b += ROW_THREADS) {
        int ip = (int)pos[b], in = (int)neg[b];
        const bool keep = logsig || ip != 0;
        ip = ip < 0 ? 0 : (ip >= V ? V - 1 : ip); in = in < 0 ? 0 : (in >= V ? V - 1 : in);
        const float* h = hlast + (long)b * hstride;
        float xp = 0.f, xn = 0.f;
        for (int c = 0; c < d; c += 4) {
            const f32x4 hv = ld4(h + c), ep = ld4(E + (long)ip * d + c), en = ld4(E + (long)in * d + c);
            xp += hv.x * ep.x + hv.y * ep.y + hv.z * ep.z + hv.w * ep.w;
            xn += hv.x * en.x + hv.y * en.y + hv.z * en.z + hv.w * en.w;
        }
        // softplus(z) = max(z, 0) + log1p(exp(-|z|))
        const float sp = fmaxf(-xp, 0.f) + log1pf(expf(-fabsf(xp))), sn = fmaxf(xn, 0.f) + log1pf(expf(-fabsf(xn)));
        if (logsig) {
            const float s1 = 1.0f / (1.0f + expf(-xp)), s2 = 1.0f / (1.0f + expf(-xn));
            lsum += -logf(s1 + 1e-24f) - logf(1.0f - s2 + 1e-24f); cnt += 1.f;
            coef[b] = -(s1 * (1.0f - s1)) / (s1 + 1e-24f);
            coef[B + b] = (s2 * (1.0f - s2)) / (1.0f - s2 + 1e-24f);
            continue;
        }
        if (keep) { lsum += sp + sn; cnt += 1.f; }
        coef[b] = keep ? -1.0f / (1.0f + expf(xp)) : 0.f;            // -sigmoid(-xp)
        coef[B + b] = keep ? 1.0f / (1.0f + expf(-xn)) : 0.f;         //  sigmoid(xn)
    }
    red[0][threadIdx.x] = lsum; red[1][threadIdx.x] = cnt;
    __syncthreads();
    for (int o = ROW_THREADS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    const float n = fmaxf(red[1][0], 1.f);
    if (threadIdx.x == 0) loss_out[0] = red[0][0] / n;
    for (int b = threadIdx.x; b < 2 * B; b += ROW_THREADS) coef[b] /= n;
}

// backward of the head: d(h_last)[b] = gp E[pos] + gn E[neg] -> slab 0 of the split-K layout the block backward
// reads; dE[pos] += gp h_last, dE[neg] += gn h_last (dE zeroed by the caller; float atomics, one row per wave)
__global__ void __launch_bounds__(64)
bce_bwd_kernel(const float* __restrict__ hlast, long hstride, const float* __restrict__ E, const int64_t* __restrict__ pos,
               const int64_t* __restrict__ neg, const float* __restrict__ coef, int B, int d, int V, float* __restrict__ dh,
               float* __restrict__ dE) {
    const int b = blockIdx.x;
    int ip = (int)pos[b], in = (int)neg[b];
    ip = ip < 0 ? 0 : (ip >= V ? V - 1 : ip); in = in < 0 ? 0 : (in >= V ? V - 1 : in);
    const float gp = coef[b], gn = coef[B + b];
    for (int c = threadIdx.x; c < d; c += 64) {
        const float h = hlast[(long)b * hstride + c];
        dh[(long)b * d + c] = gp * E[(long)ip * d + c] + gn * E[(long)in * d + c];
        if (gp != 0.f) unsafeAtomicAdd(dE + (long)ip * d + c, gp * h);
        if (gn != 0.f) unsafeAtomicAdd(dE + (long)in * d + c, gn * h);
    }
}


// =============================================================================================
// Evaluation: scores of the items a user has already interacted with are set to 0 -- not -inf -- before the top-k
// (src/trainers.py:134: rating_pred[train_matrix[user].toarray() > 0] = 0).  One workgroup per batch row walks the
// user's CSR row on the device.
// =============================================================================================
__global__ void __launch_bounds__(ROW_THREADS)
mask_seen_kernel(float* __restrict__ scores, long ld, const int64_t* __restrict__ users, const int64_t* __restrict__ indptr,
                 const int64_t* __restrict__ indices) {
    const long u = users[blockIdx.x];
    const long j0 = indptr[u], j1 = indptr[u + 1];
    float* row = scores + (long)blockIdx.x * ld;
    for (long j = j0 + threadIdx.x; j < j1; j += ROW_THREADS) row[indices[j]] = 0.f;
}

// y += x on [n4 * 4] elements: the upstream gradient of an INTERMEDIATE layer output (forward(all_sequence_output=True),
// src/model/bsarec.py:46-54) joins the gradient that flows down from the layers above.  y: the inter-block gradient
// buffer (fp32, or bf16 under storage = 1), x: the caller's fp32 tensor.
__global__ void __launch_bounds__(ROW_THREADS)
grad_join_kernel(float* __restrict__ y, const float* __restrict__ x, long n4, int y_bf16) {
    for (long i = (long)blockIdx.x * ROW_THREADS + threadIdx.x; i < n4; i += (long)gridDim.x * ROW_THREADS) {
        const float4 b = reinterpret_cast<const float4*>(x)[i];
        if (y_bf16) {
            uint2 r = reinterpret_cast<uint2*>(y)[i];
            const float a0 = __builtin_bit_cast(float, r.x << 16) + b.x, a1 = __builtin_bit_cast(float, r.x & 0xFFFF0000u) + b.y;
            const float a2 = __builtin_bit_cast(float, r.y << 16) + b.z, a3 = __builtin_bit_cast(float, r.y & 0xFFFF0000u) + b.w;
            typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
            const bf2 p0 = {(__bf16)a0, (__bf16)a1}, p1 = {(__bf16)a2, (__bf16)a3};
            r.x = __builtin_bit_cast(unsigned, p0); r.y = __builtin_bit_cast(unsigned, p1);
            reinterpret_cast<uint2*>(y)[i] = r;
        } else {
            float4 a = reinterpret_cast<float4*>(y)[i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            reinterpret_cast<float4*>(y)[i] = a;
        }
    }
}

// Top-k of every score row with the seen items zeroed first: the body of the reference's evaluation loop for one batch
// (src/trainers.py:134-149: rating_pred[train_matrix[user] > 0] = 0, np.argpartition(..., -20), argsort of the 20) in ONE
// launch, one workgroup per user.  (1) the CSR row of the user is written as zeros into the score row (as the reference
// does -- the caller may still read the masked scores); (2) every thread keeps the best k of its strided share of the row
// as a sorted list in LDS (one compare rejects almost every candidate); (3) k rounds of a workgroup arg-max over the 256
// list heads produce the result in descending order.  Ties (exact equal scores, e.g. the zeros of seen items when fewer
// than k scores are positive) go to the smaller item id; the reference's argpartition leaves their order unspecified and
// they cannot change HR / NDCG (the answer is never a seen item).  k <= TOPK_MAX (48 KB of list storage in LDS).
#define TOPK_MAX 24
__device__ __forceinline__ bool topk_better(float v, long i, float w, long j) { return v > w || (v == w && i < j); }
__global__ void __launch_bounds__(ROW_THREADS)
topk_seen_kernel(float* __restrict__ scores, long ld, int V, const int64_t* __restrict__ users, const int64_t* __restrict__ indptr,
                 const int64_t* __restrict__ indices, int k, int64_t* __restrict__ out_idx, float* __restrict__ out_val) {
    __shared__ float lv[TOPK_MAX][ROW_THREADS];          // [rank][thread]: conflict-free for "every thread touches its rank r"
    __shared__ int li[TOPK_MAX][ROW_THREADS];
    __shared__ float rv[ROW_THREADS / 64];
    __shared__ int ri[ROW_THREADS / 64], rt[ROW_THREADS / 64];
    const int tid = threadIdx.x;
    float* row = scores + (long)blockIdx.x * ld;
    if (indptr) {
        const long u = users[blockIdx.x];
        const long j0 = indptr[u], j1 = indptr[u + 1];
        for (long j = j0 + tid; j < j1; j += ROW_THREADS) { const long it = indices[j]; if (it >= 0 && it < V) row[it] = 0.f; }
        __syncthreads();                                 // (drains the stores: the scan below reads the zeros)
    }
    for (int r = 0; r < k; ++r) { lv[r][tid] = -INFINITY; li[r][tid] = 0x7fffffff; }
    float vmin = -INFINITY; int imin = 0x7fffffff;       // this thread's current k-th best
    for (int j = tid; j < V; j += ROW_THREADS) {
        const float v = row[j];
        if (!topk_better(v, j, vmin, imin)) continue;
        int r = k - 1;                                   // insertion into the sorted list (descending)
        while (r > 0 && topk_better(v, j, lv[r - 1][tid], li[r - 1][tid])) { lv[r][tid] = lv[r - 1][tid]; li[r][tid] = li[r - 1][tid]; --r; }
        lv[r][tid] = v; li[r][tid] = j;
        vmin = lv[k - 1][tid]; imin = li[k - 1][tid];
    }
    int head = 0;                                        // next unused rank of this thread's list
    for (int r = 0; r < k; ++r) {
        float v = head < k ? lv[head][tid] : -INFINITY;
        int i = head < k ? li[head][tid] : 0x7fffffff, t = tid;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {        // wave arg-max (value desc, item id asc)
            const float v2 = __shfl_xor(v, off, 64); const int i2 = __shfl_xor(i, off, 64), t2 = __shfl_xor(t, off, 64);
            if (topk_better(v2, i2, v, i)) { v = v2; i = i2; t = t2; }
        }
        if ((tid & 63) == 0) { rv[tid >> 6] = v; ri[tid >> 6] = i; rt[tid >> 6] = t; }
        __syncthreads();
        float bv = rv[0]; int bi = ri[0], bt = rt[0];
#pragma unroll
        for (int w = 1; w < ROW_THREADS / 64; ++w) if (topk_better(rv[w], ri[w], bv, bi)) { bv = rv[w]; bi = ri[w]; bt = rt[w]; }
        if (tid == bt) ++head;
        if (tid == 0) { out_idx[(long)blockIdx.x * k + r] = bi; if (out_val) out_val[(long)blockIdx.x * k + r] = bv; }
        __syncthreads();
    }
}
