// The TOP BSARecBlock of the training step at the fused shape (hidden = 64, L <= 64), restricted to what the loss
// can see.  calculate_loss reads only position L-1 of the last layer (src/model/bsarec.py:32), so of the top block
// only that row has to exist -- but it still attends to every position: K and V of all rows are needed, everything
// else (frequency layer, query, softmax row, dense, both LayerNorms, feed-forward) is a single row per sequence.
// The backward has the mirror structure: the upstream gradient is one row, so dK, dV are rank-1 per head
// (dK_j = ds_j q_last, dV_j = Drop(p)_j dC_last) and the whole input gradient is a handful of vector products:
//   dX_j = sum_h ds_hj (q_h Wk_h) + sum_h Drop(p)_hj (dC_h Wv_h) + [j = L-1] (dq Wq + dzA + dzF + beta^2 dF)
//          + P[j][L-1] ((1 - beta^2) dF)                         (P = the low-pass projector of the FrequencyLayer)
// Results are those of the full block kernels (same formulas, same dropout stream, same buffers -- only the rows
// the loss depends on are produced); the exact zero structure is used, nothing is approximated.  SURVEY C.6 / §8d:
// the algorithmic FLOP counts reported by bench.py stay the un-pruned ones.
// One workgroup of 256 threads per sequence; matrix-vector products by VALU with the weight rows streamed from L2.
#pragma once
#include "fused_layer.h"

struct TopFwdP {
    const float* X; float* Xout;
    const float *sqrt_beta, *f_g, *f_b, *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo, *a_g, *a_b, *w1, *b1, *w2, *b2, *ff_g, *ff_b;
    const float* tw; const int* ids32;
    float *xhat_f, *rstd_f, *q, *k, *v, *probs, *ctx, *xhat_a, *rstd_a, *hmix, *u, *xhat_ff, *rstd_ff;
    float* low;                    // [B, L, 64] buffer whose last rows receive the low-pass component (for d sqrt_beta)
    int L, Lp, cb, heads;
    float alpha, oma, eps;
    DropP drop_f, drop_p, drop_o, drop_ff;
    long long* stamps;             // diagnostic: per-step shader clock of workgroup 0 (null in production)
    // bf16 storage (BF instantiation): wk_sh / wv_sh = bf16 shadow of the key / value weights for the MFMA projections of
    // all rows (as the full block kernel reads them); the one-row vector products keep the fp32 masters
    const float *wk_sh, *wv_sh;
};

// keep-mask x scale of ONE element (common.h: DropSeed, drop_mult4)
__device__ __forceinline__ float drop_mult1(const DropP& d, const DropSeed& sd, uint64_t e) {
    if (d.thresh == 0) return d.scale;
    const uint64_t grp = e >> 2;
    const uint4 w = philox4x32_10((uint32_t)grp, (uint32_t)(grp >> 32), d.site, sd.step, sd.k0, sd.k1);
    const int j = (int)(e & 3);
    const uint32_t r = j == 0 ? w.x : j == 1 ? w.y : j == 2 ? w.z : w.w;
    return r >= d.thresh ? d.scale : 0.f;
}

// y[n] = sum_k W[n][k] x[k]  (nn.Linear forward, W row-major [*][ldw]); KS adjacent lanes share one output.
// Split in load / dot so that the weight rows of a later step are requested early and their L2 round trip is hidden.
template <int K, int KS>
__device__ __forceinline__ void gemv_rows_load(const float* __restrict__ W, int ldw, int n, int slice, f32x4 (&wv)[K / KS / 4]) {
    const float* w = W + (long)n * ldw + slice * (K / KS);
#pragma unroll
    for (int i = 0; i < K / KS / 4; ++i) wv[i] = gld4(w + 4 * i);
}
template <int K, int KS>
__device__ __forceinline__ float gemv_rows_dot(const f32x4 (&wv)[K / KS / 4], const float* __restrict__ sx, int slice) {
    const float* x = sx + slice * (K / KS);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < K / KS / 4; ++i) {
        const f32x4 xv = ld4(x + 4 * i);
        acc += wv[i].x * xv.x + wv[i].y * xv.y + wv[i].z * xv.z + wv[i].w * xv.w;
    }
    if (KS >= 2) acc += dpp_mov<0xB1>(acc);
    if (KS >= 4) acc += dpp_mov<0x4E>(acc);
    if (KS >= 8) acc += dpp_mov<0x141>(acc);
    if (KS >= 16) acc += dpp_mov<0x140>(acc);
    return acc;
}

// y[i] = sum_{k in [k0, k0+KC)} x[k] W[k][i]  (the transposed product of the backward; lanes = consecutive i)
template <int KC>
__device__ __forceinline__ void gemv_cols_load(const float* __restrict__ W, int ldw, int k0, int i, float (&wv)[KC]) {
    const float* w = W + (long)k0 * ldw + i;
#pragma unroll
    for (int k = 0; k < KC; ++k) wv[k] = gld(w + (long)k * ldw);
}
template <int KC>
__device__ __forceinline__ float gemv_cols_dot(const float (&wv)[KC], const float* __restrict__ sx, int k0) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < KC; ++k) acc += sx[k0 + k] * wv[k];
    return acc;
}

// The same transposed product with FOUR adjacent outputs per lane: lane (g = lane & 15, kq = lane >> 4) of a wave covers
// outputs i0 + 4 g .. + 3 over the inner indices k0 + 16 kq .. + 15, 16 dwordx4 loads (a wave instruction reads four 256-byte
// row segments) instead of 64 dword loads for the same 64 x 64 block of W; the four kq partial sums of an output meet by
// two lane swaps (quad_rows_sum).  Every lane returns the complete sums of its four outputs.
__device__ __forceinline__ void gemv_cols4_load(const float* __restrict__ W, int ldw, int k0, int i0, f32x4 (&wv)[16]) {
    const int lane = threadIdx.x & 63;
    const float* w = W + (long)(k0 + 16 * (lane >> 4)) * ldw + i0 + 4 * (lane & 15);
#pragma unroll
    for (int k = 0; k < 16; ++k) wv[k] = gld4(w + (long)k * ldw);
}
__device__ __forceinline__ f32x4 gemv_cols4_dot(const f32x4 (&wv)[16], const float* __restrict__ sx, int k0) {
    const int lane = threadIdx.x & 63;
    const float* x = sx + k0 + 16 * (lane >> 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; k += 4) {
        const f32x4 xv = ld4(x + k);
        acc += wv[k] * xv.x + wv[k + 1] * xv.y + wv[k + 2] * xv.z + wv[k + 3] * xv.w;
    }
    return f32x4{quad_rows_sum(acc.x), quad_rows_sum(acc.y), quad_rows_sum(acc.z), quad_rows_sum(acc.w)};
}

// LayerNorm of one 64-wide row held one column per lane of a wave
__device__ __forceinline__ void ln_row(float v, float eps, float& xhat, float& rstd) {
    const float mean = group_sum<64>(v) * (1.0f / 64.0f);
    const float dl = v - mean;
    const float var = group_sum<64>(dl * dl) * (1.0f / 64.0f);
    rstd = 1.0f / sqrtf(var + eps);
    xhat = dl * rstd;
}

// The forward of the top block in two pieces so that it can also run as the TAIL of the block below it
// (fused_layer_fwd_kernel<.., TAIL = true>: the lower block's output tile, ids and twiddle table are already in LDS, waves
// 4..7 of that kernel have exited, and the loads of top_fwd_prefetch were issued before the lower block's last row pass).
// Parameters are read from the kernarg segment at byte offset KOFF (0 for the stand-alone kernel; behind FusedFwdP for
// the tail) at their point of use -- see KARG in fused_layer.h.
#define TP(f) kernarg_field<decltype(TopFwdP::f)>(KOFF + (unsigned)offsetof(TopFwdP, f))
#define TSTAMP(i) do { long long* st_ = TP(stamps); if (st_ && blockIdx.x == 0 && threadIdx.x == 0) st_[i] = clock64(); } while (0)

template <bool BF>
struct TopFwdRegs {
    WFrag<BF, 64> wA, wB;                  // K / V weight fragments of this wave's two 32 x 32 tiles
    f32x4 wq4[4], wo4[4];                  // weight rows of the one-row query / dense products
    float bias_k, bias_v, bq_n, bo_n, b1_n, b2_n;
    float c_beta, c_fg, c_fb, c_ag, c_ab, c_ffg, c_ffb;
};

template <bool BF, unsigned KOFF, bool KV>
__device__ __forceinline__ void top_fwd_prefetch(TopFwdRegs<BF>& R) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    // Loads return in issue order (vmcnt): the small per-column vectors of the row steps go first so that no later step
    // waits behind a bulk weight prefetch for them.
    R.c_beta = gld(TP(sqrt_beta) + lane); R.c_fg = gld(TP(f_g) + lane); R.c_fb = gld(TP(f_b) + lane);
    R.c_ag = gld(TP(a_g) + lane); R.c_ab = gld(TP(a_b) + lane); R.c_ffg = gld(TP(ff_g) + lane); R.c_ffb = gld(TP(ff_b) + lane);
    const int wn = wave & 1, col = wn * 32 + l31;
    const int KH = (BF ? 8 : 4) * half;
    const long wrow = (long)col * 64 + KH;
    if constexpr (KV) {                    // (fused tail: waves 4..7 hold these, top_fwd_help_prefetch)
        load_w<BF, 64>(BF ? TP(wk_sh) : TP(wk), wrow, R.wA);
        load_w<BF, 64>(BF ? TP(wv_sh) : TP(wv), wrow, R.wB);
        R.bias_k = gld(TP(bk) + col); R.bias_v = gld(TP(bv) + col);
    }
    const int on = tid >> 2, osl = tid & 3;
    gemv_rows_load<64, 4>(TP(wq), 64, on, osl, R.wq4);
    gemv_rows_load<64, 4>(TP(wo), 64, on, osl, R.wo4);
    R.bq_n = gld(TP(bq) + on); R.bo_n = gld(TP(bo) + on); R.b1_n = gld(TP(b1) + tid); R.b2_n = gld(TP(b2) + on);
}

// sX: x tile (rows >= L zero), sIds, sTab filled by the caller; everything else is scratch of this function.
// 256 threads (waves 0..3 of the workgroup; any other wave must have exited: the barriers count the live waves).
template <int DH, bool BF, unsigned KOFF, bool HELPED>
__device__ __forceinline__ void top_fwd_rest(const TopFwdRegs<BF>& R, const DropSeed& dseed, float* sX, float* sK, float* sV,
                                             float* sPart, float* sTab, float* sSpec, float* sVec, const int* sIds) {
    float* sQ = sVec; float* sPd = sVec + 64; float* sCtx = sVec + 320; float* sHm = sVec + 384; float* sG = sVec + 448;
    float* sDsp = sVec + 704;
    float* sMul = sVec + 768;                  // HELPED: [3][64] dropout multipliers of row L-1 (sites f, o, ff)
    float* sMp = sVec + 960;                   // HELPED: [4][64] attention-dropout multipliers of the last query
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int L = TP(L), Lp = TP(Lp), heads = TP(heads), cb = TP(cb);
    const int b = blockIdx.x, tl = L - 1;
    const long tok0 = (long)b * L, el = (tok0 + tl) * 64;
    const int wm = wave >> 1, wn = wave & 1, col = wn * 32 + l31;
    const int KH = (BF ? 8 : 4) * half;
    const int on = tid >> 2, osl = tid & 3;
    TSTAMP(1);
    // kernarg fields in batches, a stage ahead of their use (see top_bwd_body)
    float* const Pq = TP(q); float* const Pk = TP(k); float* const Pv = TP(v);
    float* const k_low = TP(low); const DropP d_f = TP(drop_f); const float k_eps = TP(eps);
    float* const k_xhat_f = TP(xhat_f); float* const k_rstd_f = TP(rstd_f);
    const float* const pW1 = TP(w1); const float* const pW2 = TP(w2);

    // ---- K, V projections of all rows (MFMA; HELPED: by waves 4..7, top_fwd_help), spectrum of x (VALU)
    if constexpr (!HELPED) {
        const int arow = (wm * 32 + l31) * FS + KH;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        mma_w<BF, 64>(sX + arow, R.wA, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) sK[(wm * 32 + rho(r) + 4 * half) * FS + col] = acc[r] + R.bias_k;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        mma_w<BF, 64>(sX + arow, R.wB, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) sV[(wm * 32 + rho(r) + 4 * half) * FS + col] = acc[r] + R.bias_v;
    }
#ifdef BSAREC_FINE_STAMPS
    TSTAMP(9);
#endif
    if constexpr (!HELPED) {
        auto xsrc = [&](int, int t, int lc) { return ld4(sX + t * FS + lc); };
        dft_spectrum_tab<1>(xsrc, L, cb, sTab, sSpec, sPart);        // ends with a barrier: sK / sV complete too
    } else {
        // Only the low-pass component of row L-1 is needed, and that is one projector row applied to the x tile:
        //   low[L-1][c] = sum_t P[t] x[t][c],  P[t] = (1/L) sum_k w_k cos(2 pi k (t - (L-1)) / L)   (w_0 = w_{L/2} = 1, else 2)
        // (the spectrum + synthesis of dft_spectrum_tab / lowpass_tab, contracted over the bins first).  Wave w takes rows
        // [16 w, 16 w + 16): lanes 0..15 evaluate P, each lane = one column sums its 16 rows with P read lane by lane.
        if (wave == 1) sMul[lane] = drop_mult1(d_f, dseed, (uint64_t)(el + lane));      // wave 0 needs this one first
        const int t = 16 * wave + (lane & 15);
        float pl = 0.f;
        if (t < L)
            for (int k = 0; k < cb; ++k) {
                const float w = (k == 0 || 2 * k == L) ? 1.0f : 2.0f;
                pl += w * (sTab[2 * (k * 64 + t)] * sTab[2 * (k * 64 + tl)] + sTab[2 * (k * 64 + t) + 1] * sTab[2 * (k * 64 + tl) + 1]);
            }
        pl *= 1.0f / (float)L;
        float a2[2] = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 16; ++j)
            a2[j & 1] += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pl), j)) * sX[(16 * wave + j) * FS + lane];
        sPart[wave * 64 + lane] = a2[0] + a2[1];
    }
#ifdef BSAREC_FINE_STAMPS
    TSTAMP(10);
#endif

    // q_last = x_last . Wq^T + bq  (4 lanes per output), and the k, v rows -> global (the backward reads them)
    {
        const float qv = gemv_rows_dot<64, 4>(R.wq4, sX + tl * FS, osl) + R.bq_n;
        if (osl == 0) { sQ[on] = qv; ast<BF>(Pq, el + on, qv); }
        if constexpr (!HELPED) {
            const int lr = tid >> 4, lc = (tid & 15) << 2;
            for (int r = lr; r < L; r += 16) {
                ast4<BF>(Pk, (tok0 + r) * 64 + lc, ld4(sK + r * FS + lc));
                ast4<BF>(Pv, (tok0 + r) * 64 + lc, ld4(sV + r * FS + lc));
            }
        }
    }
    if constexpr (HELPED) lds_barrier();           // (the partial sums; waves 4..7: after their V projection)
#ifdef BSAREC_FINE_STAMPS
    TSTAMP(11);
#endif
    // FrequencyLayer output of the last row (wave 0, lane = column):  src/model/bsarec.py:90-104
    if (wave == 0) {
        const int c = lane, c4 = c & ~3;
        float low;
        if constexpr (HELPED) low = (sPart[c] + sPart[64 + c]) + (sPart[128 + c] + sPart[192 + c]);
        else {
            const f32x4 low4 = lowpass_tab(sSpec, tl, c4, L, cb, sTab);
            low = (c & 3) == 0 ? low4.x : (c & 3) == 1 ? low4.y : (c & 3) == 2 ? low4.z : low4.w;
        }
        const float xv = sX[tl * FS + c];
        gst(k_low + el + c, low);
        const float bt = R.c_beta;
        const float f = low + bt * bt * (xv - low);
        const float v = f * (HELPED ? sMul[c] : drop_mult1(d_f, dseed, (uint64_t)(el + c))) + xv;
        float xh, rs;
        ln_row(v, k_eps, xh, rs);
        ast<BF>(k_xhat_f, el + c, xh);
        if (c == 0) gst(k_rstd_f + tok0 + tl, rs);
        sDsp[c] = R.c_fg * xh + R.c_fb;
    }
    if constexpr (HELPED) {
        // the other dropout masks of the one-row chain, on the waves that have nothing else to do in this step: wave 2 the
        // dense output's and the attention rows of heads 0, 2; wave 3 the feed-forward output's and heads 1, 3
        if (wave >= 2) {
            const DropP dd = wave == 2 ? TP(drop_o) : TP(drop_ff);
            sMul[(wave - 1) * 64 + lane] = drop_mult1(dd, dseed, (uint64_t)(el + lane));
            const DropP dp = TP(drop_p);
            for (int h = wave - 2; h < heads; h += 2) {
                const long pe = (((long)b * heads + h) * L + tl) * Lp;
                sMp[h * 64 + lane] = drop_mult1(dp, dseed, (uint64_t)(pe + lane));
            }
        }
    }
    float* const k_probs = TP(probs); const DropP d_p = TP(drop_p); float* const k_ctx = TP(ctx);
    lds_barrier();
    TSTAMP(2);
    // feed-forward weight rows.  Lane n reads ITS row, 32-64 cache lines per wave instruction: the ISSUE of the 32 loads
    // costs ~7 k cycles of the CU's address path and the wave cannot move on before its loads are issued.  dense_1's rows
    // are requested here; dense_2's (needed two steps later) by waves 1..3 while wave 0 runs the LayerNorm + mix row
    // alone, and by wave 0 right after it -- in the shadow of steps that leave the address path idle.
    f32x4 w1r[HELPED ? 1 : 16], w2r[HELPED ? 1 : 16];
    if constexpr (!HELPED) gemv_rows_load<64, 1>(pW1, 64, tid, 0, w1r);

    // ---- attention row of the last query: one wave per head, lane = key        src/model/_modules.py:118-135
    if (wave < heads) {
        const int head = wave, key = lane;
        float s = -INFINITY;
        if (key < L) {
            float acc = 0.f;
            const float* kr = sK + key * FS + head * DH;
            const float* qr = sQ + head * DH;
#pragma unroll
            for (int c = 0; c < DH; c += 4) {
                const f32x4 kv = ld4(kr + c), qv = ld4(qr + c);
                acc += kv.x * qv.x + kv.y * qv.y + kv.z * qv.z + kv.w * qv.w;
            }
            s = acc / sqrtf((float)DH) + (sIds[key] > 0 ? 0.0f : -10000.0f);      // key <= query always holds for the last row
        }
        const float mx = group_max<64>(s);
        const float e = key < L ? __expf(s - mx) : 0.f;
        const float p = e / group_sum<64>(e);
        const long pe = (((long)b * heads + head) * L + tl) * Lp;
        if (key < Lp) ast<BF>(k_probs, pe + key, p);
        sPd[head * 64 + key] = key < L ? p * (HELPED ? sMp[head * 64 + key] : drop_mult1(d_p, dseed, (uint64_t)(pe + key))) : 0.f;
    }
    const DropP d_o = TP(drop_o); float* const k_xhat_a = TP(xhat_a); float* const k_rstd_a = TP(rstd_a);
    const float k_alpha = TP(alpha), k_oma = TP(oma); float* const k_hmix = TP(hmix); float* const k_u = TP(u);
    lds_barrier();
    TSTAMP(3);
    if (tid < 64) {                               // ctx_last[c] = sum_j Drop(p)_j v_j[c]
        const int c = tid, head = c / DH;
        float acc = 0.f;
        float a4[4] = {0.f, 0.f, 0.f, 0.f};          // 64 padded keys (Drop(p) = 0 past L): fixed trip count, 4 chains
#pragma unroll 16
        for (int j = 0; j < 64; ++j) a4[j & 3] += sPd[head * 64 + j] * sV[j * FS + c];
        acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        sCtx[c] = acc;
        ast<BF>(k_ctx, el + c, acc);
    }
    lds_barrier();
    TSTAMP(4);

    // ---- dense + dropout + residual + LayerNorm + alpha mix (one row)
    {
        const float o = gemv_rows_dot<64, 4>(R.wo4, sCtx, osl) + R.bo_n;
        if (osl == 0) sG[on] = o;
    }
    lds_barrier();
    TSTAMP(5);
    if constexpr (!HELPED) { if (wave != 0) gemv_rows_load<256, 4>(pW2, 256, on, osl, w2r); }
    if (wave == 0) {
        const int c = lane;
        const float v = sG[c] * (HELPED ? sMul[64 + c] : drop_mult1(d_o, dseed, (uint64_t)(el + c))) + sX[tl * FS + c];
        float xh, rs;
        ln_row(v, k_eps, xh, rs);
        ast<BF>(k_xhat_a, el + c, xh);
        if (c == 0) gst(k_rstd_a + tok0 + tl, rs);
        const float a = R.c_ag * xh + R.c_ab;
        const float hm = k_alpha * sDsp[c] + k_oma * a;
        sHm[c] = hm;
        ast<BF>(k_hmix, el + c, hm);
        if constexpr (!HELPED) gemv_rows_load<256, 4>(pW2, 256, on, osl, w2r);
    }
    const DropP d_ff = TP(drop_ff); float* const k_xhat_ff = TP(xhat_ff); float* const k_rstd_ff = TP(rstd_ff); float* const k_Xout = TP(Xout);
    lds_barrier();
    TSTAMP(6);

    // ---- feed-forward (one row): u = hmix W1^T + b1 (one output per thread), y = gelu(u) W2^T + b2   (HELPED: waves 4..7)
    if constexpr (!HELPED) {
        const float u = gemv_rows_dot<64, 1>(w1r, sHm, 0) + R.b1_n;
        ast<BF>(k_u, (tok0 + tl) * 256 + tid, u);
        sG[tid] = gelu_f(u);
    }
    lds_barrier();
    TSTAMP(7);
    if constexpr (!HELPED) {
        const float y = gemv_rows_dot<256, 4>(w2r, sG, osl) + R.b2_n;
        if (osl == 0) sQ[on] = y;
    }
    lds_barrier();
    TSTAMP(8);
    if (wave == 0) {
        const int c = lane;
        const float v = sQ[c] * (HELPED ? sMul[128 + c] : drop_mult1(d_ff, dseed, (uint64_t)(el + c))) + sHm[c];
        float xh, rs;
        ln_row(v, k_eps, xh, rs);
        ast<BF>(k_xhat_ff, el + c, xh);
        if (c == 0) gst(k_rstd_ff + tok0 + tl, rs);
        gst(k_Xout + el + c, R.c_ffg * xh + R.c_ffb);           // the last layer's output is an fp32 tensor in every mode
    }
    TSTAMP(15);
}

// Waves 4..7 of fused_layer_fwd_kernel<.., TopFwdP> while waves 0..3 run top_fwd_rest<.., HELPED = true>.  They take
//   * the K and V projections of all rows (MFMA; 2 x 2 waves tile 64 tokens x 64 features), and the copy of the finished
//     tiles to global memory,
//   * the one-row feed-forward: dense_1 / dense_2 rows read as WHOLE rows (one wave instruction = 1 KB contiguous; a lane per
//     row costs 64 cache lines an instruction and ~7 k cycles of the CU's address path), requested four loads at a time
//     in the steps in between, each product reduced across lanes.
// Their barriers mirror top_fwd_rest<.., HELPED = true>'s one for one (eight).
template <bool BF>
struct TopFwdHelpRegs {
    WFrag<BF, 64> wA, wB;                  // K / V weight fragments of this wave's 32 x 32 tile
    float bias_k, bias_v;
};
template <bool BF, unsigned KOFF>
__device__ __forceinline__ void top_fwd_help_prefetch(TopFwdHelpRegs<BF>& H) {
    const int lane = threadIdx.x & 63, hw = ((int)threadIdx.x >> 6) - 4;
    const int l31 = lane & 31, half = lane >> 5;
    const int wn = hw & 1, col = wn * 32 + l31;
    const int KH = (BF ? 8 : 4) * half;
    load_w<BF, 64>(BF ? TP(wv_sh) : TP(wv), (long)col * 64 + KH, H.wB);
    load_w<BF, 64>(BF ? TP(wk_sh) : TP(wk), (long)col * 64 + KH, H.wA);
    H.bias_v = gld(TP(bv) + col); H.bias_k = gld(TP(bk) + col);
}
template <int DH, bool BF, unsigned KOFF>
__device__ __forceinline__ void top_fwd_help(const TopFwdHelpRegs<BF>& H, const float* sX, float* sK, float* sV, float* sVec) {
    float* sQ = sVec; const float* sHm = sVec + 384; float* sG = sVec + 448;
    const int t = (int)threadIdx.x - 256, lane = t & 63, hw = t >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int L = TP(L);
    const int b = blockIdx.x, tl = L - 1;
    const long tok0 = (long)b * L;
    const int wm = hw >> 1, wn = hw & 1, col = wn * 32 + l31;
    const int KH = (BF ? 8 : 4) * half;
    const float* const pW1 = TP(w1); const float* const pW2 = TP(w2);
    float* const Pk = TP(k); float* const Pv = TP(v); float* const k_u = TP(u);
    // ---- V, then K projection of all rows -- the 64 fp32 MFMAs occupy the SIMD's vector ALU for ~4.2 k cycles wherever they
    //      run; on THESE waves they run beside the other group's FrequencyLayer / query / LayerNorm row instead of ahead of it
    const int arow = (wm * 32 + l31) * FS + KH;
    {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        mma_w<BF, 64>(sX + arow, H.wB, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) sV[(wm * 32 + rho(r) + 4 * half) * FS + col] = acc[r] + H.bias_v;
    }
    lds_barrier();                                               // (the low-pass partial sums of the other group)
    {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        mma_w<BF, 64>(sX + arow, H.wA, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) sK[(wm * 32 + rho(r) + 4 * half) * FS + col] = acc[r] + H.bias_k;
    }
    lds_barrier();                                               // K, V complete: the attention row starts
    // K, V tiles -> global (the backward reads them)
    {
        const int lr = t >> 4, lc = (t & 15) << 2;
        for (int r = lr; r < L; r += 16) {
            ast4<BF>(Pk, (tok0 + r) * 64 + lc, ld4(sK + r * FS + lc));
            ast4<BF>(Pv, (tok0 + r) * 64 + lc, ld4(sV + r * FS + lc));
        }
    }
    // dense_1: wave hw owns units [64 hw, 64 hw + 64); load p = rows 4 p .. 4 p + 3 of them (lane l: row 4 p + (l >> 4), columns
    // 4 (l & 15) ..); dense_2: outputs [16 hw, 16 hw + 16), load p = row p whole (lane l: inner units 4 l ..)
    f32x4 w1r[16], w2r[16];
    const float* const w1p = pW1 + (long)(64 * hw) * 64 + 4 * lane;
    const float* const w2p = pW2 + (long)(16 * hw) * 256 + 4 * lane;
    auto w1_quarter = [&](int q) {
#pragma unroll
        for (int p = 4 * q; p < 4 * q + 4; ++p) w1r[p] = gld4(w1p + p * 256);
    };
    auto w2_quarter = [&](int q) {
#pragma unroll
        for (int p = 4 * q; p < 4 * q + 4; ++p) w2r[p] = gld4(w2p + p * 256);
    };
    const int my_unit = 64 * hw + 4 * (lane & 15) + (lane >> 4);       // the unit whose sum lane l keeps (p = l & 15)
    const float b1_n = gld(TP(b1) + my_unit);
    const float b2_n = gld(TP(b2) + 16 * hw + (lane & 15));
    w1_quarter(0); w1_quarter(1);
    lds_barrier();
    w1_quarter(2); w1_quarter(3);
    lds_barrier();
    w2_quarter(0);
    lds_barrier();
    w2_quarter(1); w2_quarter(2);
    lds_barrier();
    // ---- u = hmix W1^T + b1, gelu
    {
        const f32x4 hm = ld4(sHm + 4 * (lane & 15));
        float mine = 0.f;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const float part = group_sum<16>(w1r[p].x * hm.x + w1r[p].y * hm.y + w1r[p].z * hm.z + w1r[p].w * hm.w);
            mine = (lane & 15) == p ? part : mine;
        }
        const float u = mine + b1_n;
        ast<BF>(k_u, (tok0 + tl) * 256 + my_unit, u);
        sG[my_unit] = gelu_f(u);
    }
    w2_quarter(3);
    lds_barrier();
    // ---- y = gelu(u) W2^T + b2
    {
        const f32x4 g4 = ld4(sG + 4 * lane);
        float mine = 0.f;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const float part = group_sum<64>(w2r[p].x * g4.x + w2r[p].y * g4.y + w2r[p].z * g4.z + w2r[p].w * g4.w);
            mine = (lane & 15) == p ? part : mine;
        }
        if (lane < 16) sQ[16 * hw + lane] = mine + b2_n;
    }
    lds_barrier();
}

template <int DH, bool BF>
__global__ void __launch_bounds__(256)
top_fwd_kernel(const TopFwdP P_unused) {
    constexpr unsigned KOFF = 0;
    constexpr int TS = 64 * FS;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sX = sm;                       // x tile
    float* sK = sm + TS;                  // K tile [token][feature]
    float* sV = sm + 2 * TS;              // V tile [token][feature]
    float* sPart = sm + 3 * TS;           // DFT partials [16][4][2][64] = 8192 floats
    float* sTab = sPart + 8192;           // FUSED_MAX_CB * 128
    float* sSpec = sTab + FUSED_MAX_CB * 128;   // FUSED_MAX_CB * 128
    float* sVec = sSpec + FUSED_MAX_CB * 128;   // row vectors: [0]=q [64]=pd(h*64) [320]=ctx [384]=hmix [448]=g(256) [704]=dsp
    int* sIds = reinterpret_cast<int*>(sVec + 768);
    const int tid = threadIdx.x;
    TSTAMP(0);
    const int L = TP(L), cb = TP(cb);
    const long tok0 = (long)blockIdx.x * L;
    const DropSeed dseed = drop_seed(TP(drop_f));
    TopFwdRegs<BF> R;
    top_fwd_prefetch<BF, KOFF, true>(R);
    {
        const float* const X = TP(X);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int idx = tid + p * 256, r = idx >> 4, c4 = (idx & 15) << 2;
            f32x4 v = ald4<BF>(X, (tok0 + min(r, L - 1)) * 64 + c4);      // branch-free: rows past L re-read row L-1, zeroed
            if (r >= L) v = f32x4{0, 0, 0, 0};
            st4(sX + r * FS + c4, v);
        }
    }
    if (tid < 64) sIds[tid] = tid < L ? gldi(TP(ids32) + (tok0 + tid)) : 0;
    build_twiddle_table(TP(tw), L, cb, sTab);
    lds_barrier();
    top_fwd_rest<DH, BF, KOFF, false>(R, dseed, sX, sK, sV, sPart, sTab, sSpec, sVec, sIds);
}

static inline size_t top_fwd_smem_bytes() { return (size_t)(3 * 64 * FS + 8192 + 2 * FUSED_MAX_CB * 128 + 768 + 64) * 4; }


// =============================================================================================
// backward of the top block (see the header comment): no matrix product is left, only vector chains
// =============================================================================================
struct TopBwdP {
    float* dX; const float* X;
    const float *sqrt_beta, *f_g, *wq, *wk, *wv, *wo, *a_g, *w1, *w2, *ff_g;
    const float* tw;
    const float *xhat_f, *rstd_f, *q, *k, *v, *probs, *xhat_a, *rstd_a, *u, *xhat_ff, *rstd_ff, *low;
    const float* dh_slabs; int dh_nsplit; long dh_stride;   // dY of row L-1 = sum of the logits backward's split-K slabs [s][B][64]
    float *dT, *dU, *dO, *dq;                                // compact [B][64 | 256]: operands of the last-position weight-gradient products
    float *ak, *rk, *av, *rv;                                // [B * heads][64]: dWk = AK^T RK, dWv = AV^T RV
    float *pbk, *pbv;                                        // [B][64] key / value bias-gradient partials
    float *pg_ff, *pb_ff, *pg_a, *pb_a, *pg_f, *pb_f, *pbeta;   // [B][64] LayerNorm / sqrt_beta partials
    int L, Lp, cb, heads;
    float alpha, oma;
    DropP drop_f, drop_p, drop_o, drop_ff;
    long long* stamps;
};

// LayerNorm backward of one row, one column per lane: returns dz; g = dy * gamma
__device__ __forceinline__ float ln_row_bwd(float dy, float gamma, float xhat, float rstd) {
    const float g = dy * gamma;
    const float m1 = group_sum<64>(g) * (1.0f / 64.0f);
    const float m2 = group_sum<64>(g * xhat) * (1.0f / 64.0f);
    return rstd * (g - m1 - xhat * m2);
}

// The backward of the top block as a device function so that it can also run as the HEAD of the backward of the block
// below it (fused_layer_bwd_kernel<.., HEADP = TopBwdP>): waves 0..3 of that kernel run it, its input gradient tile stays
// in LDS (sDX) instead of a round trip through global memory, and waves 4..7 meanwhile stage the lower block's gelu'
// tile.  Those waves keep the workgroup's barrier count in step: they execute TOP_BWD_BARRIERS barriers of their own --
// EVERY lds_barrier() below is unconditional and top-level, and their number is that constant (tests/test_host_cpu.py
// counts them in this source).  sTab is built by the caller.  sDX = null: dX goes to global memory (stand-alone kernel).
#define TB(f) kernarg_field<decltype(TopBwdP::f)>(KOFF + (unsigned)offsetof(TopBwdP, f))
#define BSTAMP(i) do { long long* st_ = TB(stamps); if (st_ && blockIdx.x == 0 && threadIdx.x == 0) st_[i] = clock64(); } while (0)
static_assert(TOP_BWD_BARRIERS == 11, "barriers of top_bwd_body (fused_layer.h holds the constant)");

// What the chain that opens the top block's backward needs from memory (wave 0, one column per lane): requested at the
// very top of the kernel so that the ~4 k cycles these loads take (the slabs were written by the previous kernel, on other
// XCDs) pass under the twiddle-table build instead of in front of the first LayerNorm backward.
template <bool BF, unsigned KOFF>
__device__ __forceinline__ void top_bwd_prefetch(TopBwdRegs& R) {          // wave 0 only
    const int c = threadIdx.x & 63;
    const int L = TB(L), b = blockIdx.x, tl = L - 1;
    const long tok0 = (long)b * L, el = (tok0 + tl) * 64;
    // The slabs go through a buffer descriptor over exactly min(ns, 32) of them: one scalar offset per load instead of a
    // 64-bit address, and a slab index >= ns is out of range and reads as 0 (no select) -- every instruction of wave 0 here is
    // ~5 cycles of the step's critical path.
    const int ns = TB(dh_nsplit);
    const float* const slabs = TB(dh_slabs);
    const long stride = TB(dh_stride);
    const float* const k_xhat_ff = TB(xhat_ff); const float* const k_ff_g = TB(ff_g); const float* const k_rstd_ff = TB(rstd_ff);
    const float* const k_xhat_a = TB(xhat_a); const float* const k_xhat_f = TB(xhat_f); const float* const k_a_g = TB(a_g);
    const float* const k_f_g = TB(f_g); const float* const k_rstd_a = TB(rstd_a); const float* const k_rstd_f = TB(rstd_f);
    const float* const k_beta = TB(sqrt_beta); const float* const k_low = TB(low); const float* const k_q = TB(q);
    const float* const pX = TB(X);
    const long slab_bytes = (long)min(ns, 32) * stride * 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slabs, 0, (int)min(slab_bytes, 0x7fffffffL), 0x00020000);
    const int voff = (b * 64 + c) * 4, sstep = (int)(stride * 4);
#pragma unroll
    for (int sp = 0; sp < 32; ++sp) R.sl[sp] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, sp * sstep, 0));
    R.xh_ff = ald<BF>(k_xhat_ff, el + c); R.g_ff = gld(k_ff_g + c); R.rs_ff = gld(k_rstd_ff + tok0 + tl);
    R.xa = ald<BF>(k_xhat_a, el + c); R.xf = ald<BF>(k_xhat_f, el + c); R.g_a = gld(k_a_g + c); R.g_f = gld(k_f_g + c);
    R.rs_a = gld(k_rstd_a + tok0 + tl); R.rs_f = gld(k_rstd_f + tok0 + tl);
    R.bt = gld(k_beta + c); R.low_l = gld(k_low + el + c); R.x_l = ald<BF>(pX, el + c);
    R.q_l = ald<BF>(k_q, el + c);
    const int heads = TB(heads), Lp = TB(Lp);
    R.p_pre = c < L ? ald<BF>(TB(probs), (((long)b * heads + 0) * L + tl) * Lp + c) : 0.f;
}
// waves 1..3 of the fused head: wave 1 the x tile, wave 2 k, wave 3 v (16 loads a lane; rows past L re-read row L-1 and are
// zeroed at the store), and the probability row of head `wave` where there is one
template <bool BF, unsigned KOFF>
__device__ __forceinline__ void top_bwd_prefetch_tile(TopBwdRegs& R) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int L = TB(L), Lp = TB(Lp), heads = TB(heads), b = blockIdx.x, tl = L - 1;
    const long tok0 = (long)b * L;
    const float* const src = wave == 1 ? TB(X) : wave == 2 ? TB(k) : TB(v);
    R.p_pre = (wave < heads && lane < L) ? ald<BF>(TB(probs), (((long)b * heads + wave) * L + tl) * Lp + lane) : 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const int idx = lane + p * 64, r = idx >> 4, c4 = (idx & 15) << 2;
        R.tq[p] = ald4<BF>(src, (tok0 + min(r, L - 1)) * 64 + c4);
    }
}

// per-head vectors of the attention backward (thread = (head, feature i)): the weight-gradient operands of key / value
// (dWk = AK^T RK, dWv = AV^T RV), q_h Wk_h and dC_h Wv_h.  sVec offsets as in top_bwd_body.
template <int DH, bool BF, unsigned KOFF>
__device__ __forceinline__ void top_bwd_head_vectors(int head, int i, int b, int heads, const float (&wkc)[DH], const float (&wvc)[DH],
                                                     const float* sX, float* sVec) {
    const float* sDC = sVec + 832; const float* sQ = sVec + 896; const float* sDs = sVec + 1024; const float* sPd = sVec + 1280;
    float* sQK = sVec + 1536; float* sCV = sVec + 1792;
    const int o = head * 64 + i;
    float* const k_rk = TB(rk); float* const k_rv = TB(rv); float* const k_ak = TB(ak); float* const k_av = TB(av);
    float rk = 0.f, rv = 0.f;
    float rk2 = 0.f, rv2 = 0.f;
#pragma unroll 16
    for (int j = 0; j < 64; j += 2) {               // fixed trip count, two chains each
        const float x0 = sX[j * FS + i], x1 = sX[(j + 1) * FS + i];
        rk += sDs[head * 64 + j] * x0; rv += sPd[head * 64 + j] * x0;
        rk2 += sDs[head * 64 + j + 1] * x1; rv2 += sPd[head * 64 + j + 1] * x1;
    }
    rk += rk2; rv += rv2;
    const long e = ((long)b * heads + head) * 64 + i;
    const bool mine = i / DH == head;
    ast<BF>(k_rk, e, rk); ast<BF>(k_rv, e, rv);
    ast<BF>(k_ak, e, mine ? sQ[i] : 0.f); ast<BF>(k_av, e, mine ? sDC[i] : 0.f);
    sQK[o] = gemv_cols_dot<DH>(wkc, sQ, head * DH);
    sCV[o] = gemv_cols_dot<DH>(wvc, sDC, head * DH);
}

// Waves 4..7 of fused_layer_bwd_kernel<.., TopBwdP> while waves 0..3 run top_bwd_body<.., HELPED = true>: they take the two
// wide steps whose weight columns can be requested long before they are needed -- dU (64 dense_2 columns per thread), the
// dH partials (64 dense_1 columns) and the per-head vectors -- so that neither the issue of those loads nor their registers sit on waves 0..3' dependent
// chain.  help_a holds barriers 1-3 of TOP_BWD_BARRIERS, help_b barriers 4-11 (tests/test_host_cpu.py counts them); the
// caller stages the lower block's gelu' tile between the two.
template <bool BF, unsigned KOFF>
__device__ __forceinline__ void top_bwd_help_prefetch(TopBwdHelpRegs& H) {
    const int t = (int)threadIdx.x - 256, lane = t & 63, hw = t >> 6;
    const int L = TB(L), b = blockIdx.x;
    const float* const pU = TB(u); const float* const pW2 = TB(w2); const float* const pW1 = TB(w1);
    // wave hw: inner units [64 hw, 64 hw + 64) of dU (all 64 k of dense_2), and the same units as the K slice of dU . W1
    H.u4 = ald4<BF>(pU, ((long)b * L + L - 1) * 256 + 64 * hw + 4 * (lane & 15));
    gemv_cols4_load(pW2, 256, 0, 64 * hw, H.w2c);
    gemv_cols4_load(pW1, 64, 64 * hw, 0, H.w1c);
}
template <bool BF, unsigned KOFF>
__device__ __forceinline__ void top_bwd_help_a(const TopBwdHelpRegs& H, float* sVec) {
    const float* sDT = sVec; float* sDU = sVec + 64; float* sRed = sVec + 320;
    const int t = (int)threadIdx.x - 256, lane = t & 63, hw = t >> 6;
    const int b = blockIdx.x;
    float* const k_dU = TB(dU);
    const int g = lane & 15, kq = lane >> 4, u0 = 64 * hw + 4 * g;
    const f32x4 u4 = H.u4;
    const f32x4 (&w2c)[16] = H.w2c;
    const f32x4 (&w1c)[16] = H.w1c;
    lds_barrier();
    // dU = (dT2 . W2) * gelu'(u)
    const f32x4 s4 = gemv_cols4_dot(w2c, sDT, 0);
    const f32x4 du = {s4.x * gelu_grad_f(u4.x), s4.y * gelu_grad_f(u4.y), s4.z * gelu_grad_f(u4.z), s4.w * gelu_grad_f(u4.w)};
    if (kq == 0) { st4(sDU + u0, du); ast4<BF>(k_dU, (long)b * 256 + u0, du); }
    lds_barrier();
    // d(hmix) partials = dU . W1   (4 slices of 64 inner units; wave 0 of the body sums them)
    const f32x4 h4 = gemv_cols4_dot(w1c, sDU, 64 * hw);
    if (kq == 0) st4(sRed + hw * 64 + 4 * g, h4);
    lds_barrier();
}
// gu / uw: the LOWER block's gelu' tile [L][256] -> registers, a quarter per barrier interval of the head's narrow steps.  A
// wave that requests 16 KB at once sits in the issue of those loads until the memory pipeline has room (the whole head
// moves ~240 KB through one CU's 64 B/clk path) and arrives late at the next barrier; nothing of the head needs this tile,
// so it trickles in behind the loads the head does wait for.  The caller stores uw to LDS after the head.
template <int DH, bool BF, unsigned KOFF>
__device__ __forceinline__ void top_bwd_help_b(const float* sX, float* sVec, const float* gu, f32x4 (&uw)[16]) {
    const int t = (int)threadIdx.x - 256, lane = t & 63, hw = t >> 6;
    const int heads = TB(heads), L = TB(L), b = blockIdx.x;
    const long tok0 = (long)b * L;
    const bool hv = hw < heads;
    float wkc[DH], wvc[DH];
    const float* const pWk = TB(wk); const float* const pWv = TB(wv);
    auto u_quarter = [&](int q) {
#pragma unroll
        for (int i = 4 * q; i < 4 * q + 4; ++i) {
            const int idx = t + i * 256, r = idx >> 6, c4 = (idx & 63) << 2;
            uw[i] = ald4<BF>(gu, (tok0 + min(r, L - 1)) * 256 + c4);
        }
    };
    gemv_cols_load<DH>(pWk, 64, (hv ? hw : 0) * DH, lane, wkc);
    gemv_cols_load<DH>(pWv, 64, (hv ? hw : 0) * DH, lane, wvc);
    lds_barrier();
    u_quarter(0);                 // (steps 4, 5, 8, 9 of the body: the ones in which waves 0..3 touch no global memory)
    lds_barrier();
    u_quarter(1);
    lds_barrier();
    lds_barrier();
    if (hv) top_bwd_head_vectors<DH, BF, KOFF>(hw, lane, b, heads, wkc, wvc, sX, sVec);
    lds_barrier();
    u_quarter(2);
    lds_barrier();
    u_quarter(3);
    lds_barrier();
    lds_barrier();
}

template <int DH, bool BF, unsigned KOFF, bool HELPED>
__device__ __forceinline__ void top_bwd_body(const TopBwdRegs& R, const DropSeed& dseed, float* sX, float* sK, float* sV, const float* sTab,
                                             float* sVec, float* sDX) {
    float* sDT = sVec;            // 64   dT2 (grad of the dense_2 output)
    float* sDU = sVec + 64;       // 256  dU
    float* sRed = sVec + 320;     // 256  partial sums [4][64]
    float* sDH = sVec + 576;      // 64   grad of hmix
    float* sDO = sVec + 640;      // 64
    float* sDF = sVec + 704;      // 64   (1 - beta^2) dF  (beta^2 dF is folded into sLast)
    float* sLast = sVec + 768;    // 64   extra gradient of the last row: dzA + dzF + beta^2 dF (+ dq Wq later)
    float* sDC = sVec + 832;      // 64
    float* sQ = sVec + 896;       // 64   q_last
    float* sDQ = sVec + 960;      // 64
    float* sDs = sVec + 1024;     // [4][64] ds per head
    float* sPd = sVec + 1280;     // [4][64] Drop(p) per head
    float* sQK = sVec + 1536;     // [4][64] q_h Wk_h
    float* sCV = sVec + 1792;     // [4][64] dC_h Wv_h
    float* sPl = sVec + 2048;     // 64   low-pass projector column P[j][L-1]
    float* sSum = sVec + 2112;    // [2][4] sum_j ds, sum_j Drop(p) per head

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    BSTAMP(0);
    const int L = TB(L), Lp = TB(Lp), heads = TB(heads), cb = TB(cb);
    const int b = blockIdx.x, tl = L - 1;
    const long tok0 = (long)b * L, el = (tok0 + tl) * 64;

    // The step opens with ONE dependent chain on wave 0 (upstream gradient -> LayerNorm backward -> dT2) that everything
    // else waits for, so wave 0 issues that chain's loads before anything else and NO other load of its own until the chain
    // is through (loads return in issue order, and the vmcnt field cannot name more than 63 younger loads); its three
    // dropout multipliers of row L-1 (Philox, independent of the data) are evaluated under the latency of those loads.
    // The other waves' bulk requests (dense_1 columns, the x / k / v tiles) are not waited for in this stage.
    const int c = lane;
    float dy = 0.f, dz_ff = 0.f;
    float xa = 0.f, xf = 0.f, g_a = 0.f, g_f = 0.f, rs_a = 0.f, rs_f = 0.f, bt = 0.f, low_l = 0.f, x_l = 0.f;
    float u_mine = 0.f;
    float w2c[HELPED ? 1 : 64], w1c[HELPED ? 1 : 64];
    f32x4 tx[4], tk[4], tv[4];
    // kernarg fields are read in batches AHEAD of their use: kernarg_field is one scalar load, and a field read at its
    // point of use costs that load's round trip (100-200 cycles) in front of every global access -- measured 5.4 k cycles for
    // the 45 loads that open this step when each address was fetched on its own
    const float* const pX = TB(X); const float* const pK = TB(k); const float* const pV = TB(v);
    const float* const pW1 = TB(w1); const float* const pW2 = TB(w2); const float* const pU = TB(u);
    const float* const pWo = TB(wo); const float* const pWq = TB(wq); const float* const pWk = TB(wk); const float* const pWv = TB(wv);
    auto weight_loads = [&]() {                  // (HELPED: waves 4..7 hold these columns)
        if constexpr (!HELPED) {
            u_mine = ald<BF>(pU, (tok0 + tl) * 256 + tid);
            gemv_cols_load<64>(pW2, 256, 0, tid, w2c);
            gemv_cols_load<64>(pW1, 64, 64 * wave, lane, w1c);
        }
    };
    auto tile_loads = [&]() {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int idx = tid + p * 256, r = idx >> 4, c4 = (idx & 15) << 2;
            const long et = (tok0 + min(r, L - 1)) * 64 + c4;            // branch-free: rows past L re-read row L-1, zeroed
            tx[p] = ald4<BF>(pX, et); tk[p] = ald4<BF>(pK, et); tv[p] = ald4<BF>(pV, et);
            if (r >= L) { tx[p] = f32x4{0, 0, 0, 0}; tk[p] = tx[p]; tv[p] = tx[p]; }
        }
    };
    if (wave != 0) {
        if constexpr (!HELPED) { weight_loads(); tile_loads(); }
        // the other two dropout multipliers of row L-1 (Philox, independent of the data): waves 1 and 2 have nothing else to do
        // in this stage, wave 0 picks them up in the LayerNorm step (sDH / sDQ are free until then)
        if (wave == 1) { const DropP d_o = TB(drop_o); sDH[c] = drop_mult1(d_o, dseed, (uint64_t)(el + c)); }
        if (wave == 2) { const DropP d_f = TB(drop_f); sDQ[c] = drop_mult1(d_f, dseed, (uint64_t)(el + c)); }
    }
    if (wave == 0) {
        // upstream gradient of the last row = sum of the logits backward's split-K slabs (requested by top_bwd_prefetch)
        const int ns = TB(dh_nsplit);
        const DropP d_ff = TB(drop_ff);
        float* const k_pg_ff = TB(pg_ff); float* const k_pb_ff = TB(pb_ff); float* const k_dT = TB(dT);
        xa = R.xa; xf = R.xf; g_a = R.g_a; g_f = R.g_f; rs_a = R.rs_a; rs_f = R.rs_f; bt = R.bt; low_l = R.low_l; x_l = R.x_l;
        const float m_ff = drop_mult1(d_ff, dseed, (uint64_t)(el + c));        // under what is left of the loads' latency
        asm volatile("" :: "v"(m_ff));
        __builtin_amdgcn_sched_barrier(0);
        float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sp = 0; sp < 32; ++sp) part[sp & 3] += R.sl[sp];
        if (ns > 32) {
            const float* const slabs = TB(dh_slabs);
            const long stride = TB(dh_stride);
            for (int sp = 32; sp < ns; ++sp) part[0] += gld(slabs + (long)sp * stride + (long)b * 64 + c);
        }
        dy = (part[0] + part[1]) + (part[2] + part[3]);
        // FeedForward LayerNorm backward (row L-1)
        dz_ff = ln_row_bwd(dy, R.g_ff, R.xh_ff, R.rs_ff);
        const float dt = dz_ff * m_ff;
        sDT[c] = dt;
        sQ[c] = R.q_l;
        gst(k_pg_ff + (long)b * 64 + c, dy * R.xh_ff);
        gst(k_pb_ff + (long)b * 64 + c, dy);
        ast<BF>(k_dT, (long)b * 64 + c, dt);
        if constexpr (!HELPED) { __builtin_amdgcn_sched_barrier(0); weight_loads(); tile_loads(); }
    }
    // (stage 3's fields: read here, a stage ahead)
    const float k_oma = TB(oma), k_alpha = TB(alpha);
    float* const k_pg_a = TB(pg_a); float* const k_pb_a = TB(pb_a); float* const k_pg_f = TB(pg_f); float* const k_pb_f = TB(pb_f);
    float* const k_dO = TB(dO); float* const k_pbeta = TB(pbeta); float* const k_dU = TB(dU);
    lds_barrier();
    BSTAMP(1);
    // HELPED: the x / k / v tiles requested at the top of the kernel (top_bwd_prefetch_tile) have landed: -> LDS, while waves
    // 4..7 run the next two steps
    if constexpr (HELPED) {
        if (wave != 0) {
            float* const dst = wave == 1 ? sX : wave == 2 ? sK : sV;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const int idx = lane + p * 64, r = idx >> 4, c4 = (idx & 15) << 2;
                st4(dst + r * FS + c4, r < L ? R.tq[p] : f32x4{0, 0, 0, 0});
            }
        }
    }

    // ---- dU = (dT2 . W2) * gelu'(u)   (one of the 256 inner units per thread; HELPED: waves 4..7 do it, top_bwd_help_a)
    if constexpr (!HELPED) {
        const float du = gemv_cols_dot<64>(w2c, sDT, 0) * gelu_grad_f(u_mine);
        sDU[tid] = du;
        ast<BF>(k_dU, (long)b * 256 + tid, du);
    }
    float woc[16], wqc[16], wkc[HELPED ? 1 : DH], wvc[HELPED ? 1 : DH];
    gemv_cols_load<16>(pWo, 64, 16 * wave, lane, woc);
    gemv_cols_load<16>(pWq, 64, 16 * wave, lane, wqc);
    const bool hv = tid < heads * 64;             // thread (head = wave, i = lane) of the per-head vector products
    if constexpr (!HELPED) {
        gemv_cols_load<DH>(pWk, 64, (hv ? wave : 0) * DH, lane, wkc);
        gemv_cols_load<DH>(pWv, 64, (hv ? wave : 0) * DH, lane, wvc);
    }
    // (the attention step's and the per-head step's fields)
    const float* const k_probs = TB(probs); const DropP d_p = TB(drop_p);
    float* const k_dq = TB(dq); float* const k_pbk = TB(pbk); float* const k_pbv = TB(pbv);
    lds_barrier();
    BSTAMP(2);
    // ---- d(hmix) = dU . W1 + dz   (4 slices of 64 inner units; HELPED: waves 4..7)
    if constexpr (!HELPED) {
        sRed[wave * 64 + lane] = gemv_cols_dot<64>(w1c, sDU, 64 * wave);
#pragma unroll
        for (int p = 0; p < 4; ++p) {                 // the tiles have landed by now
            const int idx = tid + p * 256, r = idx >> 4, c4 = (idx & 15) << 2;
            st4(sX + r * FS + c4, tx[p]); st4(sK + r * FS + c4, tk[p]); st4(sV + r * FS + c4, tv[p]);
        }
    }
    lds_barrier();
    BSTAMP(3);
    if (wave == 0) {
        const float dh = (sRed[c] + sRed[64 + c]) + (sRed[128 + c] + sRed[192 + c]) + dz_ff;
        // alpha mix + the two LayerNorm backwards (attention branch scaled by 1 - alpha, filter branch by alpha)
        const float dya = k_oma * dh, dyf = k_alpha * dh;
        const float dza = ln_row_bwd(dya, g_a, xa, rs_a);
        const float dzf = ln_row_bwd(dyf, g_f, xf, rs_f);
        gst(k_pg_a + (long)b * 64 + c, dya * xa); gst(k_pb_a + (long)b * 64 + c, dya);
        gst(k_pg_f + (long)b * 64 + c, dyf * xf); gst(k_pb_f + (long)b * 64 + c, dyf);
        const float dO = dza * sDH[c];              // the multipliers waves 1 / 2 left there
        const float dF = dzf * sDQ[c];
        sDO[c] = dO;
        ast<BF>(k_dO, (long)b * 64 + c, dO);
        const float b2 = bt * bt;
        sDF[c] = (1.0f - b2) * dF;
        sLast[c] = dza + dzf + b2 * dF;
        // d sqrt_beta: f = low + beta^2 (x - low)
        gst(k_pbeta + (long)b * 64 + c, 2.0f * bt * dF * (x_l - low_l));
        // column L-1 of the low-pass projector: P[j][L-1] = (1/L) sum_k w_k cos(2 pi k (j - (L-1)) / L)
        float pl = 0.f;
        if (c < L)
            for (int k = 0; k < cb; ++k) {
                const float w = (k == 0 || 2 * k == L) ? 1.0f : 2.0f;
                pl += w * (sTab[2 * (k * 64 + c)] * sTab[2 * (k * 64 + tl)] + sTab[2 * (k * 64 + c) + 1] * sTab[2 * (k * 64 + tl) + 1]);
            }
        sPl[c] = pl / (float)L;
#ifdef BSAREC_FINE_STAMPS
        BSTAMP(12);
#endif
    }
    lds_barrier();
    BSTAMP(4);
    // ---- dC = dO . Wo   (4 slices of 16 output features)
    sRed[wave * 64 + lane] = gemv_cols_dot<16>(woc, sDO, 16 * wave);
    lds_barrier();
    BSTAMP(5);
    if (tid < 64) sDC[tid] = (sRed[tid] + sRed[64 + tid]) + (sRed[128 + tid] + sRed[192 + tid]);
    lds_barrier();
    BSTAMP(6);

    // ---- attention backward of the last query: one wave per head, lane = key
    if (wave < heads) {
        const int head = wave, key = lane;
        float dpd = 0.f, p = 0.f, mp = 0.f;
        if (key < L) {
            const float* vr = sV + key * FS + head * DH;
            const float* dc = sDC + head * DH;
#pragma unroll
            for (int c = 0; c < DH; c += 4) {
                const f32x4 vv = ld4(vr + c), dv = ld4(dc + c);
                dpd += vv.x * dv.x + vv.y * dv.y + vv.z * dv.z + vv.w * dv.w;
            }
            const long pe = (((long)b * heads + head) * L + tl) * Lp + key;
            p = HELPED ? R.p_pre : ald<BF>(k_probs, pe);
            mp = drop_mult1(d_p, dseed, (uint64_t)pe);
        }
        const float dp = dpd * mp;
        const float delta = group_sum<64>(p * dp);
        const float ds = p * (dp - delta) / sqrtf((float)DH);
        const float pd = p * mp;
        sDs[head * 64 + key] = ds;
        sPd[head * 64 + key] = pd;
        const float s1 = group_sum<64>(ds), s2 = group_sum<64>(pd);
        if (key == 0) { sSum[head] = s1; sSum[4 + head] = s2; }
    }
    lds_barrier();
    BSTAMP(7);
    // ---- per-head vectors: dq, the weight-gradient operands of key / value, q_h Wk_h, dC_h Wv_h
    if (tid < 64) {
        const int c = tid, head = c / DH;
        float acc = 0.f;
        float a4[4] = {0.f, 0.f, 0.f, 0.f};          // 64 padded keys (ds = 0 past L, tiles zero-filled): fixed trip count
#pragma unroll 16
        for (int j = 0; j < 64; ++j) a4[j & 3] += sDs[head * 64 + j] * sK[j * FS + c];
        acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        sDQ[c] = acc;
        ast<BF>(k_dq, (long)b * 64 + c, acc);
        gst(k_pbk + (long)b * 64 + c, sQ[c] * sSum[head]);
        gst(k_pbv + (long)b * 64 + c, sDC[c] * sSum[4 + head]);
    }
    if constexpr (!HELPED) {
        if (hv) top_bwd_head_vectors<DH, BF, KOFF>(wave, lane, b, heads, wkc, wvc, sX, sVec);
    }
    lds_barrier();
    BSTAMP(8);
    // dq . Wq joins the last row's extra gradient (4 slices of 16 features)
    sRed[wave * 64 + lane] = gemv_cols_dot<16>(wqc, sDQ, 16 * wave);
    lds_barrier();
    BSTAMP(9);
    if (tid < 64) sLast[tid] += (sRed[tid] + sRed[64 + tid]) + (sRed[128 + tid] + sRed[192 + tid]);
    lds_barrier();
    BSTAMP(10);
    // ---- dX, all rows
    {
        const int lr = tid >> 4, lc = (tid & 15) << 2;
        const f32x4 df = ld4(sDF + lc);
        float* const pdX = sDX ? nullptr : TB(dX);
        for (int j = lr; j < L; j += 16) {
            f32x4 dx = df * sPl[j];
            for (int h = 0; h < heads; ++h)
                dx += ld4(sQK + h * 64 + lc) * sDs[h * 64 + j] + ld4(sCV + h * 64 + lc) * sPd[h * 64 + j];
            if (j == tl) dx += ld4(sLast + lc);
            if (sDX) {
                if constexpr (BF) {       // what the block below would read back from the bf16 gradient tensor
                    const unsigned a = pk_bf16(dx.x, dx.y), c2 = pk_bf16(dx.z, dx.w);
                    dx = f32x4{bf_lo(a), bf_hi(a), bf_lo(c2), bf_hi(c2)};
                }
                st4(sDX + j * FS + lc, dx);
            } else ast4<BF>(pdX, (tok0 + j) * 64 + lc, dx);
        }
    }
    lds_barrier();                // (counted in TOP_BWD_BARRIERS: the dX tile is complete)
    BSTAMP(15);
}

template <int DH, bool BF>
__global__ void __launch_bounds__(256)
top_bwd_kernel(const TopBwdP P_unused) {
    constexpr unsigned KOFF = 0;
    constexpr int TS = 64 * FS;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sX = sm;
    float* sK = sm + TS;
    float* sV = sm + 2 * TS;
    float* sTab = sm + 3 * TS;                  // FUSED_MAX_CB * 128
    float* sVec = sTab + FUSED_MAX_CB * 128;
    BSTAMP(0);
    TopBwdRegs R;
    if (threadIdx.x < 64) top_bwd_prefetch<BF, KOFF>(R);
    const DropSeed dseed = drop_seed(TB(drop_f));
    build_twiddle_table(TB(tw), TB(L), TB(cb), sTab);
    lds_barrier();
    top_bwd_body<DH, BF, KOFF, false>(R, dseed, sX, sK, sV, sTab, sVec, nullptr);
}


static inline size_t top_bwd_smem_bytes() { return (size_t)(3 * 64 * FS + FUSED_MAX_CB * 128 + 2176) * 4; }
