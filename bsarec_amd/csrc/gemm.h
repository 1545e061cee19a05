// Generic exact-fp32 MFMA GEMM core for gfx950 (v_mfma_f32_32x32x2_f32), used by every
// contraction on the BSARec path: token-parallel projections (NT), input-gradient products (NN),
// weight-gradient products (TN, split-K), the batched attention products and the full-catalogue
// logits.  C[m,n] = sum_k A(m,k) * B(n,k).
//
// Layout rules
//   * an operand whose k index is contiguous in memory ("KC") is staged in LDS row-major
//     [row][BK+4] and read as one ds_read_b128 per 8-deep k-block: lane (i, half) takes
//     k = 8*blk + 4*half .. +3, i.e. the k order inside a block is permuted identically for A and
//     B (a sum over k does not care); the +4 pad makes the 16-lane b128 groups conflict-free.
//   * an operand whose row index is contiguous ("KM": A^T or B^T products) is staged k-major
//     [BK][rows] by a straight 16-byte copy and read with four conflict-free ds_read_b32.
//   * 256 threads = 4 waves arranged WM x WN (512 = 8 waves as 2 x 4 for the 256-wide fp32 tiles: their 92 KB of staging
//     allow one workgroup per CU, and 4 waves of 344 registers left every SIMD with ONE wave); each wave owns
//     (BM/WM) x (BN/WN) of the tile as 32x32 MFMA accumulators; global->register->LDS double buffering, one barrier per k-tile.
//   * the epilogue always goes through an LDS image of the C tile so that every epilogue
//     (bias, LayerNorm, softmax, gradient fix-ups) reads whole rows and stores 16 B per lane.
//
// bf16 products (BF = true; cfg.storage = 1 outside the fused shape class): the operands stay fp32 tensors in HBM, are
// rounded to bf16 (nearest even) on their way INTO LDS and multiplied with v_mfma_f32_32x32x16_bf16 (fp32 accumulate):
// one matrix instruction per 16-deep k-block where the fp32 form needs eight, and -- unlike v_mfma_f32_*_f32 on gfx950 --
// off the vector ALU.  Epilogues, statistics and every tensor in memory are unchanged.  LDS images, in dwords of two bf16
// (k = 2j in the low half of dword j):
//   * "KC" operand: [row][16 + 4]; a loaded f32x4 becomes one ds_write_b64; lane (i, half) reads k = 16 blk + 8 half .. + 7 as
//     one ds_read_b128 (rows 80 B apart: the 16-lane groups of a b128 read fall on distinct banks).
//   * "KM" operand: a thread loads rows 4g .. 4g+3 of TWO consecutive k, packs the four (k, k + 1) pairs and writes the tile
//     TRANSPOSED as [row][16 + 2] (four ds_write_b32, 2-way conflicts; 8 lanes = 128 contiguous bytes per global k row);
//     lane (i, half) reads its 8 k as two ds_read_b64 (rows 72 B apart: conflict-free over the 64 banks).
#pragma once
#include "common.h"

enum { XF_NONE = 0, XF_GELU = 1, XF_DROP = 2 };

#define GEMM_BK 32
#define GEMM_THREADS 256

struct GemmP {
    const float* A[3];
    const float* B[3];
    int M, N, K;          // N is the padded (multiple of 4) logical width
    int Nb;               // valid rows of B (<= N); rows beyond are read as zero
    int Kv;               // valid k extent (<= K) of a k-major operand; K itself may be padded to 4
    long lda, ldb;        // stride in floats of the non-contiguous dimension of A / B
    int nseg;             // K segments accumulated into one output (A[s], B[s]), else 1
    int nprob;            // independent problems selected by blockIdx.z (A[p], B[p]), else 1
    int nsplit;           // split-K factor, else 1
    int kchunk;           // K range per split (multiple of GEMM_BK)
    int nh;               // inner batch count (heads) for batched products, else 1
    long a_sb, a_sh, b_sb, b_sh;   // batch strides (floats): outer (sequence) and inner (head)
};

struct XformP {           // operand transform applied while loading
    DropP drop;           // XF_DROP: attention-probability dropout regenerated on load
    int L, Lp;            // XF_DROP: element index = ((zb*L + q)*Lp + key)
    int act;              // XF_GELU: which hidden_act (common.h act_f; 0 = gelu)
};

struct TileCtx {          // what an epilogue needs to know about its tile
    int m0, n0, M, N;
    int zb;               // batch index (b*nh + head) or 0
    int b, hh;            // zb split into (sequence, head)
    int prob, split;
};

// ---------------------------------------------------------------------------------------------
// global -> registers -> LDS tile loader
// ---------------------------------------------------------------------------------------------
#define GEMM_LDK_KC 20      // bf16 images: dwords per row of a k-contiguous operand (16 k pairs + 4)
#define GEMM_LDK_KM 18      //              dwords per row of a transposed k-major operand (16 k pairs + 2)

template <int R, bool KM, int XF, bool BF = false, int NT = GEMM_THREADS>
struct TileLoader {
    static constexpr int NV = R * GEMM_BK / 4 / NT;
    static_assert(NV >= 1, "tile too small for the workgroup");
    static_assert(!(BF && KM) || NV % 2 == 0, "bf16 k-major staging pairs two k rows per thread");
    f32x4 v[NV];

    __device__ __forceinline__ f32x4 xform(f32x4 x, int q, int key0, bool valid, const XformP& X, int zb) const {
        if (XF == XF_GELU) {
            x.x = act_f(x.x, X.act); x.y = act_f(x.y, X.act); x.z = act_f(x.z, X.act); x.w = act_f(x.w, X.act);
        } else if (XF == XF_DROP) {
            if (valid) {
                const uint64_t e = ((uint64_t)zb * X.L + q) * X.Lp + key0;
                x = x * drop_mult4(X.drop, e >> 2);
            }
        }
        return x;
    }

    int r0_, k0_, rlim_, klim_;     // tile origin / limits of the stage in flight (the transform needs them at store time)

    // position of load p of this thread inside the tile: (k, first of 4 rows) of a k-major operand.  bf16 products: loads
    // 2q and 2q + 1 take the SAME four rows at k = 2 kp and 2 kp + 1, with j = tid + 256 q, rows 4 ((j & 7) + 8 (j >> 7)),
    // kp = (j >> 3) & 15 -- 8 lanes cover 128 contiguous bytes of a k row, a half-wave's transposed writes are 2-way.
    static __device__ __forceinline__ void km_pos(int p, int& kk, int& rr) {
        const int tid = threadIdx.x;
        if (BF) {
            const int j = tid + (p >> 1) * NT;
            kk = 2 * ((j >> 3) & 15) + (p & 1);
            rr = ((j & 7) + 8 * (j >> 7)) << 2;
        } else {
            constexpr int RV = R / 4;
            const int idx = tid + p * NT;
            kk = idx / RV; rr = (idx % RV) << 2;
        }
    }

    // rows [r0, r0+R) limited by rlim; k range [k0, k0+BK) limited by klim.  Only ISSUES the loads: the element
    // transform (GELU / dropout) is applied in store(), when the data is consumed -- applying it here would make
    // the wave wait for the round trip at issue time and serialise the prefetch.
    __device__ __forceinline__ void load(const float* __restrict__ base, long ld, int r0, int rlim, int k0, int klim) {
        const int tid = threadIdx.x;
        r0_ = r0; k0_ = k0; rlim_ = rlim; klim_ = klim;
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int idx = tid + p * NT;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (!KM) {
                const int gr = r0 + (idx >> 3), gk = k0 + ((idx & 7) << 2);
                if (gr < rlim && gk < klim) x = ld4(base + (long)gr * ld + gk);
            } else {
                int kk, rr;
                km_pos(p, kk, rr);
                const int gk = k0 + kk, gr = r0 + rr;
                if (gk < klim) {
                    const float* src = base + (long)gk * ld + gr;
                    if (gr + 3 < rlim) x = ld4(src);
                    else {
                        if (gr < rlim) x.x = src[0];
                        if (gr + 1 < rlim) x.y = src[1];
                        if (gr + 2 < rlim) x.z = src[2];
                    }
                }
            }
            v[p] = x;
        }
    }

    __device__ __forceinline__ void store(float* s, const XformP& X, int zb) const {
        const int tid = threadIdx.x;
        if constexpr (BF) {
            unsigned* su = reinterpret_cast<unsigned*>(s);
            if (!KM) {
#pragma unroll
                for (int p = 0; p < NV; ++p) {
                    const int idx = tid + p * NT;
                    f32x4 x = v[p];
                    if (XF != XF_NONE) {
                        const int gr = r0_ + (idx >> 3), gk = k0_ + ((idx & 7) << 2);
                        x = xform(x, gr, gk, gr < rlim_ && gk < klim_, X, zb);
                    }
                    *reinterpret_cast<u32x2*>(su + (idx >> 3) * GEMM_LDK_KC + ((idx & 7) << 1)) = u32x2{pk_bf16(x.x, x.y), pk_bf16(x.z, x.w)};
                }
            } else {
#pragma unroll
                for (int q = 0; q < NV / 2; ++q) {
                    int kk, rr;
                    km_pos(2 * q, kk, rr);
                    f32x4 x0 = v[2 * q], x1 = v[2 * q + 1];
                    if (XF != XF_NONE) {
                        const int gk = k0_ + kk, gr = r0_ + rr;
                        x0 = xform(x0, gk, gr, gk < klim_ && gr < rlim_, X, zb);
                        x1 = xform(x1, gk + 1, gr, gk + 1 < klim_ && gr < rlim_, X, zb);
                    }
                    unsigned* d = su + rr * GEMM_LDK_KM + (kk >> 1);
                    d[0] = pk_bf16(x0.x, x1.x);
                    d[GEMM_LDK_KM] = pk_bf16(x0.y, x1.y);
                    d[2 * GEMM_LDK_KM] = pk_bf16(x0.z, x1.z);
                    d[3 * GEMM_LDK_KM] = pk_bf16(x0.w, x1.w);
                }
            }
            return;
        }
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int idx = tid + p * NT;
            f32x4 x = v[p];
            if (!KM) {
                if (XF != XF_NONE) {
                    const int gr = r0_ + (idx >> 3), gk = k0_ + ((idx & 7) << 2);
                    x = xform(x, gr, gk, gr < rlim_ && gk < klim_, X, zb);
                }
                st4(s + (idx >> 3) * (GEMM_BK + 4) + ((idx & 7) << 2), x);
            } else {
                constexpr int RV = R / 4;
                if (XF != XF_NONE) {
                    const int gk = k0_ + idx / RV, gr = r0_ + ((idx % RV) << 2);
                    x = xform(x, gk, gr, gk < klim_ && gr < rlim_, X, zb);
                }
                st4(s + (idx / RV) * R + ((idx % RV) << 2), x);
            }
        }
    }
};

template <int BM, int BN, bool A_KM, bool B_KM, bool BF = false>
struct GemmSmem {
    static constexpr int A_TILE = BF ? BM * (A_KM ? GEMM_LDK_KM : GEMM_LDK_KC) : (A_KM ? GEMM_BK * BM : BM * (GEMM_BK + 4));
    static constexpr int B_TILE = BF ? BN * (B_KM ? GEMM_LDK_KM : GEMM_LDK_KC) : (B_KM ? GEMM_BK * BN : BN * (GEMM_BK + 4));
    static constexpr int STAGE = 2 * (A_TILE + B_TILE);
    static constexpr int LDC = BN + 4;
    static constexpr int CT = BM * LDC;
    static constexpr int FLOATS = STAGE > CT ? STAGE : CT;
    static constexpr size_t BYTES = (size_t)FLOATS * 4;
};

// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool A_KM, bool B_KM, int AXF, int BXF, bool BGRAD, bool BF = false, class Epi>
__device__ __forceinline__ void
gemm_body(const GemmP& P, const XformP& X, const Epi& epi, float* __restrict__ bgrad /* [nprob][nsplit][M] */,
          const int bx, const int by, const int bz, float* __restrict__ smem) {
    constexpr int NT = 64 * WM * WN;                 // 4 waves (256 threads) or, for the 256-wide fp32 tiles, 8 (2 x 4)
    static_assert(NT == 256 || NT == 512, "4 or 8 waves");
    constexpr int BK = GEMM_BK;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold a 32x32 MFMA");
    using SM = GemmSmem<BM, BN, A_KM, B_KM, BF>;
    constexpr int LDAS = BF ? (A_KM ? GEMM_LDK_KM : GEMM_LDK_KC) : (A_KM ? BM : BK + 4);
    constexpr int LDBS = BF ? (B_KM ? GEMM_LDK_KM : GEMM_LDK_KC) : (B_KM ? BN : BK + 4);
    float* As0 = smem;
    float* As1 = smem + SM::A_TILE;
    float* Bs0 = smem + 2 * SM::A_TILE;
    float* Bs1 = Bs0 + SM::B_TILE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wrow = (wave / WN) * WTM, wcol = (wave % WN) * WTN;

    // blockIdx.z -> (batch, problem, split)
    int z = bz;
    const int split = z % P.nsplit; z /= P.nsplit;
    const int prob = z % P.nprob;
    const int zb = z / P.nprob;
    TileCtx ctx;
    ctx.m0 = bx * BM; ctx.n0 = by * BN; ctx.M = P.M; ctx.N = P.N;
    ctx.zb = zb; ctx.b = zb / P.nh; ctx.hh = zb % P.nh; ctx.prob = prob; ctx.split = split;

    const long aoff = (long)ctx.b * P.a_sb + (long)ctx.hh * P.a_sh;
    const long boff = (long)ctx.b * P.b_sb + (long)ctx.hh * P.b_sh;
    int kbeg = 0, kend = P.K;
    if (P.nsplit > 1) { kbeg = split * P.kchunk; kend = min(P.K, kbeg + P.kchunk); }
    const int kvend = min(kend, P.Kv);
    const int ktiles = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
    const int nit = ktiles * P.nseg;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum = 0.f;

    // Two register stages per operand: tile it+2 is requested while tile it is multiplied and tile it+1 waits in
    // registers for its turn to be written to LDS, so a global/L2 round trip has two whole k-steps to land (one
    // k-step of a 64x64 tile is only 16 MFMAs per wave -- shorter than the round trip).
    TileLoader<BM, A_KM, AXF, BF, NT> la0, la1;
    TileLoader<BN, B_KM, BXF, BF, NT> lb0, lb1;
    auto issue = [&](int it, TileLoader<BM, A_KM, AXF, BF, NT>& la, TileLoader<BN, B_KM, BXF, BF, NT>& lb) {
        const int seg = (P.nseg > 1) ? it / ktiles : prob;
        const int kt = (P.nseg > 1) ? it % ktiles : it;
        const int k0 = kbeg + kt * BK;
        la.load(P.A[seg] + aoff, P.lda, ctx.m0, P.M, k0, A_KM ? kvend : kend);
        lb.load(P.B[seg] + boff, P.ldb, ctx.n0, P.Nb, k0, B_KM ? kvend : kend);
    };
    auto compute = [&](const float* __restrict__ as, const float* __restrict__ bs) {
        if constexpr (BF) {
            // operand fragment of lane (row, half) for the 16-deep k-block kb: dwords 8 kb + 4 half .. + 3 of its row
            auto frag = [&](const float* __restrict__ t, int r, int ld, bool km, int kb) -> u32x4 {
                const unsigned* q = reinterpret_cast<const unsigned*>(t) + r * ld + kb * 8 + 4 * half;
                if (!km) return *reinterpret_cast<const u32x4*>(q);
                const u32x2 lo = *reinterpret_cast<const u32x2*>(q), hi = *reinterpret_cast<const u32x2*>(q + 2);
                return u32x4{lo.x, lo.y, hi.x, hi.y};
            };
#pragma unroll
            for (int kb = 0; kb < BK / 16; ++kb) {
                u32x4 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = frag(as, wrow + i * 32 + l31, LDAS, A_KM, kb);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = frag(bs, wcol + j * 32 + l31, LDBS, B_KM, kb);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma_bf16(af[i], bf[j], acc[i][j]);
            }
#ifndef EXP_NO_BGRAD
            if (BGRAD && A_KM && by == 0 && tid < BM) {          // column sums of the (bf16-rounded) A tile
                const unsigned* q = reinterpret_cast<const unsigned*>(as) + tid * LDAS;
#pragma unroll
                for (int k = 0; k < BK / 2; ++k) bsum += bf_lo(q[k]) + bf_hi(q[k]);
            }
#endif
            return;
        }
#pragma unroll
        for (int kb = 0; kb < BK / 8; ++kb) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wrow + i * 32 + l31;
                if (!A_KM) af[i] = ld4(as + r * LDAS + kb * 8 + 4 * half);
                else {
                    const float* p = as + (kb * 8 + 4 * half) * LDAS + r;
                    af[i].x = p[0]; af[i].y = p[LDAS]; af[i].z = p[2 * LDAS]; af[i].w = p[3 * LDAS];
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wcol + j * 32 + l31;
                if (!B_KM) bf[j] = ld4(bs + r * LDBS + kb * 8 + 4 * half);
                else {
                    const float* p = bs + (kb * 8 + 4 * half) * LDBS + r;
                    bf[j].x = p[0]; bf[j].y = p[LDBS]; bf[j].z = p[2 * LDBS]; bf[j].w = p[3 * LDBS];
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
#ifndef EXP_NO_BGRAD
        if (BGRAD && A_KM && by == 0 && tid < BM) {
#pragma unroll 8
            for (int k = 0; k < BK; ++k) bsum += as[k * LDAS + tid];
        }
#endif
    };

    if (nit > 0) issue(0, la0, lb0);
    if (nit > 1) issue(1, la1, lb1);
    if (nit > 0) { la0.store(As0, X, zb); lb0.store(Bs0, X, zb); }
    lds_barrier();       // LDS-only: the prefetched global loads stay in flight
    for (int it = 0; it < nit; it += 2) {
        // even step: LDS buffer 0 = tile it, stage 1 = tile it+1, stage 0 is free
        if (it + 2 < nit) issue(it + 2, la0, lb0);
        compute(As0, Bs0);
        if (it + 1 < nit) { la1.store(As1, X, zb); lb1.store(Bs1, X, zb); }
        lds_barrier();       // LDS-only: the prefetched global loads stay in flight
        if (it + 1 >= nit) break;
        // odd step: LDS buffer 1 = tile it+1, stage 0 = tile it+2, stage 1 is free
        if (it + 3 < nit) issue(it + 3, la1, lb1);
        compute(As1, Bs1);
        if (it + 2 < nit) { la0.store(As0, X, zb); lb0.store(Bs0, X, zb); }
        lds_barrier();       // LDS-only: the prefetched global loads stay in flight
    }
    if (BGRAD && A_KM && by == 0 && tid < BM && ctx.m0 + tid < P.M)
        bgrad[((long)prob * P.nsplit + split) * P.M + ctx.m0 + tid] = bsum;

    // accumulators -> LDS image of the C tile (aliases the staging buffers; the loop's last
    // barrier has already retired every read of them)
    float* Cs = smem;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wrow + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                Cs[row * SM::LDC + wcol + j * 32 + l31] = acc[i][j][r];
            }
    __syncthreads();
    epi.template run<BM, BN, NT>(Cs, ctx);
}

template <int BM, int BN, int WM, int WN, bool A_KM, bool B_KM, int AXF, int BXF, bool BGRAD, bool BF, class Epi>
__global__ void __launch_bounds__(64 * WM * WN, (BF && WM * WN == 4) ? 2 : 1)
// (bf16 products: <= 256 registers, i.e. two workgroups per CU for the 256-wide tiles too -- their matrix time no longer covers
// the loads of a lone workgroup: C3 step 12.2 -> 10.3 ms.  The fp32 form loses with the same cap: 17.7 -> 18.4 ms, spills.)
gemm_kernel(const GemmP P, const XformP X, const Epi epi, float* __restrict__ bgrad) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // XCD-aware tile order.  The dispatcher deals consecutive workgroups to the 8 XCDs in turn and every XCD has its own L2:
    // in launch order the N/BN workgroups that read one A row block would sit under 8 different L2s, and with m fastest
    // they are a whole grid row apart in time as well -- at the C3 shape the 52 ... 210 MB activation operand was re-read
    // from the Infinity Cache / HBM once per column tile (4 ... 16 times).  Here every XCD label (id % 8) takes ONE contiguous
    // run of the (m, n) tile sequence with n fastest (bijective for any tile count): the column tiles of a row block run
    // side by side under one L2, which also keeps the whole weight matrix.  A pure speed choice: any order is correct.
    int bx = blockIdx.x, by = blockIdx.y;
    if (gridDim.y > 1) {
        const int gy = gridDim.y, nt = gridDim.x * gy, id = bx + gridDim.x * by;
        const int q = nt >> 3, r = nt & 7, xcd = id & 7;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
        by = t % gy; bx = t / gy;
    }
    gemm_body<BM, BN, WM, WN, A_KM, B_KM, AXF, BXF, BGRAD, BF, Epi>(P, X, epi, bgrad, bx, by, blockIdx.z, smem);
}
