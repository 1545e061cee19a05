// Fused BSARecBlock kernels for the headline shape class (hidden = 64, L <= 64): ONE workgroup per
// sequence runs a whole block (FrequencyLayer, QKV, attention, dense + LN + alpha-mix, FFN + LN) with
// the sequence's activations resident in LDS and the weights streamed from L2 straight into MFMA
// operand registers (no weight staging, no inter-kernel HBM round trips).  src/model/bsarec.py:56-104.
//
// MFMA orientation (v_mfma_f32_32x32x2_f32, D = A.B):
//   * "rows x weights": A = activation rows from LDS (lane = token, one ds_read_b128 per 8-deep k-block
//     with the k order permuted as in gemm.h), B = weight rows from global (lane = output feature, 16 B
//     per lane per k-block).  Accumulator: lane = feature column, register = token row.
//   * attention is computed transposed (S^T = K.Q^T: lane = query, registers = keys) so the softmax is a
//     reduction over registers plus one cross-half shuffle, and P^T is consumed in place as the B operand
//     of ctx^T = V^T.P^T (an accumulator tile is a valid B operand of a product that sums over its rows).
// 8 waves = 2 groups of 4 (each 2 x 2 over a 64-token x 64-feature tile), two waves per SIMD, 1 workgroup per CU
// (~145 KB LDS forward, ~159 KB backward).
#pragma once
#include "common.h"
#include "kernels.h"
#include <type_traits>

#define FS 68          // LDS row stride (floats) of a 64-wide tile: 16-B aligned rows, conflict-free b128 reads
#define FU 260         // LDS row stride of the 256-wide FFN tile
#define FUSED_MAX_CB 8

struct FusedFwdP {
    const float* X; float* Xout;
    const float *sqrt_beta, *f_g, *f_b, *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo, *a_g, *a_b, *w1, *b1, *w2, *b2, *ff_g, *ff_b;
    const float* tw; const int* ids32;
    float *xhat_f, *rstd_f, *q, *k, *v, *probs, *ctx, *xhat_a, *rstd_a, *hmix, *u, *xhat_ff, *rstd_ff, *dsp;
    float* gp;              // [B, L, 4d] gelu'(pre-activation); `u` receives gelu(pre-activation) -- what the backward's two
                            // consumers need (dU = (dT2 . W2) * gelu'; dW2 = dT2^T . gelu), neither has to evaluate erf again
    int L, Lp, cb, heads;
    float alpha, oma, eps;
    DropP drop_f, drop_p, drop_o, drop_ff;
    float* trash;           // >= 1 KiB scratch: target of the stores of padded rows (keeps the store stream branch-free)
    long long* stamps;      // diagnostic: per-phase s_memtime of workgroup 0 (null in production)
    // bottom block only (e_E != null): the embedding front-end rides in phase 0 -- batch assembly from the device table
    // (or ids), E[ids] + Pos, LayerNorm, dropout (src/model/_abstract_model.py:14-24) -> the x tile in LDS, and X[0],
    // xhat0, rstd0, ids32 to global for the backward
    const float *e_E, *e_pos, *e_g, *e_b; const int64_t* e_ids; GatherP e_gp; DropP e_drop; int e_V;
    float *e_X0, *e_xhat, *e_rstd; int* e_ids32;
    int xout_f32;           // bf16 storage: Xout is the LAST layer's output, which stays an fp32 tensor (logits / API)
    const float* filter_cw; // FM instantiation only: FMLPRec's complex_weight [L/2 + 1][64][2] (re, im)
};


// Kernel parameters are read from the kernarg segment at their point of use.  With ~40 pointers in the block the
// compiler otherwise hoists every kernarg load to the entry and spills >100 SGPRs (1400 v_readlane in the ISA);
// the opaque asm pins each load after the preceding barrier so live ranges stay inside one phase.
template <class T> struct GlobalPtr { static __device__ __forceinline__ T fix(T v) { return v; } };
template <class U> struct GlobalPtr<U*> {           // pointers read from memory are generic: re-tag them as global
    static __device__ __forceinline__ U* fix(U* v) {
        return (U*)(__attribute__((address_space(1))) U*)v;
    }
};
template <class T>
__device__ __forceinline__ T kernarg_field(unsigned byte_off) {
    typedef __attribute__((address_space(4))) const char* kptr_t;
    typedef __attribute__((address_space(4))) const unsigned* kwords_t;
    kptr_t base = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(base));
    kwords_t w = (kwords_t)(base + byte_off);
    union { T v; unsigned u[(sizeof(T) + 3) / 4]; } x;
#pragma unroll
    for (unsigned i = 0; i < (sizeof(T) + 3) / 4; ++i) x.u[i] = w[i];
    return GlobalPtr<T>::fix(x.v);
}
#define KARG(S, f) kernarg_field<decltype(S::f)>((unsigned)offsetof(S, f))
// Touch every 64-byte line of the kernarg segment: the first scalar load of a line after a launch misses the scalar cache
// all the way to memory, and a kernel that reads its ~40 pointers phase by phase pays that miss again and again on its
// dependent chain.  Wave w requests lines w, w + 8, w + 16; the values are handed to kernarg_touch_done() at a point where
// the wave waits for scalar loads anyway (an s_load whose destination the compiler does not track must never be left
// in flight: the late write would land in a register that has been given to something else).
struct KernargTouch { unsigned v[3]; };
template <unsigned BYTES>
__device__ __forceinline__ KernargTouch kernarg_touch() {
    static_assert(BYTES <= 24 * 64, "three lines per wave");
    typedef __attribute__((address_space(4))) const unsigned* kwords_t;
    kwords_t base = (kwords_t)__builtin_amdgcn_kernarg_segment_ptr();
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr unsigned LAST = (BYTES - 4) / 64;                     // last line of the segment
    KernargTouch t;
#pragma unroll
    for (unsigned j = 0; j < 3; ++j) t.v[j] = base[16 * min(8 * j + wave, LAST)];
    return t;
}
__device__ __forceinline__ void kernarg_touch_done(const KernargTouch& t) { asm volatile("" :: "s"(t.v[0]), "s"(t.v[1]), "s"(t.v[2])); }
#define STAMP(i) do { long long* st_ = KARG(PTYPE, stamps); if (st_ && blockIdx.x == 0 && threadIdx.x == 0) st_[i] = clock64(); } while (0)


// Explicit global-address-space accessors.  Pointers read from the kernarg block at run time are generic to
// the compiler, and generic accesses become flat_load / flat_store, which tick BOTH vmcnt and lgkmcnt: every LDS
// wait would then also wait for the global stores in flight.  Casting at the access site keeps them global_*.
#define AS_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ f32x4 gld4(const float* p) { return *reinterpret_cast<const AS_GLOBAL f32x4*>((const AS_GLOBAL float*)p); }
__device__ __forceinline__ void gst4(float* p, f32x4 v) { *reinterpret_cast<AS_GLOBAL f32x4*>((AS_GLOBAL float*)p) = v; }
__device__ __forceinline__ float gld(const float* p) { return *(const AS_GLOBAL float*)p; }
__device__ __forceinline__ void gst(float* p, float v) { *(AS_GLOBAL float*)p = v; }
__device__ __forceinline__ int gldi(const int* p) { return *(const AS_GLOBAL int*)p; }

__device__ __forceinline__ int rho(int r) { return (r & 3) + 8 * (r >> 2); }

// ---------------------------------------------------------------------------------------------
// bf16 storage (cfg.storage = 1, config C2).  BF = true instantiations keep every tensor that crosses a kernel boundary
// inside the block stack -- the activations saved for the backward, the inter-block gradients, the operands of the
// weight-gradient products -- as bf16 (behind the same float* the fp32 build uses: element e lives at byte 2e), read
// the Linear weights from a bf16 shadow of the fp32 masters, and feed v_mfma_f32_32x32x16_bf16 (fp32 accumulate).
// LDS tiles, LayerNorm, softmax, GELU, dropout and all statistics stay fp32.
// ---------------------------------------------------------------------------------------------
// (u32x2 / u32x4, pk_bf16, bf_lo / bf_hi, pk8, mfma_bf16: common.h)
// activation tensor accessors: 4 consecutive elements at element index e (16 B fp32 / 8 B bf16), or one element
template <bool BF> __device__ __forceinline__ f32x4 ald4(const float* base, long e) {
    if constexpr (BF) {
        const u32x2 r = *reinterpret_cast<const AS_GLOBAL u32x2*>((const AS_GLOBAL char*)base + 2 * e);
        return f32x4{bf_lo(r.x), bf_hi(r.x), bf_lo(r.y), bf_hi(r.y)};
    } else return gld4(base + e);
}
template <bool BF> __device__ __forceinline__ void ast4(float* base, long e, const f32x4& v) {
    if constexpr (BF) *reinterpret_cast<AS_GLOBAL u32x2*>((AS_GLOBAL char*)base + 2 * e) = u32x2{pk_bf16(v.x, v.y), pk_bf16(v.z, v.w)};
    else gst4(base + e, v);
}
template <bool BF> __device__ __forceinline__ float ald(const float* base, long e) {
    if constexpr (BF) return bf_lo(*reinterpret_cast<const AS_GLOBAL unsigned short*>((const AS_GLOBAL char*)base + 2 * e));
    else return gld(base + e);
}
template <bool BF> __device__ __forceinline__ void ast(float* base, long e, float v) {
    if constexpr (BF) *reinterpret_cast<AS_GLOBAL unsigned short*>((AS_GLOBAL char*)base + 2 * e) = (unsigned short)(pk_bf16(v, 0.f) & 0xFFFFu);
    else gst(base + e, v);
}



// ---- pruned DFT with a per-workgroup twiddle table tab[k][t] = (cos, sin)(2 pi k t / L), k < cb, t < 64
// (first / nthr: the threads that build it -- by default the whole workgroup)
__device__ __forceinline__ void build_twiddle_table(const float* __restrict__ tw, int L, int cb, float* __restrict__ tab,
                                                    int first = 0, int nthr = 0) {
    if (nthr == 0) nthr = blockDim.x;
    if ((int)threadIdx.x < first) return;
    for (int i = threadIdx.x - first; i < cb * 64; i += nthr) {
        const int k = i >> 6, t = i & 63;
        float c = 0.f, s = 0.f;
        if (t < L) { const int a = (int)((unsigned)(k * t) % (unsigned)L); c = gld(tw + 2 * a); s = gld(tw + 2 * a + 1); }
        tab[2 * i] = c; tab[2 * i + 1] = s;
    }
}

// spectrum of NSRC sources held as 64 x 64 LDS tiles: spec[s][k][re|im][64]; bins are processed 4 at a time,
// part = scratch [16 row groups][NSRC][4][2][64]
template <int NSRC, class Src>
__device__ __forceinline__ void dft_spectrum_tab(const Src& src, int L, int cb, const float* __restrict__ tab,
                                                 float* __restrict__ spec, float* __restrict__ part) {
    const int lr = threadIdx.x >> 4, lc = (threadIdx.x & 15) << 2;
    for (int k0 = 0; k0 < cb; k0 += 4) {
        f32x4 re[NSRC][4], im[NSRC][4];
#pragma unroll
        for (int s = 0; s < NSRC; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) { re[s][j] = f32x4{0, 0, 0, 0}; im[s][j] = f32x4{0, 0, 0, 0}; }
#pragma unroll
        for (int r0 = 0; r0 < 64; r0 += 16) {
            const int t = r0 + lr;
            if (t < L) {
                f32x4 x[NSRC];
#pragma unroll
                for (int s = 0; s < NSRC; ++s) x[s] = src(s, t, lc);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k0 + j < cb) {
                        const float c = tab[2 * ((k0 + j) * 64 + t)], sn = tab[2 * ((k0 + j) * 64 + t) + 1];
#pragma unroll
                        for (int s = 0; s < NSRC; ++s) { re[s][j] += x[s] * c; im[s][j] -= x[s] * sn; }
                    }
            }
        }
#pragma unroll
        for (int s = 0; s < NSRC; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                st4(part + (((lr * NSRC + s) * 4 + j) * 2 + 0) * 64 + lc, re[s][j]);
                st4(part + (((lr * NSRC + s) * 4 + j) * 2 + 1) * 64 + lc, im[s][j]);
            }
        lds_barrier();
        for (int i = threadIdx.x; i < NSRC * 4 * 128; i += 256) {
            const int s = i / 512, j = (i >> 7) & 3, rc = i & 127;
            if (k0 + j < cb) {
                float acc = 0.f;
#pragma unroll
                for (int g = 0; g < 16; ++g) acc += part[g * NSRC * 512 + i];
                spec[(s * cb + k0 + j) * 128 + rc] = acc;
            }
        }
        lds_barrier();
    }
}

__device__ __forceinline__ f32x4 lowpass_tab(const float* __restrict__ spec, int t, int lc, int L, int cb,
                                             const float* __restrict__ tab) {
    f32x4 low = {0, 0, 0, 0};
    for (int k = 0; k < cb; ++k) {
        const float w = (k == 0 || (2 * k == L)) ? 1.0f : 2.0f;
        const float c = tab[2 * (k * 64 + t)] * w, sn = tab[2 * (k * 64 + t) + 1] * w;
        low += ld4(spec + (k * 2 + 0) * 64 + lc) * c - ld4(spec + (k * 2 + 1) * 64 + lc) * sn;
    }
    return low * (1.0f / (float)L);
}

// Weight fragments of a K-deep chunk (K = 64 or 32) in MFMA B-operand registers: issue the loads of the whole chunk,
// use them later (latency hidden by the caller).  Product mode MM:
//   0  fp32 MFMA (v_mfma_f32_32x32x2_f32): per 8-deep k-block kb, lane half h holds k = 8 kb + 4 h + {0..3}  (one 16-byte load)
//   1  bf16 MFMA (v_mfma_f32_32x32x16_bf16) on a bf16 shadow of the weights: per 16-deep k-block s, lane half h holds
//      k = 16 s + 8 h + {0..7}   (one 16-byte load of the shadow)
//   2  "x3": fp32 operands, the SAME k layout as mode 1 (two 16-byte loads per k-block), every fp32 product evaluated on
//      the bf16 matrix cores as six bf16 x bf16 partial products -- see mfma_x3 below.
// The per-lane element offsets carry KH = 4 h (mode 0) resp. 8 h (modes 1, 2), for the LDS rows and the weight rows alike.
template <int MM, int K> struct WFrag;
template <int K> struct WFrag<0, K> { f32x4 w[K / 8]; };
template <int K> struct WFrag<1, K> { u32x4 w[K / 16]; };
template <int K> struct WFrag<2, K> { f32x4 w[K / 8]; };            // w[2 s], w[2 s + 1] = the 8 consecutive k of k-block s

// fp32 products on the bf16 matrix cores.  On gfx950 v_mfma_f32_*_f32 executes on the SIMD's vector ALU (no vector
// instruction of either resident wave issues meanwhile) at 1/16 of the bf16 MFMA rate; the bf16 matrix pipe is
// separate silicon.  An fp32 value is EXACTLY the sum of three bf16 pieces (a = a1 + a2 + a3, each the round-to-nearest
// bf16 of the remainder: 3 x 8 significant bits), so a . b = sum over the nine piece products; the six with i + j <= 4
// carry everything down to 2^-27 |a||b| -- below the fp32 rounding of the accumulation itself (measured on 64 x 256 x 64
// products against fp64: max error 3.5e-7 of the largest entry, plain fp32 matmul 6.1e-7) -- and each bf16 x bf16
// product is exact in the MFMA's fp32 accumulator.  Cost per 16-deep k-block of a 32 x 32 tile: 2 x 44 vector instructions
// (the two splits) + 6 MFMAs on the matrix pipe, against 8 fp32 MFMAs = 520 cycles of the vector ALU.
struct Split3 { u32x4 h, m, l; };
__device__ __forceinline__ f32x4 bf_lo4(const u32x4& p, int half) {     // pieces 0..3 (half = 0) or 4..7 (half = 1) widened back
    return half == 0 ? f32x4{bf_lo(p.x), bf_hi(p.x), bf_lo(p.y), bf_hi(p.y)} : f32x4{bf_lo(p.z), bf_hi(p.z), bf_lo(p.w), bf_hi(p.w)};
}
__device__ __forceinline__ Split3 split3(f32x4 a0, f32x4 a1) {
    Split3 r;
    r.h = pk8(a0, a1);
    a0 = a0 - bf_lo4(r.h, 0); a1 = a1 - bf_lo4(r.h, 1);                 // exact: the remainder of a round-to-nearest cut
    r.m = pk8(a0, a1);
    a0 = a0 - bf_lo4(r.m, 0); a1 = a1 - bf_lo4(r.m, 1);
    r.l = pk8(a0, a1);
    return r;
}
__device__ __forceinline__ f32x16 mfma_x3(const Split3& a, const Split3& b, f32x16 acc) {
    acc = mfma_bf16(a.l, b.h, acc);                                     // smallest terms first
    acc = mfma_bf16(a.h, b.l, acc);
    acc = mfma_bf16(a.m, b.m, acc);
    acc = mfma_bf16(a.m, b.h, acc);
    acc = mfma_bf16(a.h, b.m, acc);
    return mfma_bf16(a.h, b.h, acc);
}

// "rows x weights": B[k][n] = W[n][k], lane n reads its own weight row (contiguous in k); e = element offset of
// W[n][k0 + KH] in the fp32 master (MM = 0, 2) resp. the bf16 shadow (MM = 1) behind `base`
template <int MM, int K>
__device__ __forceinline__ void load_w(const float* __restrict__ base, long e, WFrag<MM, K>& f) {
    if constexpr (MM == 1) {
#pragma unroll
        for (int s = 0; s < K / 16; ++s) f.w[s] = *reinterpret_cast<const AS_GLOBAL u32x4*>((const AS_GLOBAL char*)base + 2 * (e + 16 * s));
    } else if constexpr (MM == 2) {
#pragma unroll
        for (int s = 0; s < K / 16; ++s) { f.w[2 * s] = gld4(base + e + 16 * s); f.w[2 * s + 1] = gld4(base + e + 16 * s + 4); }
    } else {
#pragma unroll
        for (int kb = 0; kb < K / 8; ++kb) f.w[kb] = gld4(base + e + 8 * kb);
    }
}
// "x . W": B[k][j] = W[k][j], lane j reads a weight column (stride LDW elements): coalesced dword (fp32) or
// halfword (bf16) loads; e = element offset of W[k0 + KH][j]
template <int MM, int K, int LDW>
__device__ __forceinline__ void load_wT(const float* __restrict__ base, long e, WFrag<MM, K>& f) {
    if constexpr (MM == 1) {
        const AS_GLOBAL unsigned short* g = reinterpret_cast<const AS_GLOBAL unsigned short*>((const AS_GLOBAL char*)base + 2 * e);
#pragma unroll
        for (int s = 0; s < K / 16; ++s) {
            unsigned h[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = g[(16 * s + j) * LDW];
            f.w[s] = u32x4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
        }
    } else if constexpr (MM == 2) {
        const float* gw = base + e;
#pragma unroll
        for (int s = 0; s < K / 16; ++s) {
            f.w[2 * s] = f32x4{gld(gw + (16 * s + 0) * LDW), gld(gw + (16 * s + 1) * LDW), gld(gw + (16 * s + 2) * LDW), gld(gw + (16 * s + 3) * LDW)};
            f.w[2 * s + 1] = f32x4{gld(gw + (16 * s + 4) * LDW), gld(gw + (16 * s + 5) * LDW), gld(gw + (16 * s + 6) * LDW), gld(gw + (16 * s + 7) * LDW)};
        }
    } else {
        const float* gw = base + e;
#pragma unroll
        for (int kb = 0; kb < K / 8; ++kb) {
            f.w[kb].x = gld(gw + (8 * kb + 0) * LDW); f.w[kb].y = gld(gw + (8 * kb + 1) * LDW);
            f.w[kb].z = gld(gw + (8 * kb + 2) * LDW); f.w[kb].w = gld(gw + (8 * kb + 3) * LDW);
        }
    }
}
// acc += rows(sa) . frag: sa = LDS row of this lane + k0 + KH (fp32 tile; converted to bf16 on the way for MM = 1, split for MM = 2)
template <int MM, int K>
__device__ __forceinline__ void mma_w(const float* __restrict__ sa, const WFrag<MM, K>& f, f32x16& acc) {
    if constexpr (MM == 1) {
#pragma unroll
        for (int s = 0; s < K / 16; ++s) acc = mfma_bf16(pk8(ld4(sa + 16 * s), ld4(sa + 16 * s + 4)), f.w[s], acc);
    } else if constexpr (MM == 2) {
#pragma unroll
        for (int s = 0; s < K / 16; ++s)
            acc = mfma_x3(split3(ld4(sa + 16 * s), ld4(sa + 16 * s + 4)), split3(f.w[2 * s], f.w[2 * s + 1]), acc);
    } else {
#pragma unroll
        for (int kb = 0; kb < K / 8; ++kb) {
            const f32x4 a = ld4(sa + 8 * kb);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], f.w[kb][s], acc, 0, 0, 0);
        }
    }
}
// both operands from LDS rows (k contiguous in each; pointers carry KH): acc += rows(sa) . rows(sb)^T over K
template <int MM, int K>
__device__ __forceinline__ void mma_ll(const float* __restrict__ sa, const float* __restrict__ sb, f32x16& acc) {
    if constexpr (MM == 1) {
#pragma unroll
        for (int s = 0; s < K / 16; ++s)
            acc = mfma_bf16(pk8(ld4(sa + 16 * s), ld4(sa + 16 * s + 4)), pk8(ld4(sb + 16 * s), ld4(sb + 16 * s + 4)), acc);
    } else if constexpr (MM == 2) {
#pragma unroll
        for (int s = 0; s < K / 16; ++s)
            acc = mfma_x3(split3(ld4(sa + 16 * s), ld4(sa + 16 * s + 4)), split3(ld4(sb + 16 * s), ld4(sb + 16 * s + 4)), acc);
    } else {
#pragma unroll
        for (int kb = 0; kb < K / 8; ++kb) {
            const f32x4 a = ld4(sa + 8 * kb), b = ld4(sb + 8 * kb);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
        }
    }
}

// LayerNorm row pass over a 64 x 64 LDS tile held as two split-K partial tiles (16 lanes x float4 per row, 512 threads = 32
// rows per pass, two passes):  v = (tileA + tileB + bias) * dropout + residual ; xhat, rstd -> global ;
// y = gamma*xhat + beta ; MIX: y = alpha*dsp + (1-alpha)*y.  Result -> LDS (outL, may be null) and global (outG).
// In two pieces: ln_pre requests the three per-column vectors and evaluates the thread's two dropout masks (Philox) -- neither
// depends on the tile, so the caller places it BEFORE the matrix product that fills the tile and ahead of any weight
// prefetch (loads return in issue order: a 16-byte vector requested behind 64 KB of weights waits for all of them).
struct LnPre { f32x4 bi, g, be, dm[2]; };
__device__ __forceinline__ LnPre ln_pre(const float* __restrict__ bias, const float* __restrict__ gamma, const float* __restrict__ beta,
                                        const DropP& drop, const DropSeed& dseed, long tok0) {
    const int lr = threadIdx.x >> 4, lc = (threadIdx.x & 15) << 2;
    LnPre P;
    P.bi = gld4(bias + lc); P.g = gld4(gamma + lc); P.be = gld4(beta + lc);
#pragma unroll
    for (int i = 0; i < 2; ++i) P.dm[i] = drop_mult4(drop, dseed, (uint64_t)((tok0 + lr + 32 * i) * 64 + lc) >> 2);
    return P;
}
template <bool MIX, bool BF>
__device__ __forceinline__ void ln_rows_64(const LnPre& P, const float* __restrict__ tileA, const float* __restrict__ tileB,
                                           const float* __restrict__ resid, float eps, const float* __restrict__ dsp,
                                           float alpha, float oma, long tok0, int L, float* __restrict__ outL,
                                           float* __restrict__ outG, float* __restrict__ xhatG, float* __restrict__ rstdG,
                                           bool out_f32 = false /* BF: outG is an fp32 tensor (the last layer's output) */,
                                           bool round_outL = false /* BF: the LDS copy holds what a reader of outG would see */) {
    const int lr = threadIdx.x >> 4, lc = (threadIdx.x & 15) << 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = lr + 32 * i;
        const bool ok = r < L;
        const long e = (tok0 + r) * 64 + lc;
        f32x4 v = {0, 0, 0, 0};
        if (ok) v = (ld4(tileA + r * FS + lc) + ld4(tileB + r * FS + lc) + P.bi) * P.dm[i] + ld4(resid + r * FS + lc);
        const float mean = group_sum<16>(v.x + v.y + v.z + v.w) * (1.0f / 64.0f);
        f32x4 dl = {0, 0, 0, 0};
        if (ok) dl = v - mean;
        const float var = group_sum<16>(dl.x * dl.x + dl.y * dl.y + dl.z * dl.z + dl.w * dl.w) * (1.0f / 64.0f);
        const float rs = 1.0f / sqrtf(var + eps);
        f32x4 y = {0, 0, 0, 0};
        if (ok) {
            const f32x4 xh = dl * rs;
            y = P.g * xh + P.be;
            if (MIX) y = alpha * ld4(dsp + r * FS + lc) + oma * y;
            ast4<BF>(xhatG, e, xh);
            if (BF && out_f32) gst4(outG + e, y); else ast4<BF>(outG, e, y);
            if (lc == 0) gst(rstdG + tok0 + r, rs);
        }
        if (BF && round_outL) {
            const unsigned a = pk_bf16(y.x, y.y), c = pk_bf16(y.z, y.w);
            y = f32x4{bf_lo(a), bf_hi(a), bf_lo(c), bf_hi(c)};
        }
        if (outL) st4(outL + r * FS + lc, y);
    }
}

// The one-row top block as the tail of this kernel (fused_top.h; declared here, defined there).
struct NoTail {};
template <bool BF> struct TopFwdRegs;
template <bool BF, unsigned KOFF, bool KV> __device__ __forceinline__ void top_fwd_prefetch(TopFwdRegs<BF>& R);
template <int DH, bool BF, unsigned KOFF, bool HELPED>
__device__ __forceinline__ void top_fwd_rest(const TopFwdRegs<BF>& R, const DropSeed& dseed, float* sX, float* sK, float* sV,
                                             float* sPart, float* sTab, float* sSpec, float* sVec, const int* sIds);
template <bool BF> struct TopFwdHelpRegs;
template <bool BF, unsigned KOFF> __device__ __forceinline__ void top_fwd_help_prefetch(TopFwdHelpRegs<BF>& H);
template <int DH, bool BF, unsigned KOFF>
__device__ __forceinline__ void top_fwd_help(const TopFwdHelpRegs<BF>& H, const float* sX, float* sK, float* sV, float* sVec);
template <class T> struct IsTail { static constexpr bool value = true; };
template <> struct IsTail<NoTail> { static constexpr bool value = false; };
// ... and its backward as the head of this block's backward kernel.  Waves 4..7 of that kernel execute exactly
// TOP_BWD_BARRIERS barriers (top_bwd_help_a + top_bwd_help_b: the two wide steps they take over) while waves 0..3 run
// top_bwd_body<.., HELPED = true> (which contains that many, all unconditional).
constexpr int TOP_BWD_BARRIERS = 11;
struct TopBwdRegs {                                 // what top_bwd_prefetch requests at the top of the kernel (wave 0, lane = column)
    float sl[32];                                   // the first 32 split-K slabs of the upstream gradient (0 past dh_nsplit)
    float xh_ff, g_ff, rs_ff, xa, xf, g_a, g_f, rs_a, rs_f, bt, low_l, x_l, q_l;
    f32x4 tq[16];                                   // waves 1..3 (fused head only): the x / k / v tile, one per wave
    float p_pre;                                    // waves < heads (fused head only): the probability row of the last query
};
struct TopBwdHelpRegs { f32x4 w2c[16], w1c[16], u4; };      // waves 4..7: dense_2 / dense_1 columns, gelu' input of their units
template <bool BF, unsigned KOFF> __device__ __forceinline__ void top_bwd_prefetch(TopBwdRegs& R);
template <bool BF, unsigned KOFF> __device__ __forceinline__ void top_bwd_prefetch_tile(TopBwdRegs& R);
template <bool BF, unsigned KOFF> __device__ __forceinline__ void top_bwd_help_prefetch(TopBwdHelpRegs& H);
template <int DH, bool BF, unsigned KOFF, bool HELPED>
__device__ __forceinline__ void top_bwd_body(const TopBwdRegs& R, const DropSeed& dseed, float* sX, float* sK, float* sV, const float* sTab,
                                             float* sVec, float* sDX);
template <bool BF, unsigned KOFF> __device__ __forceinline__ void top_bwd_help_a(const TopBwdHelpRegs& H, float* sVec);
template <int DH, bool BF, unsigned KOFF> __device__ __forceinline__ void top_bwd_help_b(const float* sX, float* sVec, const float* gu, f32x4 (&uw)[16]);


// Forward: 8 waves = 2 groups of 4 (each group tiles 64 tokens x 64 features as 2 x 2 waves), two waves per SIMD
// so that one wave's MFMA chain overlaps the other's VALU / memory waits:
//   * FrequencyLayer (group 0, VALU + LDS) runs concurrently with the Q, K, V projections (group 1, MFMA);
//   * attention: wave pair (2c, 2c+1) owns combo c = (head, query tile) and splits its two key tiles;
//     row max / sum are exchanged through LDS, the two partial contexts are summed in a row pass;
//   * dense and dense_2 split K across the groups (two partial tiles, summed by the LayerNorm row pass),
//     dense_1 splits its four 64-wide output blocks.
// TAILP = TopFwdP: the block above is the one-row top block of the loss path and runs as this kernel's tail -- the
// output tile, the ids and the twiddle table stay in LDS, waves 4..7 exit, waves 0..3 carry on (one launch and the
// top block's whole load phase saved).
template <int DH, bool BF, class TAILP, bool X3 = false, bool FM = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
fused_layer_fwd_kernel(const FusedFwdP P_unused, const TAILP T_unused) {
#define PTYPE FusedFwdP
    static_assert(!(BF && X3), "x3 products work on fp32 tensors");
    constexpr int MM = X3 ? 2 : (BF ? 1 : 0);       // product mode of the MFMA helpers (WFrag)
    // FM: the sibling model FMLPRec's block (src/model/fmlprec.py:78-113) -- FilterLayer with the learnable complex filter
    // y = irfft(rfft(x) * W) over ALL L/2 + 1 bins, LayerNorm(Drop(y) + x), then the same feed-forward; no attention branch
    // (the reference's FMLPRecBlock has none).  Phases 1 (as the whole-spectrum DFT with the complex multiply), 5, 6, 7 run; the
    // 36-bin twiddle / spectrum tables live in the attention tiles this variant never uses.
    static_assert(!FM || (!BF && !X3 && !IsTail<TAILP>::value), "the FMLPRec block runs in fp32, without a top-block tail");
    constexpr int MAXCB = FM ? 36 : FUSED_MAX_CB;
    constexpr bool TAIL = IsTail<TAILP>::value;
    constexpr unsigned KOFF = (unsigned)((sizeof(FusedFwdP) + 7) & ~(size_t)7);     // kernarg offset of T_unused
    const auto R0_L = KARG(FusedFwdP, L);
    const auto R0_Lp = KARG(FusedFwdP, Lp);
    const auto R0_cb = KARG(FusedFwdP, cb);
    const auto R0_heads = KARG(FusedFwdP, heads);
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int TS = 64 * FS;
    float* sX = sm;
    float* sD = sm + TS;
    float* sH = sm + 2 * TS;                    // hmix; before that: DFT partials (with sE)
    float* sE = sm + 3 * TS;                    // second partial tile
    float* sR = sm + 4 * TS;                    // union: {sQ, sK, sVt, sC} | sU
    float* sQ = sR;
    float* sK = sR + TS;
    float* sVt = sR + 2 * TS;                   // V transposed: [feature][token]
    float* sC = sR + 3 * TS;
    float* sU = sR;                             // [64][FU]
    float* sTab = sR + 4 * TS;                  // FUSED_MAX_CB * 64 * 2
    float* sSpec = sTab + FUSED_MAX_CB * 128;   // FUSED_MAX_CB * 2 * 64
    float* sRed = sSpec + FUSED_MAX_CB * 128;   // softmax exchange: [4 pairs][2 key tiles][64 lanes]
    int* sIds = reinterpret_cast<int*>(sRed + 512);   // 64
    if constexpr (FM) { sTab = sR; sSpec = sR + 2 * TS; }     // MAXCB * 128 floats each (<= 2 tiles); sU covers them only after the filter

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int grp = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int L = R0_L, Lp = R0_Lp, heads = R0_heads, cb = R0_cb;
    const int b = blockIdx.x;
    const long tok0 = (long)b * L;
    const int col = wn * 32 + l31;              // this lane's output feature inside a 64-wide block
    constexpr int KHM = MM ? 8 : 4;             // k values per lane half and k-block (see WFrag)
    const int KH = KHM * half;
    const long wrow = (long)col * 64 + KH;
    const int arow = (wm * 32 + l31) * FS + KH;
    float* const trash = KARG(FusedFwdP, trash) + 4 * lane;

    STAMP(0);
    const KernargTouch ktouch = kernarg_touch<IsTail<TAILP>::value ? KOFF + (unsigned)sizeof(TAILP) : (unsigned)sizeof(FusedFwdP)>();
    const DropSeed dseed = drop_seed(KARG(FusedFwdP, drop_f));     // first loads of the kernel
    kernarg_touch_done(ktouch);
    const auto R1_X = KARG(FusedFwdP, X);
    const auto R1_ids32 = KARG(FusedFwdP, ids32);
    const auto R1_tw = KARG(FusedFwdP, tw);
    const auto R1_wq = KARG(FusedFwdP, wq);
    // every global LOAD of a phase is issued before the phase's first global STORE: vmcnt retires in issue
    // order, so a load queued behind stores would wait for their write acknowledgements
    WFrag<MM, 64> wA, wB;
    float qkv_bias[3] = {0.f, 0.f, 0.f};
    if (!FM && grp == 1) {
        load_w<MM, 64>(R1_wq, wrow, wA);
        qkv_bias[0] = gld(KARG(FusedFwdP, bq) + col); qkv_bias[1] = gld(KARG(FusedFwdP, bk) + col);
        qkv_bias[2] = gld(KARG(FusedFwdP, bv) + col);
    }
    // ---- phase 0: sequence tile, ids, twiddle table -> LDS
    const float* const eE = KARG(FusedFwdP, e_E);
    if (eE) {
        // embedding front-end of this sequence (bottom block): 16 lanes x float4 per token row, 32 rows per pass
        const GatherP gp = KARG(FusedFwdP, e_gp);
        const int64_t* const eids = KARG(FusedFwdP, e_ids);
        const float* const epos = KARG(FusedFwdP, e_pos);
        const int V = KARG(FusedFwdP, e_V);
        const float eeps = KARG(FusedFwdP, eps);
        const float* const e_g = KARG(FusedFwdP, e_g); const float* const e_b = KARG(FusedFwdP, e_b);
        int* const e_ids32 = KARG(FusedFwdP, e_ids32); const DropP e_drop = KARG(FusedFwdP, e_drop);
        float* const e_xhat = KARG(FusedFwdP, e_xhat); float* const e_X0 = KARG(FusedFwdP, e_X0); float* const e_rstd = KARG(FusedFwdP, e_rstd);
        long src = 0;
        if (gp.table) {
            src = *(const AS_GLOBAL long long*)gp.cursor + b;
            src = src < gp.n ? (long)*(const AS_GLOBAL int64_t*)(gp.perm + src) : 0;
            if (tid == 0) *(AS_GLOBAL int64_t*)(gp.ans_out + b) = *(const AS_GLOBAL int64_t*)(gp.ans_table + src);
        }
        const f32x4 eg = gld4(e_g + ((tid & 15) << 2)), eb = gld4(e_b + ((tid & 15) << 2));
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
            if (c4 == 0) { sIds[r] = id; if (ok) *(AS_GLOBAL int*)(e_ids32 + tok0 + r) = id; }
            const float mean = group_sum<16>(v.x + v.y + v.z + v.w) * (1.0f / 64.0f);
            f32x4 dl = {0, 0, 0, 0};
            if (ok) dl = v - mean;
            const float var = group_sum<16>(dl.x * dl.x + dl.y * dl.y + dl.z * dl.z + dl.w * dl.w) * (1.0f / 64.0f);
            const float rs = 1.0f / sqrtf(var + eeps);
            f32x4 y = {0, 0, 0, 0};
            if (ok) {
                const f32x4 xh = dl * rs;
                y = (eg * xh + eb) * drop_mult4(e_drop, dseed, (uint64_t)e >> 2);
                ast4<BF>(e_xhat, e, xh);
                ast4<BF>(e_X0, e, y);
                if (c4 == 0) gst(e_rstd + tok0 + r, rs);
            }
            st4(sX + r * FS + c4, y);
        }
    } else {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int idx = tid + p * 512, r = idx >> 4, c4 = (idx & 15) << 2;
        f32x4 v = ald4<BF>(R1_X, (tok0 + min(r, L - 1)) * 64 + c4);
        if (r >= L) v = f32x4{0, 0, 0, 0};
        st4(sX + r * FS + c4, v);
    }
    if (tid < 64) sIds[tid] = tid < L ? gldi(R1_ids32 + (tok0 + tid)) : 0;
    }
    build_twiddle_table(R1_tw, L, cb, sTab);
    lds_barrier();

    STAMP(1);
    const auto R2_bk = KARG(FusedFwdP, bk);
    const auto R2_bq = KARG(FusedFwdP, bq);
    const auto R2_bv = KARG(FusedFwdP, bv);
    const auto R2_drop_f = KARG(FusedFwdP, drop_f);
    const auto R2_dsp = KARG(FusedFwdP, dsp);
    const auto R2_eps = KARG(FusedFwdP, eps);
    const auto R2_f_b = KARG(FusedFwdP, f_b);
    const auto R2_f_g = KARG(FusedFwdP, f_g);
    const auto R2_k = KARG(FusedFwdP, k);
    const auto R2_q = KARG(FusedFwdP, q);
    const auto R2_rstd_f = KARG(FusedFwdP, rstd_f);
    const auto R2_sqrt_beta = KARG(FusedFwdP, sqrt_beta);
    const auto R2_v = KARG(FusedFwdP, v);
    const auto R2_wk = KARG(FusedFwdP, wk);
    const auto R2_wo = KARG(FusedFwdP, wo);
    const auto R2_wv = KARG(FusedFwdP, wv);
    const auto R2_xhat_f = KARG(FusedFwdP, xhat_f);
    // ---- phases 1 || 2: FrequencyLayer (group 0)  ||  Q, K, V projections (group 1)
    {
        float* part = sH;                            // [16][4][2][64] = 8192 floats over sH + sE
        const int lr = (tid & 255) >> 4, lc = (tid & 15) << 2;
        auto qkv = [&](auto whichc) {                // group 1: one 32 x 32 tile of Q, K or V per wave
            constexpr int which = decltype(whichc)::value;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (which == 0) { load_w<MM, 64>(R2_wk, wrow, wB); mma_w<MM, 64>(sX + arow, wA, acc); }
            else if (which == 1) { load_w<MM, 64>(R2_wv, wrow, wA); mma_w<MM, 64>(sX + arow, wB, acc); }
            else mma_w<MM, 64>(sX + arow, wA, acc);
            const float bias = qkv_bias[which];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + rho(r) + 4 * half;
                const float val = acc[r] + bias;
                if (which == 0) sQ[row * FS + col] = val;
                else if (which == 1) sK[row * FS + col] = val;
                else sVt[col * FS + row] = val;
            }
        };
#pragma unroll
        for (int ch = 0; ch < MAXCB / 4; ++ch) {
            const int k0 = 4 * ch;
            if (k0 >= cb) break;
            if (grp == 0) {                          // accumulate 4 bins over this thread's rows
                f32x4 re[4], im[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { re[j] = f32x4{0, 0, 0, 0}; im[j] = f32x4{0, 0, 0, 0}; }
#pragma unroll
                for (int r0 = 0; r0 < 64; r0 += 16) {
                    const int t = r0 + lr;
                    if (t < L) {
                        const f32x4 x = ld4(sX + t * FS + lc);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (k0 + j < cb) {
                                const float c = sTab[2 * ((k0 + j) * 64 + t)], sn = sTab[2 * ((k0 + j) * 64 + t) + 1];
                                re[j] += x * c; im[j] -= x * sn;
                            }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    st4(part + ((lr * 4 + j) * 2 + 0) * 64 + lc, re[j]);
                    st4(part + ((lr * 4 + j) * 2 + 1) * 64 + lc, im[j]);
                }
            } else if (!FM && ch == 0) qkv(std::integral_constant<int, 0>{});
            lds_barrier();
            if (grp == 0) {
                for (int i = tid; i < 512; i += 256) {
                    const int j = i >> 7, rc = i & 127;
                    if (k0 + j < cb) {
                        float acc = 0.f;
#pragma unroll
                        for (int g = 0; g < 16; ++g) acc += part[g * 512 + i];
                        sSpec[(k0 + j) * 128 + rc] = acc;
                    }
                }
            } else if (!FM && ch == 0) qkv(std::integral_constant<int, 1>{});
            lds_barrier();
        }
        if (!FM && grp == 1) qkv(std::integral_constant<int, 2>{});
        if constexpr (FM) {      // Y_k = X_k W_k (complex, per bin and feature; complex_weight [cb][64][2], fmlprec.py:103-108)
            const float* const cwp = KARG(FusedFwdP, filter_cw);
            for (int i = tid; i < cb * 64; i += 512) {
                const int k = i >> 6, c = i & 63;
                const float xr = sSpec[k * 128 + c], xi = sSpec[k * 128 + 64 + c];
                const float wr = gld(cwp + 2 * i), wi = gld(cwp + 2 * i + 1);
                sSpec[k * 128 + c] = xr * wr - xi * wi;
                sSpec[k * 128 + 64 + c] = xr * wi + xi * wr;
            }
            lds_barrier();
        }
        // low-pass, beta^2 high-pass, dropout, residual, LayerNorm -> sD: group 0 takes rows 0..47 while group 1 is
        // still on the V projection, group 1 takes rows 48..63 afterwards
        {
            f32x4 b2 = gld4(R2_sqrt_beta + lc);
            b2 = b2 * b2;
            if constexpr (FM) b2 = f32x4{0, 0, 0, 0};              // f = the filtered signal itself (no high-pass remainder)
            float* const R2_hmix = KARG(FusedFwdP, hmix);
            const f32x4 g = gld4(R2_f_g + lc), be = gld4(R2_f_b + lc);
            const int rbeg = grp == 0 ? 0 : 48, rend = grp == 0 ? 48 : 64;
            for (int r0 = rbeg; r0 < rend; r0 += 16) {
                const int t = r0 + lr;
                const bool ok = t < L;
                const long e = (tok0 + t) * 64 + lc;
                f32x4 v = {0, 0, 0, 0};
                if (ok) {
                    const f32x4 xv = ld4(sX + t * FS + lc);
                    const f32x4 low = lowpass_tab(sSpec, t, lc, L, cb, sTab);
                    v = (low + b2 * (xv - low)) * drop_mult4(R2_drop_f, dseed, (uint64_t)e >> 2) + xv;
                }
                const float mean = group_sum<16>(v.x + v.y + v.z + v.w) * (1.0f / 64.0f);
                f32x4 dl = {0, 0, 0, 0};
                if (ok) dl = v - mean;
                const float var = group_sum<16>(dl.x * dl.x + dl.y * dl.y + dl.z * dl.z + dl.w * dl.w) * (1.0f / 64.0f);
                const float rs = 1.0f / sqrtf(var + R2_eps);
                f32x4 y = {0, 0, 0, 0};
                if (ok) {
                    const f32x4 xh = dl * rs;
                    y = g * xh + be;
                    ast4<BF>(R2_xhat_f, e, xh);
                    if (R2_dsp) gst4(R2_dsp + e, y);
                    if constexpr (FM) gst4(R2_hmix + e, y);          // alpha = 1: the block's mixed activation IS this output
                    if (lc == 0) gst(R2_rstd_f + (tok0 + t), rs);
                }
                st4((FM ? sH : sD) + t * FS + lc, y);
            }
        }
    }
    // dense weights for phase 4 (this group's K half), held across the attention
    WFrag<MM, 32> wO;
    if constexpr (!FM) load_w<MM, 32>(R2_wo, wrow + 32 * grp, wO);
    lds_barrier();
    if constexpr (!FM) {
    // q, k, v -> global as whole rows, 16 B per lane (per-lane dword stores are store-issue bound)
    {
        const int lr = tid >> 4, lc = (tid & 15) << 2;
#pragma unroll
        for (int r0 = 0; r0 < 64; r0 += 32) {
            const int r = r0 + lr;
            if (r < L) {
                const long e = (tok0 + r) * 64 + lc;
                ast4<BF>(R2_q, e, ld4(sQ + r * FS + lc));
                ast4<BF>(R2_k, e, ld4(sK + r * FS + lc));
                ast4<BF>(R2_v, e, f32x4{sVt[lc * FS + r], sVt[(lc + 1) * FS + r], sVt[(lc + 2) * FS + r], sVt[(lc + 3) * FS + r]});
            }
        }
    }

    STAMP(3);
    const auto R3_ctx = KARG(FusedFwdP, ctx);
    const auto R3_drop_p = KARG(FusedFwdP, drop_p);
    const auto R3_probs = KARG(FusedFwdP, probs);
    // ---- phase 3: attention, transposed: lane = query, registers = keys    src/model/_modules.py:118-135
    {
        const int nt = (L + 31) >> 5;                        // token tiles actually populated (1 or 2)
        const float sqrt_dh = sqrtf((float)DH);
        constexpr int NCT = (DH + 31) / 32;
        const int ncombo = heads * nt;
        const int pair = wave >> 1, kt = wave & 1;
        for (int c0 = 0; c0 < ncombo; c0 += 4) {
            const int combo = c0 + pair;
            const bool act = combo < ncombo && kt < nt;
            const int head = act ? combo / nt : 0, qt = act ? combo % nt : 0;
            const int query = 32 * qt + l31;
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
            float mx = -INFINITY;
            if (act) {
                mma_ll<MM, DH>(sK + (32 * kt + l31) * FS + head * DH + KH, sQ + query * FS + head * DH + KH, st);
                // scale, mask (-10000, additive, fp32)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * kt + rho(r) + 4 * half;
                    float s = -INFINITY;
                    if (key < L) s = st[r] / sqrt_dh + ((key <= query && sIds[key] > 0) ? 0.0f : -10000.0f);
                    st[r] = s;
                    mx = fmaxf(mx, s);
                }
                mx = xor32_max(mx);
                sRed[(pair * 2 + kt) * 64 + lane] = mx;
            }
            lds_barrier();
            float sum = 0.f;
            if (act) {
                if (nt == 2) mx = fmaxf(mx, sRed[(pair * 2 + (kt ^ 1)) * 64 + lane]);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * kt + rho(r) + 4 * half;
                    const float e = key < L ? __expf(st[r] - mx) : 0.f;
                    st[r] = e;
                    sum += e;
                }
                sum = xor32_sum(sum);
            }
            lds_barrier();                                 // every partner has read the maxima
            if (act) sRed[(pair * 2 + kt) * 64 + lane] = sum;
            lds_barrier();
            if (act) {
                if (nt == 2) sum += sRed[(pair * 2 + (kt ^ 1)) * 64 + lane];
                const float inv = 1.0f / sum;
                // probabilities -> global (16 B per lane), then dropout in place
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int key0 = 32 * kt + 8 * g + 4 * half;
                    f32x4 p = {st[4 * g] * inv, st[4 * g + 1] * inv, st[4 * g + 2] * inv, st[4 * g + 3] * inv};
                    f32x4 m = {1.f, 1.f, 1.f, 1.f};
                    if (query < L && key0 < Lp) {
                        const long e = (((long)b * heads + head) * L + query) * Lp + key0;
                        ast4<BF>(R3_probs, e, p);
                        m = drop_mult4(R3_drop_p, dseed, (uint64_t)e >> 2);
                    }
                    p = p * m;
                    st[4 * g] = p.x; st[4 * g + 1] = p.y; st[4 * g + 2] = p.z; st[4 * g + 3] = p.w;
                }
                // partial ctx^T over this wave's key tile  (P^T accumulators are the B operand as they stand)
                float* dstT = kt == 0 ? sC : sE;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    f32x16 cacc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) cacc[r] = 0.f;
                    const bool crow_ok = 32 * ct + l31 < DH;
                    const float* va = sVt + (head * DH + 32 * ct + (crow_ok ? l31 : 0)) * FS + 32 * kt + 4 * half;
                    if constexpr (MM != 0) {
                        // the P^T accumulator tile is the B operand as it stands: registers 8s .. 8s+7 of a lane half are
                        // keys 16 s + 8 (j >> 2) + 4 h + (j & 3), so the V^T fragment takes the same two groups of 4 keys
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) {
                            f32x4 a0 = ld4(va + 16 * s2), a1 = ld4(va + 16 * s2 + 8);
                            if (!crow_ok) { a0 = f32x4{0, 0, 0, 0}; a1 = a0; }
                            if constexpr (MM == 1) {
                                const u32x4 pb = {pk_bf16(st[8 * s2], st[8 * s2 + 1]), pk_bf16(st[8 * s2 + 2], st[8 * s2 + 3]),
                                                  pk_bf16(st[8 * s2 + 4], st[8 * s2 + 5]), pk_bf16(st[8 * s2 + 6], st[8 * s2 + 7])};
                                cacc = mfma_bf16(pk8(a0, a1), pb, cacc);
                            } else {
                                cacc = mfma_x3(split3(a0, a1), split3(f32x4{st[8 * s2], st[8 * s2 + 1], st[8 * s2 + 2], st[8 * s2 + 3]},
                                                                      f32x4{st[8 * s2 + 4], st[8 * s2 + 5], st[8 * s2 + 6], st[8 * s2 + 7]}), cacc);
                            }
                        }
                    } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 a = ld4(va + 8 * g);
                        if (!crow_ok) a = f32x4{0, 0, 0, 0};
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            cacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], st[4 * g + j], cacc, 0, 0, 0);
                    }
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c = 32 * ct + 8 * g + 4 * half;
                        if (c < DH)
                            st4(dstT + query * FS + head * DH + c, f32x4{cacc[4 * g], cacc[4 * g + 1], cacc[4 * g + 2], cacc[4 * g + 3]});
                    }
                }
            }
            lds_barrier();                                 // sRed is reused by the next round of combos
        }
        // context = sum of the key-tile partials -> sC (rows >= 32*nt are zero) and global, 16 B per lane
        {
            const int lr = tid >> 4, lc = (tid & 15) << 2;
#pragma unroll
            for (int r0 = 0; r0 < 64; r0 += 32) {
                const int r = r0 + lr;
                f32x4 v = {0, 0, 0, 0};
                if (r < 32 * nt) { v = ld4(sC + r * FS + lc); if (nt == 2) v += ld4(sE + r * FS + lc); }
                st4(sC + r * FS + lc, v);
                if (r < L) ast4<BF>(R3_ctx, (tok0 + r) * 64 + lc, v);
            }
        }
    }
    lds_barrier();
    }   // !FM: q / k / v write-out and the attention

    STAMP(4);
    const auto R4_a_b = KARG(FusedFwdP, a_b);
    const auto R4_a_g = KARG(FusedFwdP, a_g);
    const auto R4_alpha = KARG(FusedFwdP, alpha);
    const auto R4_bo = KARG(FusedFwdP, bo);
    const auto R4_drop_o = KARG(FusedFwdP, drop_o);
    const auto R4_eps = KARG(FusedFwdP, eps);
    const auto R4_hmix = KARG(FusedFwdP, hmix);
    const auto R4_oma = KARG(FusedFwdP, oma);
    const auto R4_rstd_a = KARG(FusedFwdP, rstd_a);
    const auto R4_w1 = KARG(FusedFwdP, w1);
    const auto R4_xhat_a = KARG(FusedFwdP, xhat_a);
    float ffn_bias[2];
    // ---- phase 4: dense (K split across the groups) + dropout + residual + LayerNorm + alpha mix
    LnPre lnA;
    if constexpr (!FM) lnA = ln_pre(R4_bo, R4_a_g, R4_a_b, R4_drop_o, dseed, tok0);
    {
        load_w<MM, 64>(R4_w1, (long)(128 * grp + col) * 64 + KH, wA);      // first dense_1 block of this group
        ffn_bias[0] = gld(KARG(FusedFwdP, b1) + 128 * grp + col); ffn_bias[1] = gld(KARG(FusedFwdP, b1) + 128 * grp + 64 + col);
        if constexpr (!FM) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        mma_w<MM, 32>(sC + arow + 32 * grp, wO, acc);
        float* part = grp == 0 ? sQ : sK;                                  // sQ / sK are dead: partial tiles
#pragma unroll
        for (int r = 0; r < 16; ++r) part[(wm * 32 + rho(r) + 4 * half) * FS + col] = acc[r];
        }
    }
    if constexpr (!FM) {
    lds_barrier();
    ln_rows_64<true, BF>(lnA, sQ, sK, sX, R4_eps, sD, R4_alpha, R4_oma, tok0, L, sH, R4_hmix, R4_xhat_a, R4_rstd_a);
    lds_barrier();
    }

    STAMP(5);
    const auto R5_b1 = KARG(FusedFwdP, b1);
    const auto R5_u = KARG(FusedFwdP, u);
    const auto R5_w1 = KARG(FusedFwdP, w1);
    const auto R5_w2 = KARG(FusedFwdP, w2);
    // ---- phase 5: dense_1 pre-activation -> sU: group g owns blocks 2g, 2g+1
    {
        const float* sa = sH + arow;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c256 = (2 * grp + i) * 64 + col;
            WFrag<MM, 64>& wcur = i ? wB : wA;
            WFrag<MM, 64>& wnxt = i ? wA : wB;
            if (i == 0) load_w<MM, 64>(R5_w1, (long)(c256 + 64) * 64 + KH, wnxt);
            else load_w<MM, 64>(R5_w2, (long)col * 256 + 128 * grp + KH, wnxt);     // first dense_2 chunk of this group
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            mma_w<MM, 64>(sa, wcur, acc);
            const float bias = ffn_bias[i];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + rho(r) + 4 * half;
                sU[row * FU + c256] = acc[r] + bias;
            }
        }
    }
    lds_barrier();

    STAMP(6);
    const auto R6_w2 = KARG(FusedFwdP, w2);
    const auto R6_u = KARG(FusedFwdP, u);
    const auto R6_gp = KARG(FusedFwdP, gp);
    const auto R7_b2 = KARG(FusedFwdP, b2);
    const auto R7_drop_ff = KARG(FusedFwdP, drop_ff);
    const auto R7_ff_b = KARG(FusedFwdP, ff_b);
    const auto R7_ff_g = KARG(FusedFwdP, ff_g);
    LnPre lnF;
    // ---- phase 6: erf-GELU pass (+ write-out for the backward), dense_2 (K split: group g owns inner columns [128g, 128g+128)) + dropout + residual + LN
    {
        // gelu(u) replaces the pre-activation in LDS, once per element (applying it in dense_2's operand loads would
        // evaluate it once per column tile); gelu(u) and gelu'(u) -> global as whole rows for the backward
        for (int idx = tid; idx < 64 * 64; idx += 512) {
            const int r = idx >> 6, c4 = (idx & 63) << 2;
            const f32x4 v = ld4(sU + r * FU + c4);
            float g0, g1, g2, g3, p0, p1, p2, p3;
            gelu_both(v.x, g0, p0); gelu_both(v.y, g1, p1); gelu_both(v.z, g2, p2); gelu_both(v.w, g3, p3);
            const f32x4 g = {g0, g1, g2, g3}, gp = {p0, p1, p2, p3};
            if (r < L) { ast4<BF>(R6_u, (tok0 + r) * 256 + c4, g); ast4<BF>(R6_gp, (tok0 + r) * 256 + c4, gp); }
            st4(sU + r * FU + c4, g);
        }
        lds_barrier();
        lnF = ln_pre(R7_b2, R7_ff_g, R7_ff_b, R7_drop_ff, dseed, tok0);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* sa = sU + (wm * 32 + l31) * FU + 128 * grp + KH;
        load_w<MM, 64>(R6_w2, (long)col * 256 + 128 * grp + 64 + KH, wB);
        mma_w<MM, 64>(sa, wA, acc);                          // chunk 0 of this group sits in wA
        mma_w<MM, 64>(sa + 64, wB, acc);
        float* part = grp == 0 ? sX : sE;                    // sX / sE are dead: partial tiles
#pragma unroll
        for (int r = 0; r < 16; ++r) part[(wm * 32 + rho(r) + 4 * half) * FS + col] = acc[r];
    }
    lds_barrier();
    STAMP(7);
    TopFwdRegs<BF> TR;
    TopFwdHelpRegs<BF> TH;
    if constexpr (TAIL) { if (wave < 4) top_fwd_prefetch<BF, KOFF, false>(TR); else top_fwd_help_prefetch<BF, KOFF>(TH); }
    const auto R7_Xout = KARG(FusedFwdP, Xout);
    const auto R7_eps = KARG(FusedFwdP, eps);
    const auto R7_rstd_ff = KARG(FusedFwdP, rstd_ff);
    const auto R7_xhat_ff = KARG(FusedFwdP, xhat_ff);
    ln_rows_64<false, BF>(lnF, sX, sE, sH, R7_eps, nullptr, 0.f, 1.f, tok0, L,
                          TAIL ? sD : nullptr, R7_Xout, R7_xhat_ff, R7_rstd_ff, KARG(FusedFwdP, xout_f32) != 0, TAIL);
    STAMP(8);
    if constexpr (TAIL) {
        // x tile of the top block = sD; K, V tiles and the DFT partials alias the dead feed-forward tile; row vectors alias
        // tile 0 (read by the row pass above until the barrier)
        lds_barrier();
        if (wave < 4) top_fwd_rest<DH, BF, KOFF, true>(TR, dseed, sD, sR, sR + TS, sR + 2 * TS, sTab, sSpec, sX, sIds);
        else top_fwd_help<DH, BF, KOFF>(TH, sD, sR, sR + TS, sX);
    }
}
#undef PTYPE

static inline size_t fused_fwd_smem_bytes() {
    return (size_t)(8 * 64 * FS + 2 * FUSED_MAX_CB * 128 + 512 + 64) * 4;
}

// =============================================================================================
// Fused backward of one BSARecBlock (input-gradient chain): one workgroup per sequence.
//   dY -> [FFN LN bwd] -> dT2 -> dU = (dT2.W2) gelu'(u) -> dH = dU.W1 + dz -> [attn LN / filter LN bwd]
//   -> dO -> dC = dO.Wo -> per head {dA, dS, dV, dK, dQ} -> dQ.Wq + dK.Wk + dV.Wv + dzA + dzF
//   -> + FrequencyLayer backward -> dX.
// Weight/bias gradients are NOT formed here (a per-sequence partial would be 196 KB): dT2, dU, dO, dq, dk,
// dv are written out for the grouped token-parallel split-K product; LayerNorm gamma/beta and sqrt_beta
// partials are written per sequence ([B][64]) and reduced by multi_reduce_kernel.
// "x . W" products need W^T as the MFMA B operand: B[k][j] = W[k][j] is read with coalesced dword loads
// (lane = j), 4 per 8-deep k-block, k order permuted to match the b128 A fragments.
// =============================================================================================
struct FusedBwdP {
    const float* dY; float* dX; const float* X;
    const float *sqrt_beta, *f_g, *wq, *wk, *wv, *wo, *a_g, *w1, *w2, *ff_g;
    const float* tw;
    const float *xhat_f, *rstd_f, *q, *k, *v, *probs, *xhat_a, *rstd_a, *u, *xhat_ff, *rstd_ff;    // u: gelu'(pre-activation), as the forward saved it
    const float* dh_slabs; int dh_nsplit; long dh_stride;   // top layer: dY = 0 except row L-1 = sum of split-K slabs [s][B][64]
    float *dT, *dU, *dO, *dq, *dk, *dv;                     // operands of the weight-gradient products
    float *pg_ff, *pb_ff, *pg_a, *pb_a, *pg_f, *pb_f, *pbeta;   // [B][64] partials
    int L, Lp, cb, heads;
    float alpha, oma;
    DropP drop_f, drop_p, drop_o, drop_ff;
    float* trash;
    long long* stamps;
    // bottom block only (e_dz != null): the embedding front-end's backward rides in the epilogue --
    // y = Drop(LN(e)) (src/model/_abstract_model.py:14-24): de = LNbwd(dX * keep/(1-p)) -> e_dz instead of dX
    float* e_dz; const float *e_xhat, *e_rstd, *e_g; float *e_pg, *e_pb; DropP e_drop;
    const float* filter_cw; float* pcw;     // FM instantiation only: FMLPRec's complex_weight [cb][64][2]; per-sequence d(complex_weight) [B][cb][64][2]
    const float* e_dx_extra;      // bottom block, optional: an upstream gradient of the EMBEDDING output itself (fp32 [B, L, 64]),
                                  // added to dX before the embedding LayerNorm backward (forward(all_sequence_output=True)[0])
};


// column sums held per lane (over this thread's rows) -> [64] partial of the sequence, via LDS scratch [rows][64]
__device__ __forceinline__ void seq_partial_64(const f32x4& v, float* __restrict__ red, float* __restrict__ dst, float scale_by = 1.f,
                                               const float* __restrict__ mul = nullptr) {
    const int lr = threadIdx.x >> 4, lc = (threadIdx.x & 15) << 2;
    const int ng = blockDim.x >> 4;
    lds_barrier();
    st4(red + lr * 64 + lc, v);
    lds_barrier();
    if (threadIdx.x < 64) {
        float s = 0.f;
        for (int g = 0; g < ng; ++g) s += red[g * 64 + threadIdx.x];
        if (mul) s *= gld(mul + threadIdx.x);
        gst(dst + threadIdx.x, s * scale_by);
    }
}


// N such partials with ONE pair of barriers (the one-at-a-time form costs two barriers per vector: eight in the
// mix / LayerNorm stage alone): vector i goes through red + i * 2048, thread (i, c) = (tid >> 6, tid & 63) sums column c.
template <int N>
__device__ __forceinline__ void seq_partials_64(const f32x4 (&v)[N], float* const (&red)[N], float* const (&dst)[N],
                                                float scale0 = 1.f, const float* __restrict__ mul0 = nullptr) {
    const int lr = threadIdx.x >> 4, lc = (threadIdx.x & 15) << 2;
    const int ng = blockDim.x >> 4;
    lds_barrier();
#pragma unroll
    for (int i = 0; i < N; ++i) st4(red[i] + lr * 64 + lc, v[i]);
    lds_barrier();
    const int i = threadIdx.x >> 6, c = threadIdx.x & 63;
    if (i < N) {
        const float* r = red[0];
        float* d = dst[0];
#pragma unroll
        for (int k = 1; k < N; ++k) if (i == k) { r = red[k]; d = dst[k]; }
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int g = 0; g < ng; g += 4) {
            s0 += r[g * 64 + c]; s1 += r[(g + 1) * 64 + c]; s2 += r[(g + 2) * 64 + c]; s3 += r[(g + 3) * 64 + c];
        }
        float sum = (s0 + s1) + (s2 + s3);
        if (i == 0) { if (mul0) sum *= gld(mul0 + c); sum *= scale0; }
        gst(d + c, sum);
    }
}

// Backward: 8 waves = 2 groups of 4, two waves per SIMD.  dU splits its four 64-wide blocks across the groups;
// dH, dC and the QKV input-gradient split K (two partial tiles, summed by the next row pass); attention backward
// runs (query tile, key tile) per wave and the 6 dQ/dK/dV tiles of a head on 6 of the 8 waves; the two DFT
// sources of the FrequencyLayer backward run one per group.
// HEADP = TopBwdP: the block above is the one-row top block of the loss path; its backward runs first, inside this
// kernel, on waves 0..3 (its dX tile stays in LDS), while waves 4..7 stage this block's gelu' tile.
template <int DH, bool BF, class HEADP, bool X3 = false, bool FM = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
fused_layer_bwd_kernel(const FusedBwdP P_unused, const HEADP H_unused) {
#define PTYPE FusedBwdP
    static_assert(!(BF && X3), "x3 products work on fp32 tensors");
    constexpr int MM = X3 ? 2 : (BF ? 1 : 0);
    // FM: backward of the sibling model FMLPRec's block (see the forward): stages A1..A3 (feed-forward), then its own tail
    static_assert(!FM || (!BF && !X3 && !IsTail<HEADP>::value), "the FMLPRec block runs in fp32, without a top-block head");
    constexpr bool HEAD = IsTail<HEADP>::value;
    constexpr unsigned KOFF = (unsigned)((sizeof(FusedBwdP) + 7) & ~(size_t)7);     // kernarg offset of H_unused
    const auto R0_L = KARG(FusedBwdP, L);
    const auto R0_Lp = KARG(FusedBwdP, Lp);
    const auto R0_cb = KARG(FusedBwdP, cb);
    const auto R0_heads = KARG(FusedBwdP, heads);
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int TS = 64 * FS;
    float* sAcc = sm;                 // T0: dz (FFN) -> dzA + dzF -> DFT spectra
    float* sT = sm + TS;              // T1: dT2 -> dO -> x tile
    float* sQ = sm + 2 * TS;          // T2..T5 alias dU [64][FU] in the FFN stage and the DFT partials at the end
    float* sK = sm + 3 * TS;
    float* sV = sm + 4 * TS;
    float* sS = sm + 5 * TS;          // dS^T [key][query] of the current head
    float* sdU = sQ;
    float* sG = sm + 6 * TS;          // T6: partial tile 0 / dC / dX before the filter term
    float* sdF = sm + 7 * TS;         // T7: partial tile of dH, then dF
    float* sPm = sm + 8 * TS;         // T8: Drop(P)^T [key][query] | reduction scratch | partial tile 1
    float* sTab = sm + 9 * TS;        // FUSED_MAX_CB * 128
    float* sRed = sTab + FUSED_MAX_CB * 128;   // 512: delta exchange [2 query tiles][2 key tiles][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int grp = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int lr = tid >> 4, lc = (tid & 15) << 2;
    const int L = R0_L, Lp = R0_Lp, heads = R0_heads, cb = R0_cb;
    const int b = blockIdx.x;
    const long tok0 = (long)b * L;
    const int col = wn * 32 + l31;
    constexpr int KHM = MM ? 8 : 4;
    const int KH = KHM * half;
    const int arow = (wm * 32 + l31) * FS + KH;
    float* const trash = KARG(FusedBwdP, trash) + 4 * lane;

    STAMP(0);
    const KernargTouch ktouch = kernarg_touch<KOFF + (unsigned)sizeof(HEADP)>();
    // first loads of the kernel: everything the head reads that does not depend on its own results -- wave 0 the operands of
    // the chain that opens it, waves 1..3 a tile each, waves 4..7 their weight columns -- so that these ~180 KB cross the CU's
    // memory pipeline under the twiddle-table build, not between the head's barriers
    TopBwdRegs HR;
    TopBwdHelpRegs HH;
    if constexpr (HEAD) {
        if (wave == 0) top_bwd_prefetch<BF, KOFF>(HR);
        else if (wave < 4) {
            // wave 3 (the one without a dropout multiplier to evaluate in the head's first step) builds the twiddle table, ahead
            // of its tile request: a load behind 16 KB of tile rows would not return for thousands of cycles
            if (wave == 3) build_twiddle_table(KARG(FusedBwdP, tw), L, cb, sTab, 192, 64);
            top_bwd_prefetch_tile<BF, KOFF>(HR);
        } else top_bwd_help_prefetch<BF, KOFF>(HH);
    }
    const DropSeed dseed = drop_seed(KARG(FusedBwdP, drop_f));
    kernarg_touch_done(ktouch);
    const auto R1_dT = KARG(FusedBwdP, dT);
    const auto R1_dY = KARG(FusedBwdP, dY);
    const auto R1_dh_nsplit = KARG(FusedBwdP, dh_nsplit);
    const auto R1_dh_slabs = KARG(FusedBwdP, dh_slabs);
    const auto R1_dh_stride = KARG(FusedBwdP, dh_stride);
    const auto R1_drop_ff = KARG(FusedBwdP, drop_ff);
    const auto R1_ff_g = KARG(FusedBwdP, ff_g);
    const auto R1_pb_ff = KARG(FusedBwdP, pb_ff);
    const auto R1_pg_ff = KARG(FusedBwdP, pg_ff);
    const auto R1_rstd_ff = KARG(FusedBwdP, rstd_ff);
    const auto R1_tw = KARG(FusedBwdP, tw);
    const auto R1_w2 = KARG(FusedBwdP, w2);
    const auto R1_xhat_ff = KARG(FusedBwdP, xhat_ff);
    if constexpr (!FM && !HEAD) build_twiddle_table(R1_tw, L, cb, sTab);      // (FM: all L/2 + 1 bins do not fit here; built in its tail)
    if constexpr (HEAD) {
        // (no barrier here: the head first reads the twiddle table after its third barrier, and every wave has written its
        // part of the table before it reaches the head's first one)
        if (wave < 4) {
            // top block's tiles alias T6..T8, its row vectors T0, its dX tile T1 (= this block's dY); T2..T5 stay free for
            // the gelu' tile the other waves are staging
            top_bwd_body<DH, BF, KOFF, true>(HR, dseed, sG, sdF, sPm, sTab, sAcc, sT);
        } else {
            top_bwd_help_a<BF, KOFF>(HH, sAcc);
            // this block's gelu' tile trickles into registers during the head's narrow steps (top_bwd_help_b) and is stored
            // once the head is through: stage A1's barrier below orders the stores before stage A2
            f32x4 uw[16];
            top_bwd_help_b<DH, BF, KOFF>(sG, sAcc, KARG(FusedBwdP, u), uw);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int idx = (tid - 256) + i * 256, r = idx >> 6, c4 = (idx & 63) << 2;
                st4(sdU + r * FU + c4, r < L ? uw[i] : f32x4{0, 0, 0, 0});
            }
        }
    }
    WFrag<MM, 64> wA, wB;
    WFrag<MM, 32> wk4;                              // stage D's key-weight fragments (requested at the end of stage C)
    f32x4 e_bt, e_x[2];                             // stage E's sqrt_beta / x tile rows (requested there too)
    load_wT<MM, 64, 256>(R1_w2, (long)KH * 256 + 128 * grp + col, wA);             // first dU block of this group
    // stage A1's operands are requested BEFORE the u tile: loads return in issue order, so the LayerNorm row pass waits
    // only for them while the 64 KB of u are still in flight
    const f32x4 g = gld4(R1_ff_g + lc);
    f32x4 dy[2], xh[2];
    float rs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 32 * i + lr;
        const long e = (tok0 + r) * 64 + lc;
        dy[i] = f32x4{0, 0, 0, 0}; xh[i] = dy[i]; rs[i] = 0.f;
        if (R1_dh_slabs) {                              // only the last position feeds the loss (bsarec.py:32)
            if (r < L) {
                if (r == L - 1)
                    for (int sp = 0; sp < R1_dh_nsplit; ++sp) dy[i] += gld4(R1_dh_slabs + sp * R1_dh_stride + (long)b * 64 + lc);
                xh[i] = ald4<BF>(R1_xhat_ff, e); rs[i] = gld(R1_rstd_ff + tok0 + r);
            }
        } else {                                        // branch-free (rows past L re-read row L-1, then zeroed)
            const bool ok = r < L;
            const long ec = (tok0 + min(r, L - 1)) * 64 + lc;
            f32x4 d4 = {0, 0, 0, 0};                    // HEAD: the upstream gradient comes out of the head below, through LDS
            if constexpr (!HEAD) d4 = ald4<BF>(R1_dY, ec);
            const f32x4 x4 = ald4<BF>(R1_xhat_ff, ec);
            const float r1 = gld(R1_rstd_ff + tok0 + min(r, L - 1));
            if constexpr (HEAD) d4 = ld4(sT + r * FS + lc);          // the head's dX tile
            dy[i] = ok ? d4 : f32x4{0, 0, 0, 0}; xh[i] = ok ? x4 : f32x4{0, 0, 0, 0}; rs[i] = ok ? r1 : 0.f;
        }
    }
    f32x4 uv[8];
    if constexpr (!HEAD)
    {   // gelu'(pre-activation) tile [64][256] (saved by the forward) -> registers now, LDS after stage A1's row pass; consumed by stage A2
        const float* gu = KARG(FusedBwdP, u);
        // branch-free: a predicated load becomes a branch and the loop then waits for every load before issuing the next
        // (8 serial round trips); rows past L re-read row L-1 (a valid address) and are zeroed on the way to LDS
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + i * 512, r = idx >> 6, c4 = (idx & 63) << 2;
            uv[i] = ald4<BF>(gu, (tok0 + min(r, L - 1)) * 256 + c4);
        }
        // (the tile goes to LDS after the LayerNorm row pass below: its 64 KB arrive while that pass computes)
    }

    // ---- stage A1: FeedForward LayerNorm backward (row pass): dz -> sAcc, dT2 -> sT, global
    {
        f32x4 sg = {0, 0, 0, 0}, sb = sg;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 32 * i + lr;
            const bool ok = r < L;
            const long e = (tok0 + r) * 64 + lc;
            const f32x4 gg = dy[i] * g;
            const float m1 = group_sum<16>(gg.x + gg.y + gg.z + gg.w) * (1.0f / 64.0f);
            const float m2 = group_sum<16>(gg.x * xh[i].x + gg.y * xh[i].y + gg.z * xh[i].z + gg.w * xh[i].w) * (1.0f / 64.0f);
            const f32x4 dz = rs[i] * (gg - m1 - xh[i] * m2);
            sg += dy[i] * xh[i]; sb += dy[i];
            f32x4 dt = {0, 0, 0, 0};
            if (ok) { dt = dz * drop_mult4(R1_drop_ff, dseed, (uint64_t)e >> 2); ast4<BF>(R1_dT, e, dt); }
            st4(sAcc + r * FS + lc, dz);
            st4(sT + r * FS + lc, dt);
        }
        {
            const f32x4 pv[2] = {sg, sb};
            float* const pr[2] = {sPm, sPm + 2048};
            float* const pd[2] = {R1_pg_ff + (long)b * 64, R1_pb_ff + (long)b * 64};
            seq_partials_64<2>(pv, pr, pd);
        }
    }
    if constexpr (!HEAD) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + i * 512, r = idx >> 6, c4 = (idx & 63) << 2;
        st4(sdU + r * FU + c4, r < L ? uv[i] : f32x4{0, 0, 0, 0});
    }
    }
    lds_barrier();

    STAMP(1);
    const auto R2_dU = KARG(FusedBwdP, dU);
    const auto R2_u = KARG(FusedBwdP, u);
    const auto R2_w1 = KARG(FusedBwdP, w1);
    const auto R2_w2 = KARG(FusedBwdP, w2);
    // ---- stage A2: dU = (dT2 . W2) * gelu'(u) -> sdU (in place over the staged u tile): group g owns blocks 2g, 2g+1
    if constexpr (MM == 0) {
        // fp32: block 1's MFMA chain (one accumulator, 64 cycles per dependent MFMA) is interleaved with block 0's
        // gelu' epilogue -- 2 MFMAs, then one output element (~30 VALU + an LDS round trip) -- so the vector work runs in
        // the issue slots the chain leaves empty instead of after it; both waves of a SIMD otherwise reach their MFMA
        // bursts and their epilogues together (lockstep) and the matrix pipe idles during every epilogue.
        const float* sa = sT + arow;
        const int c0 = (2 * grp) * 64 + col, c1 = c0 + 64;
        load_wT<MM, 64, 256>(R2_w2, (long)KH * 256 + c1, wB);                          // block 1's weights
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        mma_w<MM, 64>(sa, wA, acc0);
        load_wT<MM, 64, 64>(R2_w1, (long)(128 * grp + KH) * 64 + col, wA);             // first dH chunk of this group
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const f32x4 a = ld4(sa + 8 * (r >> 1));
            const int s0 = 2 * (r & 1);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s0], wB.w[r >> 1][s0], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s0 + 1], wB.w[r >> 1][s0 + 1], acc1, 0, 0, 0);
            const int row = wm * 32 + rho(r) + 4 * half;
            float* pu = sdU + row * FU + c0;
            *pu = row < L ? acc0[r] * *pu : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + rho(r) + 4 * half;
            float* pu = sdU + row * FU + c1;
            *pu = row < L ? acc1[r] * *pu : 0.f;
        }
    } else {
        const float* sa = sT + arow;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int blk = 2 * grp + i, c256 = blk * 64 + col;
            WFrag<MM, 64>& wcur = i ? wB : wA;
            WFrag<MM, 64>& wnxt = i ? wA : wB;
            if (i == 0) load_wT<MM, 64, 256>(R2_w2, (long)KH * 256 + c256 + 64, wnxt);
            else load_wT<MM, 64, 64>(R2_w1, (long)(128 * grp + KH) * 64 + col, wnxt);      // first dH chunk of this group
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            mma_w<MM, 64>(sa, wcur, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + rho(r) + 4 * half;
                float* pu = sdU + row * FU + c256;
                *pu = row < L ? acc[r] * *pu : 0.f;
            }
        }
    }
    lds_barrier();
    // dU -> global as whole rows, 16 B per lane (operand of the dense_1 weight gradient)
    for (int idx = tid; idx < 64 * 64; idx += 512) {
        const int r = idx >> 6, c4 = (idx & 63) << 2;
        if (r < L) ast4<BF>(R2_dU, (tok0 + r) * 256 + c4, ld4(sdU + r * FU + c4));
    }

    STAMP(2);
    const auto R3_w1 = KARG(FusedBwdP, w1);
    const auto R3_wo = KARG(FusedBwdP, wo);
    // ---- stage A3: dH = dU . W1, K split: group g owns inner units [128g, 128g+128) -> partial tiles sG / sdF
    WFrag<MM, 32> wO;
    f32x4 pq[2], pk[2], pv[2], xa[2], xf[2];                 // stage B1's operands, prefetched below
    float ra[2], rf[2];
    {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* sa = sdU + (wm * 32 + l31) * FU + 128 * grp + KH;
        load_wT<MM, 64, 64>(R3_w1, (long)(128 * grp + 64 + KH) * 64 + col, wB);
        if constexpr (!FM) load_wT<MM, 32, 64>(R3_wo, (long)(32 * grp + KH) * 64 + col, wO);          // dense^T half for stage B2
        // stage B1's operands (q, k, v, xhat of both LayerNorms: 80 KB per sequence) are requested here, AFTER the weight
        // fragments this stage waits for (loads return in issue order), and land while the MFMAs below run
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 32 * i + lr;
            const long ec = (tok0 + min(r, L - 1)) * 64 + lc;
            if constexpr (!FM) {
            pq[i] = ald4<BF>(KARG(FusedBwdP, q), ec); pk[i] = ald4<BF>(KARG(FusedBwdP, k), ec); pv[i] = ald4<BF>(KARG(FusedBwdP, v), ec);
            xa[i] = ald4<BF>(KARG(FusedBwdP, xhat_a), ec); ra[i] = gld(KARG(FusedBwdP, rstd_a) + tok0 + min(r, L - 1));
            }
            xf[i] = ald4<BF>(KARG(FusedBwdP, xhat_f), ec);
            rf[i] = gld(KARG(FusedBwdP, rstd_f) + tok0 + min(r, L - 1));
        }
        mma_w<MM, 64>(sa, wA, acc);                          // chunk 0 of this group sits in wA
        mma_w<MM, 64>(sa + 64, wB, acc);
        float* part = grp == 0 ? sG : sdF;
#pragma unroll
        for (int r = 0; r < 16; ++r) part[(wm * 32 + rho(r) + 4 * half) * FS + col] = acc[r];
    }
    lds_barrier();

    STAMP(3);
    if constexpr (FM) {
        // ================= FMLPRec tail: filter LayerNorm backward, complex-filter backward =================
        // y = irfft(rfft(x) * W), h = LN_f(Drop(y) + x) is the block's mixed activation (no attention branch).  With D the
        // spectrum of dF = Drop'(dz_f):  dX = dz_f + inverse(D conj(W)),  dW_k = (w_k / L) conj(X_k) D_k per sequence
        // (w_k = 1 for DC / Nyquist, else 2) -- the formulas of freq_bwd_kernel (kernels.h), all L/2 + 1 bins.
        // LDS: T0 dz_f | T1 x | T2 / T3 DFT partials of group 0 / 1 | T4-T5 twiddles | T6-T7 spectra [2][cb][2][64] | T8 dF
        float* const sXin = sT;  float* const sdFm = sPm;  float* const tab = sV;  float* const spec = sG;
        const DropP drop_f = KARG(FusedBwdP, drop_f);
        const f32x4 gf = gld4(KARG(FusedBwdP, f_g) + lc);
        const float* const Xg = KARG(FusedBwdP, X);
        f32x4 sgf = {0, 0, 0, 0}, sbf = sgf;
        const f32x4 z4 = {0, 0, 0, 0};
        f32x4 dzf2[2], dF2[2], x2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 32 * i + lr;
            const bool ok = r < L;
            const long e = (tok0 + r) * 64 + lc;
            if (!ok) { xf[i] = z4; rf[i] = 0.f; }
            f32x4 dh = z4;
            if (ok) dh = ld4(sG + r * FS + lc) + ld4(sdF + r * FS + lc) + ld4(sAcc + r * FS + lc);
            const f32x4 g2 = dh * gf;                                  // alpha = 1: the whole gradient goes to the filter branch
            const float n1 = group_sum<16>(g2.x + g2.y + g2.z + g2.w) * (1.0f / 64.0f);
            const float n2 = group_sum<16>(g2.x * xf[i].x + g2.y * xf[i].y + g2.z * xf[i].z + g2.w * xf[i].w) * (1.0f / 64.0f);
            dzf2[i] = rf[i] * (g2 - n1 - xf[i] * n2);
            sgf += dh * xf[i]; sbf += dh;
            dF2[i] = z4; x2[i] = z4;
            if (ok) {
                dF2[i] = dzf2[i] * drop_mult4(drop_f, dseed, (uint64_t)e >> 2);
                x2[i] = gld4(Xg + e);
                // the attention tensors exist in the arena (the sibling model reuses BSARec's layout): their weight-gradient
                // operands are exact zeros
                gst4(KARG(FusedBwdP, dO) + e, z4); gst4(KARG(FusedBwdP, dq) + e, z4);
                gst4(KARG(FusedBwdP, dk) + e, z4); gst4(KARG(FusedBwdP, dv) + e, z4);
            }
        }
        lds_barrier();                                                 // every read of the dH partial tiles (sG, sdF) is done
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 32 * i + lr;
            st4(sAcc + r * FS + lc, dzf2[i]);
            st4(sdFm + r * FS + lc, dF2[i]);
            st4(sXin + r * FS + lc, x2[i]);
        }
        build_twiddle_table(R1_tw, L, cb, tab);
        {
            const f32x4 pv[4] = {z4, z4, sgf, sbf};
            float* const pr[4] = {sQ, sQ + 2048, sK, sK + 2048};
            float* const pd[4] = {KARG(FusedBwdP, pg_a) + (long)b * 64, KARG(FusedBwdP, pb_a) + (long)b * 64,
                                  KARG(FusedBwdP, pg_f) + (long)b * 64, KARG(FusedBwdP, pb_f) + (long)b * 64};
            seq_partials_64<4>(pv, pr, pd);                            // (two barriers inside: tiles and table are visible after it)
        }
        lds_barrier();
        // spectra: group 0 transforms x, group 1 transforms dF; 2 bins per pass (partials [16 row groups][2][2][64] per group)
        {
            const int lr16 = (tid & 255) >> 4;
            const float* srcT = grp == 0 ? sXin : sdFm;
            float* part = grp == 0 ? sQ : sK;
            float* specg = spec + grp * cb * 128;
            for (int k0 = 0; k0 < cb; k0 += 2) {
                f32x4 re[2], im[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) { re[j] = z4; im[j] = z4; }
#pragma unroll
                for (int r0 = 0; r0 < 64; r0 += 16) {
                    const int t = r0 + lr16;
                    if (t < L) {
                        const f32x4 x = ld4(srcT + t * FS + lc);
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            if (k0 + j < cb) {
                                const float c = tab[2 * ((k0 + j) * 64 + t)], sn = tab[2 * ((k0 + j) * 64 + t) + 1];
                                re[j] += x * c; im[j] -= x * sn;
                            }
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    st4(part + ((lr16 * 2 + j) * 2 + 0) * 64 + lc, re[j]);
                    st4(part + ((lr16 * 2 + j) * 2 + 1) * 64 + lc, im[j]);
                }
                lds_barrier();
                {
                    const int i = tid & 255;                           // 2 bins x (re | im) x 64 = 256 sums per group
                    const int j = i >> 7;
                    if (k0 + j < cb) {
                        float acc = 0.f;
#pragma unroll
                        for (int gq = 0; gq < 16; ++gq) acc += part[gq * 256 + i];
                        specg[(k0 + j) * 128 + (i & 127)] = acc;
                    }
                }
                lds_barrier();
            }
        }
        {   // dW partials of this sequence, D <- D conj(W)
            const float* const cwp = KARG(FusedBwdP, filter_cw);
            float* const pcw = KARG(FusedBwdP, pcw) + (long)b * cb * 128;
            float* const sDs = spec + cb * 128;
            for (int i = tid; i < cb * 64; i += 512) {
                const int k = i >> 6, c = i & 63;
                const float xr = spec[k * 128 + c], xi = spec[k * 128 + 64 + c];
                const float dr = sDs[k * 128 + c], di = sDs[k * 128 + 64 + c];
                const float wr = gld(cwp + 2 * i), wi = gld(cwp + 2 * i + 1);
                const float wl = ((k == 0 || 2 * k == L) ? 1.0f : 2.0f) / (float)L;
                gst(pcw + 2 * i, wl * (dr * xr + di * xi));
                gst(pcw + 2 * i + 1, wl * (di * xr - dr * xi));
                sDs[k * 128 + c] = dr * wr + di * wi;
                sDs[k * 128 + 64 + c] = di * wr - dr * wi;
            }
        }
        lds_barrier();
        {   // dX = dz_f + inverse(D conj(W)); bottom block: the embedding front-end's backward on the finished row
            float* const e_dz = KARG(FusedBwdP, e_dz);
            const float* const e_xhat = KARG(FusedBwdP, e_xhat);
            const float* const e_rstd = KARG(FusedBwdP, e_rstd);
            const DropP e_drop = KARG(FusedBwdP, e_drop);
            const float* const e_dx_extra = KARG(FusedBwdP, e_dx_extra);
            float* const dXg = KARG(FusedBwdP, dX);
            f32x4 g0 = z4, sg0 = z4, sb0 = z4;
            if (e_dz) g0 = gld4(KARG(FusedBwdP, e_g) + lc);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int t = 32 * i + lr;
                const bool ok = t < L;
                const long e = (tok0 + t) * 64 + lc;
                f32x4 dx = z4, xh = z4;
                float rs = 0.f;
                if (ok) {
                    dx = ld4(sAcc + t * FS + lc) + lowpass_tab(spec + cb * 128, t, lc, L, cb, tab);
                    if (e_dx_extra) dx += gld4(e_dx_extra + e);
                    if (e_dz) { xh = gld4(e_xhat + e); rs = gld(e_rstd + tok0 + t); dx = dx * drop_mult4(e_drop, dseed, (uint64_t)e >> 2); }
                    else gst4(dXg + e, dx);
                }
                if (e_dz) {
                    const f32x4 gg = dx * g0;
                    const float m1 = group_sum<16>(gg.x + gg.y + gg.z + gg.w) * (1.0f / 64.0f);
                    const float m2 = group_sum<16>(gg.x * xh.x + gg.y * xh.y + gg.z * xh.z + gg.w * xh.w) * (1.0f / 64.0f);
                    if (ok) gst4(e_dz + e, rs * (gg - m1 - xh * m2));
                    sg0 += dx * xh; sb0 += dx;
                }
            }
            float* const pbeta = KARG(FusedBwdP, pbeta) + (long)b * 64;
            if (e_dz) {
                const f32x4 pv[3] = {z4, sg0, sb0};                    // (FMLPRec has no sqrt_beta: its partial is zero)
                float* const pr[3] = {sQ, sQ + 2048, sK};
                float* const pd[3] = {pbeta, KARG(FusedBwdP, e_pg) + (long)b * 64, KARG(FusedBwdP, e_pb) + (long)b * 64};
                seq_partials_64<3>(pv, pr, pd);
            } else if (tid < 64) gst(pbeta + tid, 0.f);
        }
        return;
    }
    const auto R4_a_g = KARG(FusedBwdP, a_g);
    const auto R4_alpha = KARG(FusedBwdP, alpha);
    const auto R4_dO = KARG(FusedBwdP, dO);
    const auto R4_drop_f = KARG(FusedBwdP, drop_f);
    const auto R4_drop_o = KARG(FusedBwdP, drop_o);
    const auto R4_f_g = KARG(FusedBwdP, f_g);
    const auto R4_k = KARG(FusedBwdP, k);
    const auto R4_oma = KARG(FusedBwdP, oma);
    const auto R4_pb_a = KARG(FusedBwdP, pb_a);
    const auto R4_pb_f = KARG(FusedBwdP, pb_f);
    const auto R4_pg_a = KARG(FusedBwdP, pg_a);
    const auto R4_pg_f = KARG(FusedBwdP, pg_f);
    const auto R4_q = KARG(FusedBwdP, q);
    const auto R4_rstd_a = KARG(FusedBwdP, rstd_a);
    const auto R4_rstd_f = KARG(FusedBwdP, rstd_f);
    const auto R4_v = KARG(FusedBwdP, v);
    const auto R4_xhat_a = KARG(FusedBwdP, xhat_a);
    const auto R4_xhat_f = KARG(FusedBwdP, xhat_f);
    // ---- stage B1: alpha-mix + attention LayerNorm / filter LayerNorm backward (row pass)
    //      dO -> sT + global, dF -> sdF, dzA + dzF -> sAcc; q, k, v tiles -> LDS (dU is dead)
    {
        const f32x4 ga = gld4(R4_a_g + lc), gf = gld4(R4_f_g + lc);
        f32x4 sga = {0, 0, 0, 0}, sba = sga, sgf = sga, sbf = sga;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 32 * i + lr;
            const f32x4 z4 = {0, 0, 0, 0};
            f32x4 q4 = pq[i], k4 = pk[i], v4 = pv[i];          // prefetched in stage A3 (rows past L hold row L-1: zeroed here)
            if (r >= L) { q4 = z4; k4 = z4; v4 = z4; xa[i] = z4; xf[i] = z4; ra[i] = 0.f; rf[i] = 0.f; }
            st4(sQ + r * FS + lc, q4); st4(sK + r * FS + lc, k4); st4(sV + r * FS + lc, v4);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 32 * i + lr;
            const bool ok = r < L;
            const long e = (tok0 + r) * 64 + lc;
            f32x4 dh = {0, 0, 0, 0};
            if (ok) dh = ld4(sG + r * FS + lc) + ld4(sdF + r * FS + lc) + ld4(sAcc + r * FS + lc);
            const f32x4 dya = dh * R4_oma, dyf = dh * R4_alpha;
            const f32x4 g1 = dya * ga, g2 = dyf * gf;
            const float m1 = group_sum<16>(g1.x + g1.y + g1.z + g1.w) * (1.0f / 64.0f);
            const float m2 = group_sum<16>(g1.x * xa[i].x + g1.y * xa[i].y + g1.z * xa[i].z + g1.w * xa[i].w) * (1.0f / 64.0f);
            const float n1 = group_sum<16>(g2.x + g2.y + g2.z + g2.w) * (1.0f / 64.0f);
            const float n2 = group_sum<16>(g2.x * xf[i].x + g2.y * xf[i].y + g2.z * xf[i].z + g2.w * xf[i].w) * (1.0f / 64.0f);
            const f32x4 dza = ra[i] * (g1 - m1 - xa[i] * m2), dzf = rf[i] * (g2 - n1 - xf[i] * n2);
            sga += dya * xa[i]; sba += dya; sgf += dyf * xf[i]; sbf += dyf;
            f32x4 dO = {0, 0, 0, 0}, dF = dO;
            if (ok) {
                dO = dza * drop_mult4(R4_drop_o, dseed, (uint64_t)e >> 2);
                dF = dzf * drop_mult4(R4_drop_f, dseed, (uint64_t)e >> 2);
                ast4<BF>(R4_dO, e, dO);
            }
            st4(sAcc + r * FS + lc, dza + dzf);
            st4(sT + r * FS + lc, dO);
            st4(sdF + r * FS + lc, dF);
        }
        {   // (sG's partial tile was consumed by the row pass above: scratch for two of the four vectors)
            const f32x4 pv[4] = {sga, sba, sgf, sbf};
            float* const pr[4] = {sPm, sPm + 2048, sG, sG + 2048};
            float* const pd[4] = {R4_pg_a + (long)b * 64, R4_pb_a + (long)b * 64, R4_pg_f + (long)b * 64, R4_pb_f + (long)b * 64};
            seq_partials_64<4>(pv, pr, pd);
        }
    }
    lds_barrier();

    STAMP(4);
    // ---- stage B2: dC = dO . Wo, K split across the groups (partials sG / sPm), then summed into sG
    {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        mma_w<MM, 32>(sT + arow + 32 * grp, wO, acc);
        float* part = grp == 0 ? sG : sPm;
#pragma unroll
        for (int r = 0; r < 16; ++r) part[(wm * 32 + rho(r) + 4 * half) * FS + col] = acc[r];
    }
    lds_barrier();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 32 * i + lr;
        st4(sG + r * FS + lc, ld4(sG + r * FS + lc) + ld4(sPm + r * FS + lc));
    }
    lds_barrier();

    STAMP(5);
    const auto R6_dk = KARG(FusedBwdP, dk);
    const auto R6_dq = KARG(FusedBwdP, dq);
    const auto R6_drop_p = KARG(FusedBwdP, drop_p);
    const auto R6_dv = KARG(FusedBwdP, dv);
    const auto R6_probs = KARG(FusedBwdP, probs);
    // ---- stage C: attention backward, one head at a time; lane = query, registers = keys
    {
        const int nt = (L + 31) >> 5;
        const float inv_sqrt_dh = 1.0f / sqrtf((float)DH);
        constexpr int NCT = (DH + 31) / 32;
        constexpr int NRES = (6 * NCT + 7) / 8;
        for (int head = 0; head < heads; ++head) {
            const int hc = head * DH;
            // C1: waves 0..3 own one (query tile, key tile) each: Drop(P)^T -> sPm, dS^T -> sS   (as [key][query])
            const int qt = (wave >> 1) & 1, kt = wave & 1;
            const int query = 32 * qt + l31;
            const bool c1 = wave < 4;
            f32x4 pp[4], mm[4];
            f32x16 da;
            float delta = 0.f;
            if (c1) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int key0 = 32 * kt + 8 * g + 4 * half;
                    pp[g] = f32x4{0, 0, 0, 0}; mm[g] = pp[g];
                    if (query < L && key0 < Lp) {
                        const long e = (((long)b * heads + head) * L + query) * Lp + key0;
                        pp[g] = ald4<BF>(R6_probs, e);
                        mm[g] = drop_mult4(R6_drop_p, dseed, (uint64_t)e >> 2);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) da[r] = 0.f;
                if (kt < nt && qt < nt)
                    mma_ll<MM, DH>(sV + (32 * kt + l31) * FS + hc + KH, sG + query * FS + hc + KH, da);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float dA = da[4 * g + j] * mm[g][j];
                        delta += dA * pp[g][j];
                        da[4 * g + j] = dA;
                        sPm[(32 * kt + 8 * g + 4 * half + j) * FS + query] = pp[g][j] * mm[g][j];
                    }
                delta = xor32_sum(delta);
                sRed[(qt * 2 + kt) * 64 + lane] = delta;
            }
            lds_barrier();
            if (c1) {
                delta += sRed[(qt * 2 + (kt ^ 1)) * 64 + lane];      // the other key tile of this query tile
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        sS[(32 * kt + 8 * g + 4 * half + j) * FS + query] = pp[g][j] * (da[4 * g + j] - delta) * inv_sqrt_dh;
            }
            lds_barrier();
            // C2: 6*NCT output tiles [32 tokens x 32 features]: dQ (rows = queries), dK, dV (rows = keys)
            f32x16 res[NRES];
#pragma unroll
            for (int ti = 0; ti < NRES; ++ti) {
                const int t = wave + 8 * ti;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                if (t < 6 * NCT) {
                    const int kind = t / (2 * NCT), rt = (t / NCT) & 1, ct = t % NCT;
                    const int c = 32 * ct + l31;
                    const bool cok = c < DH;
                    const int cc = hc + (cok ? c : 0);
                    if (rt < nt) {
                        const float* bsrc = kind == 0 ? sK : kind == 1 ? sQ : sG;
                        const float* asrc = kind == 2 ? sPm : sS;
                        if constexpr (MM != 0) {
#pragma unroll 2
                            for (int s2 = 0; s2 < 2 * nt; ++s2) {          // K = 32*nt tokens, 16 per bf16 MFMA: k = 16 s2 + 8 h + j
                                f32x4 a0, a1;
                                if (kind == 0) {                           // dQ = dS . K : A k-major from sS
                                    const float* ap = asrc + (16 * s2 + KH) * FS + 32 * rt + l31;
                                    a0 = f32x4{ap[0], ap[FS], ap[2 * FS], ap[3 * FS]};
                                    a1 = f32x4{ap[4 * FS], ap[5 * FS], ap[6 * FS], ap[7 * FS]};
                                } else {                                   // dK = dS^T . Q ; dV = Drop(P)^T . dC
                                    const float* ap = asrc + (32 * rt + l31) * FS + 16 * s2 + KH;
                                    a0 = ld4(ap); a1 = ld4(ap + 4);
                                }
                                const float* bp = bsrc + (16 * s2 + KH) * FS + cc;
                                f32x4 b0 = {bp[0], bp[FS], bp[2 * FS], bp[3 * FS]}, b1 = {bp[4 * FS], bp[5 * FS], bp[6 * FS], bp[7 * FS]};
                                if (!cok) { b0 = f32x4{0, 0, 0, 0}; b1 = b0; }
                                if constexpr (MM == 1) acc = mfma_bf16(pk8(a0, a1), pk8(b0, b1), acc);
                                else acc = mfma_x3(split3(a0, a1), split3(b0, b1), acc);
                            }
                        } else {
#pragma unroll 4
                        for (int kb = 0; kb < 4 * nt; ++kb) {              // K = 32*nt tokens
                            f32x4 a;
                            if (kind == 0) {                               // dQ = dS . K : A k-major from sS
                                const float* ap = asrc + (8 * kb + 4 * half) * FS + 32 * rt + l31;
                                a.x = ap[0]; a.y = ap[FS]; a.z = ap[2 * FS]; a.w = ap[3 * FS];
                            } else {                                       // dK = dS^T . Q ; dV = Drop(P)^T . dC
                                a = ld4(asrc + (32 * rt + l31) * FS + 8 * kb + 4 * half);
                            }
                            const float* bp = bsrc + (8 * kb + 4 * half) * FS + cc;
                            float w0 = bp[0], w1 = bp[FS], w2 = bp[2 * FS], w3 = bp[3 * FS];
                            if (!cok) { w0 = 0.f; w1 = 0.f; w2 = 0.f; w3 = 0.f; }
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w0, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w1, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w2, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w3, acc, 0, 0, 0);
                        }
                        }
                    }
                }
                res[ti] = acc;
            }
            lds_barrier();                   // every read of this head's q / k / v columns is done
#pragma unroll
            for (int ti = 0; ti < NRES; ++ti) {
                const int t = wave + 8 * ti;
                if (t < 6 * NCT) {
                    const int kind = t / (2 * NCT), rt = (t / NCT) & 1, ct = t % NCT;
                    const int c = 32 * ct + l31;
                    if (c < DH) {
                        float* sdst = kind == 0 ? sQ : kind == 1 ? sK : sV;      // in place: dq, dk, dv
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = 32 * rt + rho(r) + 4 * half;
                            const float val = rt < nt ? res[ti][r] : 0.f;
                            sdst[row * FS + hc + c] = val;
                        }
                    }
                }
            }
            lds_barrier();
        }
        // requests of the next two stages, ahead of the write-out below (loads return in issue order; the stores do not hold
        // them up): stage D's weight fragments, and stage E's x tile / sqrt_beta, which would otherwise be asked for at
        // their point of use with the whole round trip exposed
        {
            const long wofs = (long)KH * 64 + col;
            if (grp == 0) { load_wT<MM, 64, 64>(KARG(FusedBwdP, wq), wofs, wA); load_wT<MM, 32, 64>(KARG(FusedBwdP, wk), wofs, wk4); }
            else { load_wT<MM, 32, 64>(KARG(FusedBwdP, wk), wofs + 32 * 64, wk4); load_wT<MM, 64, 64>(KARG(FusedBwdP, wv), wofs, wA); }
            const float* const pX = KARG(FusedBwdP, X);
            e_bt = gld4(KARG(FusedBwdP, sqrt_beta) + lc);
#pragma unroll
            for (int i = 0; i < 2; ++i) e_x[i] = ald4<BF>(pX, (tok0 + min(32 * i + lr, L - 1)) * 64 + lc);
        }
        // dq, dk, dv -> global as whole rows, 16 B per lane (operands of the Q/K/V weight gradients)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 32 * i + lr;
            if (r < L) {
                const long e = (tok0 + r) * 64 + lc;
                ast4<BF>(R6_dq, e, ld4(sQ + r * FS + lc));
                ast4<BF>(R6_dk, e, ld4(sK + r * FS + lc));
                ast4<BF>(R6_dv, e, ld4(sV + r * FS + lc));
            }
        }
    }

    STAMP(6);
    // ---- stage D: dQ.Wq + dK.Wk + dV.Wv, K = 192 split: group 0 = dQ.Wq + dK[:, :32].Wk[:32], group 1 = the rest
    {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        mma_w<MM, 64>((grp == 0 ? sQ : sV) + arow, wA, acc);
        mma_w<MM, 32>(sK + arow + 32 * grp, wk4, acc);
        float* part = grp == 0 ? sG : sPm;
#pragma unroll
        for (int r = 0; r < 16; ++r) part[(wm * 32 + rho(r) + 4 * half) * FS + col] = acc[r];
    }
    lds_barrier();

    STAMP(7);
    const auto R8_dX = KARG(FusedBwdP, dX);
    const auto R8_pbeta = KARG(FusedBwdP, pbeta);
    const auto R8_sqrt_beta = KARG(FusedBwdP, sqrt_beta);
    // ---- stage E: FrequencyLayer backward: dX = (sG + sPm + sAcc) + beta^2 dF + lowpass((1-beta^2) dF); dbeta partial
    {
        float* sXin = sT;
        float* spec = sAcc;                            // [2][cb][2][64]  (sAcc is folded into sG first)
        float* part = (grp == 0 ? sQ : sV);            // [16][4][2][64] = 8192 floats per group (T2-T3 / T4-T5)
        const f32x4 bt = e_bt;
        const f32x4 b2 = bt * bt, omb2 = 1.0f - b2;
        // the embedding LayerNorm backward's operands and dropout masks (bottom block): asked for / evaluated here, used after
        // the transforms below
        float* const e_dz = KARG(FusedBwdP, e_dz);
        const float* const e_dx_extra = KARG(FusedBwdP, e_dx_extra);
        f32x4 g0 = {0, 0, 0, 0}, e_xh[2], e_ex[2], e_dm[2];
        float e_rs[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { e_xh[i] = f32x4{0, 0, 0, 0}; e_ex[i] = e_xh[i]; e_dm[i] = e_xh[i]; e_rs[i] = 0.f; }
        if (e_dz || e_dx_extra) {
            const float* const e_xhat = KARG(FusedBwdP, e_xhat);
            const float* const e_rstd = KARG(FusedBwdP, e_rstd);
            const DropP e_drop = KARG(FusedBwdP, e_drop);
            if (e_dz) g0 = gld4(KARG(FusedBwdP, e_g) + lc);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int t = 32 * i + lr;
                const long e = (tok0 + min(t, L - 1)) * 64 + lc;
                if (e_dx_extra) e_ex[i] = gld4(e_dx_extra + e);
                if (e_dz) { e_xh[i] = ald4<BF>(e_xhat, e); e_rs[i] = gld(e_rstd + tok0 + min(t, L - 1)); e_dm[i] = drop_mult4(e_drop, dseed, (uint64_t)e >> 2); }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 32 * i + lr;
            f32x4 x = e_x[i];
            if (r >= L) x = f32x4{0, 0, 0, 0};
            st4(sG + r * FS + lc, ld4(sG + r * FS + lc) + ld4(sPm + r * FS + lc) + ld4(sAcc + r * FS + lc));
            st4(sXin + r * FS + lc, x);
        }
        lds_barrier();
        // group 0 transforms x, group 1 transforms (1 - beta^2) dF: same code, same barriers
        const int lr16 = (tid & 255) >> 4;
        const float* srcT = grp == 0 ? sXin : sdF;
        float* specg = spec + grp * cb * 128;
#pragma unroll
        for (int ch = 0; ch < FUSED_MAX_CB / 4; ++ch) {
            const int k0 = 4 * ch;
            if (k0 >= cb) break;
            f32x4 re[4], im[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { re[j] = f32x4{0, 0, 0, 0}; im[j] = f32x4{0, 0, 0, 0}; }
#pragma unroll
            for (int r0 = 0; r0 < 64; r0 += 16) {
                const int t = r0 + lr16;
                if (t < L) {
                    f32x4 x = ld4(srcT + t * FS + lc);
                    if (grp == 1) x = x * omb2;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (k0 + j < cb) {
                            const float c = sTab[2 * ((k0 + j) * 64 + t)], sn = sTab[2 * ((k0 + j) * 64 + t) + 1];
                            re[j] += x * c; im[j] -= x * sn;
                        }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                st4(part + ((lr16 * 4 + j) * 2 + 0) * 64 + lc, re[j]);
                st4(part + ((lr16 * 4 + j) * 2 + 1) * 64 + lc, im[j]);
            }
            lds_barrier();
            for (int i = tid & 255; i < 512; i += 256) {
                const int j = i >> 7, rc = i & 127;
                if (k0 + j < cb) {
                    float acc = 0.f;
#pragma unroll
                    for (int g = 0; g < 16; ++g) acc += part[g * 512 + i];
                    specg[(k0 + j) * 128 + rc] = acc;
                }
            }
            lds_barrier();
        }
        f32x4 sb = {0, 0, 0, 0}, sg0 = sb, sb0 = sb;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = 32 * i + lr;
            const bool ok = t < L;
            const long e = (tok0 + t) * 64 + lc;
            f32x4 dx = {0, 0, 0, 0}, xh = dx;
            float rs = 0.f;
            if (ok) {
                const f32x4 xv = ld4(sXin + t * FS + lc), df = ld4(sdF + t * FS + lc);
                const f32x4 lowx = lowpass_tab(spec, t, lc, L, cb, sTab);
                const f32x4 lowg = lowpass_tab(spec + cb * 128, t, lc, L, cb, sTab);
                dx = ld4(sG + t * FS + lc) + b2 * df + lowg;
                if (e_dx_extra) dx += e_ex[i];
                sb += df * (xv - lowx);
                if (e_dz) { xh = e_xh[i]; rs = e_rs[i]; dx = dx * e_dm[i]; }
                else ast4<BF>(R8_dX, e, dx);
            }
            if (e_dz) {                                     // embedding LayerNorm backward on the finished row
                const f32x4 gg = dx * g0;
                const float m1 = group_sum<16>(gg.x + gg.y + gg.z + gg.w) * (1.0f / 64.0f);
                const float m2 = group_sum<16>(gg.x * xh.x + gg.y * xh.y + gg.z * xh.z + gg.w * xh.w) * (1.0f / 64.0f);
                if (ok) gst4(e_dz + e, rs * (gg - m1 - xh * m2));
                sg0 += dx * xh; sb0 += dx;
            }
        }
        if (e_dz) {
            const f32x4 pv[3] = {sb, sg0, sb0};
            float* const pr[3] = {sQ, sQ + 2048, sQ + 4096};
            float* const pd[3] = {R8_pbeta + (long)b * 64, KARG(FusedBwdP, e_pg) + (long)b * 64, KARG(FusedBwdP, e_pb) + (long)b * 64};
            seq_partials_64<3>(pv, pr, pd, 2.0f, R8_sqrt_beta);
        } else {
            const f32x4 pv[1] = {sb};
            float* const pr[1] = {sQ};
            float* const pd[1] = {R8_pbeta + (long)b * 64};
            seq_partials_64<1>(pv, pr, pd, 2.0f, R8_sqrt_beta);
        }
    }
    STAMP(8);
}
#undef PTYPE

static inline size_t fused_bwd_smem_bytes() { return (size_t)(9 * 64 * FS + FUSED_MAX_CB * 128 + 512) * 4; }
