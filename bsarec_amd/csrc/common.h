// Shared device helpers for the BSARec gfx950 kernels: Philox dropout stream, erf-GELU,
// wave reductions.  Wave = 64 lanes everywhere (CDNA4); nothing here is portable on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// Philox4x32-10, counter-based.  Same stream as oracle/bsarec_oracle.py::dropout_keep: element i
// of dropout site `site` at optimisation step `step` uses counter (i>>2, 0, site, step) keyed by
// the 64-bit seed and takes output word i&3; an element is kept iff word >= thresh.
// ---------------------------------------------------------------------------------------------
struct DropP {
    uint32_t thresh;            // floor(p * 2^32); 0 disables dropout (eval mode / p = 0)
    float scale;                // 1 / (1 - p)
    const uint64_t* rng;        // device: rng[0] = seed, rng[1] = step (read at run time, so a
                                // captured hipGraph replays with fresh masks)
    uint32_t site;
};

__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

// keep-mask x scale for the 4 elements of group `grp` (= element index >> 2)
__device__ __forceinline__ f32x4 drop_mult4(const DropP& d, uint64_t grp) {
    f32x4 m = {d.scale, d.scale, d.scale, d.scale};
    if (d.thresh == 0) return m;
    const __attribute__((address_space(1))) uint64_t* rng = (const __attribute__((address_space(1))) uint64_t*)d.rng;
    const uint64_t seed = rng[0];
    const uint4 w = philox4x32_10((uint32_t)grp, (uint32_t)(grp >> 32), d.site, (uint32_t)rng[1],
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
    m.x = w.x >= d.thresh ? d.scale : 0.f;
    m.y = w.y >= d.thresh ? d.scale : 0.f;
    m.z = w.z >= d.thresh ? d.scale : 0.f;
    m.w = w.w >= d.thresh ? d.scale : 0.f;
    return m;
}

// The Philox seed and step read ONCE per kernel.  drop_mult4(d, grp) reads them from the device state at every call:
// two global loads, i.e. a round trip on the critical path of a latency-bound kernel that also waits for every load
// issued before it (loads return in issue order) -- weight prefetches in flight get drained at each dropout site.
struct DropSeed { uint32_t k0, k1, step; };
__device__ __forceinline__ DropSeed drop_seed(const DropP& d) {
    DropSeed s = {0u, 0u, 0u};
    if (d.rng) {
        const __attribute__((address_space(1))) uint64_t* rng = (const __attribute__((address_space(1))) uint64_t*)d.rng;
        const uint64_t seed = rng[0];
        s.k0 = (uint32_t)seed; s.k1 = (uint32_t)(seed >> 32); s.step = (uint32_t)rng[1];
    }
    return s;
}
__device__ __forceinline__ f32x4 drop_mult4(const DropP& d, const DropSeed& sd, uint64_t grp) {
    f32x4 m = {d.scale, d.scale, d.scale, d.scale};
    if (d.thresh == 0) return m;
    const uint4 w = philox4x32_10((uint32_t)grp, (uint32_t)(grp >> 32), d.site, sd.step, sd.k0, sd.k1);
    m.x = w.x >= d.thresh ? d.scale : 0.f;
    m.y = w.y >= d.thresh ? d.scale : 0.f;
    m.z = w.z >= d.thresh ? d.scale : 0.f;
    m.w = w.w >= d.thresh ? d.scale : 0.f;
    return m;
}

// ---------------------------------------------------------------------------------------------
// erf-GELU as the reference writes it, x * 0.5 * (1 + erf(x / sqrt(2)))  (src/model/_modules.py:56), and its
// derivative.  erf is the degree-13/8 odd/even rational minimax on [-4, 4] (|error| <= 4.5e-7 absolute,
// measured against fp64 erf over 2M points in [-6, 6]): 12 FMAs + one reciprocal instead
// of libm's branchy ~60-instruction erff -- the GELU epilogues are VALU-bound at one wave per SIMD.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_erf(float x) {
    x = fminf(fmaxf(x, -4.0f), 4.0f);
    const float x2 = x * x;
    float p = -2.72614225801306e-10f;
    p = fmaf(p, x2, 2.77068142495902e-08f);
    p = fmaf(p, x2, -2.10102402082508e-06f);
    p = fmaf(p, x2, -5.69250639462346e-05f);
    p = fmaf(p, x2, -7.34990630326855e-04f);
    p = fmaf(p, x2, -2.95459980854025e-03f);
    p = fmaf(p, x2, -1.60960333262415e-02f);
    p *= x;
    float q = -1.45660718464996e-05f;
    q = fmaf(q, x2, -2.13374055278905e-04f);
    q = fmaf(q, x2, -1.68282697438203e-03f);
    q = fmaf(q, x2, -7.37332916720468e-03f);
    q = fmaf(q, x2, -1.42647390514189e-02f);
    return p * __builtin_amdgcn_rcpf(q);
}
__device__ __forceinline__ float gelu_f(float x) {
    return x * 0.5f * (1.0f + fast_erf(x * 0.70710678118654752f));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752f));
    const float pdf = __expf(-0.5f * x * x) * 0.39894228040143270f;
    return cdf + x * pdf;
}

// gelu(x) and gelu'(x) together: the fused forward saves both, so that the backward's dU epilogue is a single multiply
// instead of ~40 vector instructions per element in a stage that is bound by vector issue.
// Phi(x) = 0.5 erfc(-x / sqrt 2) through Abramowitz & Stegun 7.1.26: with z = |x| / sqrt 2, t = 1 / (1 + p z),
//   0.5 erfc(z) = 0.5 (a1 t + ... + a5 t^5) e^{-z^2}     (|error| <= 0.75e-7 absolute, RELATIVE accuracy kept in the lower tail),
// whose exponential e^{-z^2} = e^{-x^2 / 2} is the one the density needs: ~18 vector instructions for both values (the
// rational erf + a separate exp took ~26; on the fp32 path vector instructions and MFMAs share the SIMD's ALU, every one counts).
__device__ __forceinline__ void gelu_both(float x, float& g, float& gp) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = 0.5f * 1.061405429f;
    p = fmaf(p, t, 0.5f * -1.453152027f);
    p = fmaf(p, t, 0.5f * 1.421413741f);
    p = fmaf(p, t, 0.5f * -0.284496736f);
    p = fmaf(p, t, 0.5f * 0.254829592f);
    const float e = __expf(-z * z);
    const float h = p * t * e;                       // 0.5 erfc(z) = Phi(-|x|)
    const float cdf = x < 0.f ? h : 1.0f - h;
    g = x * cdf;
    gp = fmaf(x * e, 0.39894228040143270f, cdf);
}

// The reference's other hidden_act choices (src/model/_modules.py:38-59: ACT2FN), selected at run time by the plan's
// cfg.hidden_act on the generic tiled path (the fused per-sequence kernels implement the default, gelu):
//   0 gelu (erf form)   1 relu   2 swish = x sigmoid(x)   3 tanh   4 sigmoid
__device__ __forceinline__ float act_f(float x, int act) {
    if (act == 0) return gelu_f(x);
    if (act == 1) return fmaxf(x, 0.f);
    if (act == 3) return tanhf(x);
    const float sg = 1.0f / (1.0f + __expf(-x));
    return act == 2 ? x * sg : sg;
}
__device__ __forceinline__ float act_grad_f(float x, int act) {
    if (act == 0) return gelu_grad_f(x);
    if (act == 1) return x > 0.f ? 1.0f : 0.f;
    if (act == 3) { const float t = tanhf(x); return 1.0f - t * t; }
    const float sg = 1.0f / (1.0f + __expf(-x));
    return act == 2 ? sg * (1.0f + x * (1.0f - sg)) : sg * (1.0f - sg);
}

// ---------------------------------------------------------------------------------------------
// all-lanes reductions inside a group of W consecutive lanes (W = 16, 32 or 64).  The 16-lane part is four
// DPP steps (quad_perm xor 1, xor 2, row_half_mirror, row_mirror) -- no LDS traffic, unlike __shfl_xor which
// lowers to ds_bpermute_b32; 32 <-> 32 uses v_permlane32_swap; only the 16 <-> 16 step still permutes via LDS.
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// lanes l and l^32 exchange: after v_permlane32_swap on two copies of v, a = {lo, lo}, b = {hi, hi}.
// Inline asm with two read-write operands guarantees two distinct registers (the builtin called with the same
// value twice may be given ONE register, which then merely swaps its own halves); s_nop 1 covers the
// VALU-write -> permlane-read hazard that hipcc does not pad inside asm.
__device__ __forceinline__ void halves_of(float v, float& lo, float& hi) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    lo = __builtin_bit_cast(float, a);
    hi = __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float xor32_sum(float v) { float lo, hi; halves_of(v, lo, hi); return lo + hi; }
__device__ __forceinline__ float xor32_max(float v) { float lo, hi; halves_of(v, lo, hi); return fmaxf(lo, hi); }
// sum / max over the four lanes n, n + 16, n + 32, n + 48 (v_permlane16_swap: odd rows of the first operand <-> even rows of
// the second; v_permlane32_swap: upper half of the first <-> lower half of the second)
__device__ __forceinline__ void rows_of(float v, float& a_, float& b_) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    a_ = __builtin_bit_cast(float, a); b_ = __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float quad_rows_sum(float v) { float a, b; rows_of(v, a, b); return xor32_sum(a + b); }
__device__ __forceinline__ float quad_rows_max(float v) { float a, b; rows_of(v, a, b); return xor32_max(fmaxf(a, b)); }
template <int W>
__device__ __forceinline__ float group_sum(float v) {
    v += dpp_mov<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);          // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);         // row_half_mirror
    v += dpp_mov<0x140>(v);         // row_mirror
    if (W >= 32) v += __shfl_xor(v, 16, 64);
    if (W >= 64) v = xor32_sum(v);
    return v;
}
template <int W>
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    if (W >= 32) v = fmaxf(v, __shfl_xor(v, 16, 64));
    if (W >= 64) v = xor32_max(v);
    return v;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0): every global store
// still in flight (the activations saved for backward) would have to be acknowledged at each of the ~10 phase
// boundaries.  Use it only where nothing written to global memory before the barrier is read back by another wave after it;
// it also lets global LOADS stay in flight across the barrier (software prefetch).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// ---------------------------------------------------------------------------------------------
// bf16 pieces shared by the bf16 modes (fused bf16 storage, bf16 products of the generic tiled GEMM): two bf16 per dword
// (element 2j in the low half), v_cvt_pk_bf16_f32 rounding (nearest even), v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
// ---------------------------------------------------------------------------------------------
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {           // v_cvt_pk_bf16_f32 (round to nearest even)
    const bf16x2_t r = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xFFFF0000u); }
__device__ __forceinline__ u32x4 pk8(const f32x4& a, const f32x4& b) {
    return u32x4{pk_bf16(a.x, a.y), pk_bf16(a.z, a.w), pk_bf16(b.x, b.y), pk_bf16(b.z, b.w)};
}
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
