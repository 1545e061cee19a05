// Micro-benchmark: do fp32 MFMA (one wave) and plain VALU (its SIMD partner) overlap?  512-thread workgroups, 1 per CU:
// waves 0..3 (one per SIMD) issue MFMAs, waves 4..7 (their partners) issue VALU fma; each group also alone.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define N 256
template <int MF /*0 none, 1 f32 16x16x4, 2 f32 32x32x2, 3 bf16 32x32x16*/, int VA /*VALU fma per MFMA slot in the partner wave*/>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, float a, float b) {
    const int wave = threadIdx.x >> 6;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f32x16 d0, d1;
    for (int i = 0; i < 16; ++i) { d0[i] = 0; d1[i] = 0; }
    float v0 = a, v1 = b, v2 = a + b, v3 = a - b, v4 = a * 3.f, v5 = b * 5.f, v6 = a * 7.f, v7 = b * 9.f;
    const float a1 = a * 1.25f, a2 = a * 1.5f, a3 = a * 1.75f;
    bf16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(a + i); hb[i] = (__bf16)(b + i); }
    __syncthreads();
    const long long t0 = clock64();
    __builtin_amdgcn_sched_barrier(0);
    if (wave < 4) {
        if (MF == 1) {
#pragma unroll
            for (int i = 0; i < N / 4; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b, c3, 0, 0, 0);
            }
        } else if (MF == 2) {
#pragma unroll
            for (int i = 0; i < N / 2; ++i) {
                d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, d1, 0, 0, 0);
            }
        } else if (MF == 3) {
#pragma unroll
            for (int i = 0; i < N / 2; ++i) {
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hb, ha, d1, 0, 0, 0);
            }
        }
    } else {
#pragma unroll 8
        for (int i = 0; i < N * VA / 8; ++i) {
            v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b); v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b);
            v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b); v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b);
        }
    }
    asm volatile("" :: "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(d0), "v"(d1), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7));
    __builtin_amdgcn_sched_barrier(0);
    const long long t1 = clock64();
    float s = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[5] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) cyc[threadIdx.x >> 8] = t1 - t0;
}
template <int MF, int VA> void run(const char* name, float* out, long long* cyc) {
    long long c[2] = {0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        k<MF, VA><<<256, 512>>>(out, cyc, 1.0f, 0.5f);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    }
    printf("%-44s MFMA wave %6lld cycles (%.1f per MFMA)   VALU partner %6lld cycles (%.2f per fma)\n", name, c[0], (double)c[0] / N, c[1],
           VA ? (double)c[1] / (N * VA) : 0.0);
}
int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 16);
    run<1, 0>("f32 16x16x4 alone", out, cyc);
    run<0, 4>("VALU alone (4 per slot)", out, cyc);
    run<1, 4>("f32 16x16x4 | partner 4 fma per MFMA", out, cyc);
    run<1, 8>("f32 16x16x4 | partner 8 fma per MFMA", out, cyc);
    run<2, 0>("f32 32x32x2 alone", out, cyc);
    run<2, 8>("f32 32x32x2 | partner 8 fma per MFMA", out, cyc);
    run<2, 16>("f32 32x32x2 | partner 16 fma per MFMA", out, cyc);
    run<3, 0>("bf16 32x32x16 alone", out, cyc);
    run<3, 8>("bf16 32x32x16 | partner 8 fma per MFMA", out, cyc);
    return 0;
}
