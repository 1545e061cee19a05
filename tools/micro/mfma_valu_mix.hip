// Micro-benchmark: vector instructions beside MFMAs -- in the same wave, and in the SIMD partner wave with either wave older /
// prioritised.  512-thread workgroups (two waves per SIMD), 1 per CU.  Cycles = s_memtime ticks of wave 0 / wave 4.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define N 256
#define V1 v0 = fmaf(v0, a, b);
#define V2 V1 v1 = fmaf(v1, a, b);
#define V4 V2 v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b);
#define V6 V4 v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b);
#define V8 V6 v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b);
// KIND: 1 f32 16x16x4 (4 accumulators), 2 f32 32x32x2 (2 acc), 3 bf16 32x32x16 (2 acc).  NV: fma per MFMA in the SAME wave.
// PAIR: 0 only waves 0..3 work; 1 waves 0..3 MFMA, waves 4..7 VALU (PV fma per MFMA slot); 2 roles swapped (VALU in the older waves);
// PRIO: s_setprio value of the VALU wave.
template <int KIND, int NV, int PAIR, int PV, int PRIO>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, float a_, float b_) {
    const int wave = threadIdx.x >> 6;
    const float a = a_ + (float)threadIdx.x * 1e-20f, b = b_ + (float)threadIdx.x * 1e-21f;      // per-lane values: VGPR operands
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f32x16 d0, d1;
    for (int i = 0; i < 16; ++i) { d0[i] = 0; d1[i] = 0; }
    float v0 = a, v1 = b, v2 = a + b, v3 = a - b, v4 = a * 3.f, v5 = b * 5.f, v6 = a * 7.f, v7 = b * 9.f;
    const float a1 = a * 1.25f, a2 = a * 1.5f, a3 = a * 1.75f;
    bf16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(a + i); hb[i] = (__bf16)(b + i); }
    const bool mfma_wave = PAIR == 2 ? wave >= 4 : wave < 4;
    const bool valu_wave = PAIR == 1 ? wave >= 4 : (PAIR == 2 ? wave < 4 : false);
    __syncthreads();
    const long long t0 = clock64();
    __builtin_amdgcn_sched_barrier(0);
    if (mfma_wave) {
#define FILL if (NV == 1) { V1 } else if (NV == 2) { V2 } else if (NV == 4) { V4 } else if (NV == 6) { V6 } else if (NV == 8) { V8 }
        if (KIND == 1) {
#pragma unroll
            for (int i = 0; i < N / 4; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); FILL
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, c1, 0, 0, 0); FILL
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b, c2, 0, 0, 0); FILL
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b, c3, 0, 0, 0); FILL
            }
        } else if (KIND == 2) {
#pragma unroll
            for (int i = 0; i < N / 2; ++i) {
                d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0); FILL
                d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, d1, 0, 0, 0); FILL
            }
        } else if (KIND == 3) {
#pragma unroll
            for (int i = 0; i < N / 2; ++i) {
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, d0, 0, 0, 0); FILL
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hb, ha, d1, 0, 0, 0); FILL
            }
        }
    } else if (valu_wave) {
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
        if (PRIO == 3) __builtin_amdgcn_s_setprio(3);
#pragma unroll 8
        for (int i = 0; i < N * PV / 8; ++i) { V8 }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("" :: "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(d0), "v"(d1), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7));
    __builtin_amdgcn_sched_barrier(0);
    const long long t1 = clock64();
    float s = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[5] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) cyc[threadIdx.x >> 8] = t1 - t0;
}
template <int KIND, int NV, int PAIR, int PV, int PRIO> void run(const char* name, float* out, long long* cyc) {
    long long c[2] = {0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        k<KIND, NV, PAIR, PV, PRIO><<<256, 512>>>(out, cyc, 1.0f, 0.5f);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    }
    printf("%-58s wave0 %6lld  wave4 %6lld   (per MFMA slot: %.1f / %.1f)\n", name, c[0], c[1], (double)c[0] / N, (double)c[1] / N);
}
int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 16);
    run<1, 0, 0, 0, 0>("f32 16x16x4, nothing else", out, cyc);
    run<1, 2, 0, 0, 0>("f32 16x16x4 + 2 fma in the same wave", out, cyc);
    run<1, 4, 0, 0, 0>("f32 16x16x4 + 4 fma in the same wave", out, cyc);
    run<2, 4, 0, 0, 0>("f32 32x32x2 + 4 fma in the same wave", out, cyc);
    run<2, 8, 0, 0, 0>("f32 32x32x2 + 8 fma in the same wave", out, cyc);
    run<3, 0, 0, 0, 0>("bf16 32x32x16, nothing else", out, cyc);
    run<3, 2, 0, 0, 0>("bf16 32x32x16 + 2 fma in the same wave", out, cyc);
    run<3, 4, 0, 0, 0>("bf16 32x32x16 + 4 fma in the same wave", out, cyc);
    run<3, 6, 0, 0, 0>("bf16 32x32x16 + 6 fma in the same wave", out, cyc);
    run<3, 8, 0, 0, 0>("bf16 32x32x16 + 8 fma in the same wave", out, cyc);
    run<1, 0, 1, 4, 0>("f32 16x16x4 (old waves) | 4 fma/slot in young partner", out, cyc);
    run<1, 0, 1, 4, 3>("f32 16x16x4 (old waves) | 4 fma/slot, partner at prio 3", out, cyc);
    run<1, 0, 2, 4, 0>("f32 16x16x4 (YOUNG waves) | 4 fma/slot in old partner", out, cyc);
    run<2, 0, 1, 8, 3>("f32 32x32x2 (old waves) | 8 fma/slot, partner at prio 3", out, cyc);
    run<2, 0, 2, 8, 0>("f32 32x32x2 (YOUNG waves) | 8 fma/slot in old partner", out, cyc);
    run<3, 0, 1, 4, 0>("bf16 32x32x16 (old waves) | 4 fma/slot in young partner", out, cyc);
    run<3, 0, 1, 4, 3>("bf16 32x32x16 (old waves) | 4 fma/slot, partner at prio 3", out, cyc);
    run<3, 0, 2, 4, 0>("bf16 32x32x16 (YOUNG waves) | 4 fma/slot in old partner", out, cyc);
    return 0;
}
