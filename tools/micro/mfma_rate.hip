// Micro-benchmark: issue cost of the fp32 MFMA shapes and of plain VALU, one wave per SIMD (4 waves per workgroup, 1 workgroup per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o tools/micro/mfma_rate && tools/micro/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define N 256
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, long long* cyc, float a, float b) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f32x16 d0, d1;
    for (int i = 0; i < 16; ++i) { d0[i] = 0; d1[i] = 0; }
    float v0 = a, v1 = b, v2 = a + b, v3 = a - b, v4 = a * 3.f, v5 = b * 5.f, v6 = a * 7.f, v7 = b * 9.f;
    const float a1 = a * 1.25f, a2 = a * 1.5f, a3 = a * 1.75f;
    __syncthreads();
    const long long t0 = clock64();
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 0) {            // 16x16x4, 4 independent accumulators
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b, c3, 0, 0, 0);
        }
    } else if (MODE == 1) {     // 16x16x4, one dependent chain
#pragma unroll
        for (int i = 0; i < N; ++i) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
    } else if (MODE == 2) {     // 32x32x2, 2 independent accumulators
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, d1, 0, 0, 0);
        }
    } else if (MODE == 3) {     // 32x32x2, one dependent chain
#pragma unroll
        for (int i = 0; i < N; ++i) d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0);
    } else if (MODE == 4) {     // VALU only: 8 independent fma chains, N*4 instructions
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b); v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b);
            v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b); v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b);
        }
    } else if (MODE == 5) {     // 16x16x4 (4 accumulators) with 4 independent VALU fma per MFMA
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b); v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, c1, 0, 0, 0); v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b); v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b, c2, 0, 0, 0); v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b); v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b, c3, 0, 0, 0); v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b); v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b);
        }
    } else if (MODE == 6) {     // 32x32x2 (2 accumulators) with 8 independent VALU fma per MFMA
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0); v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b); v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b);
            v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b); v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b);
            d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, d1, 0, 0, 0); v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b); v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b);
            v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b); v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b);
        }
    } else if (MODE == 7) {     // 16x16x4 (4 accumulators) with 8 VALU per MFMA
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
#define V8 v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b); v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b); v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b); v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); V8
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, c1, 0, 0, 0); V8
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b, c2, 0, 0, 0); V8
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b, c3, 0, 0, 0); V8
        }
    }
    asm volatile("" :: "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(d0), "v"(d1), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7));
    __builtin_amdgcn_sched_barrier(0);
    const long long t1 = clock64();
    float s = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[5] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    const char* names[] = {"16x16x4 x4 acc", "16x16x4 chain", "32x32x2 x2 acc", "32x32x2 chain", "VALU fma only (4N)", "16x16x4 + 4 VALU/MFMA", "32x32x2 + 8 VALU/MFMA", "16x16x4 + 8 VALU/MFMA"};
    for (int m = 0; m < 8; ++m) {
        long long c = 0;
        for (int rep = 0; rep < 3; ++rep) {
            switch (m) {
                case 0: k<0><<<256, 256>>>(out, cyc, 1.0f, 0.5f); break; case 1: k<1><<<256, 256>>>(out, cyc, 1.0f, 0.5f); break;
                case 2: k<2><<<256, 256>>>(out, cyc, 1.0f, 0.5f); break; case 3: k<3><<<256, 256>>>(out, cyc, 1.0f, 0.5f); break;
                case 4: k<4><<<256, 256>>>(out, cyc, 1.0f, 0.5f); break; case 5: k<5><<<256, 256>>>(out, cyc, 1.0f, 0.5f); break;
                case 6: k<6><<<256, 256>>>(out, cyc, 1.0f, 0.5f); break; case 7: k<7><<<256, 256>>>(out, cyc, 1.0f, 0.5f); break;
            }
            hipDeviceSynchronize();
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        }
        printf("%-26s %lld cycles for %d MFMA-slots -> %.1f cycles each\n", names[m], c, N, (double)c / N);
    }
    return 0;
}
