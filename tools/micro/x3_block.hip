// Micro-benchmark: one 16-deep k-block of the "x3" product (two three-way bf16 splits + six bf16 MFMAs) against its fp32
// equivalent (8 x v_mfma_f32_32x32x2_f32), two waves per SIMD, operands in registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../bsarec_amd/csrc/fused_layer.h"
#define NB 64
template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, const float* in) {
    f32x4 a0 = gld4(in + 8 * threadIdx.x), a1 = gld4(in + 8 * threadIdx.x + 4), b0 = gld4(in + 4096 + 8 * threadIdx.x), b1 = gld4(in + 4100 + 8 * threadIdx.x);
    f32x16 acc, acc2;
    for (int i = 0; i < 16; ++i) { acc[i] = 0; acc2[i] = 0; }
    __syncthreads();
    const long long t0 = clock64();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 4
    for (int i = 0; i < NB; ++i) {
        if (MODE == 0) {            // fp32: 8 MFMAs
#pragma unroll
            for (int s = 0; s < 4; ++s) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], acc, 0, 0, 0); }
        } else if (MODE == 1) {     // x3, one accumulator
            acc = mfma_x3(split3(a0, a1), split3(b0, b1), acc);
        } else if (MODE == 2) {     // x3, two tiles interleaved (shared A split)
            const Split3 A = split3(a0, a1), B = split3(b0, b1), B2 = split3(b1, b0);
            acc = mfma_x3(A, B, acc); acc2 = mfma_x3(A, B2, acc2);
        } else if (MODE == 3) {     // splits only
            const Split3 A = split3(a0, a1), B = split3(b0, b1);
            const u32x4 x = A.h ^ A.m ^ A.l ^ B.h ^ B.m ^ B.l;
            acc[0] += __builtin_bit_cast(float, x.x ^ x.y ^ x.z ^ x.w) * 1e-30f;
        } else if (MODE == 4) {     // bf16 MFMAs only (6, dependent)
            const u32x4 x = __builtin_bit_cast(u32x4, a0), y = __builtin_bit_cast(u32x4, b0);
#pragma unroll
            for (int s = 0; s < 6; ++s) acc = mfma_bf16(x, y, acc);
        }
        { const float e0 = acc[0] * 1e-30f, e1 = acc[1] * 1e-30f; a0 = a0 + e0; a1 = a1 + e1; b0 = b0 + e1; b1 = b1 + e0; }      // loop-carried on EVERY element: no hoisting of the splits (16 extra adds per k-block in every mode)
    }
    asm volatile("" :: "v"(acc), "v"(acc2));
    __builtin_amdgcn_sched_barrier(0);
    const long long t1 = clock64();
    out[blockIdx.x * 512 + threadIdx.x] = acc[3] + acc2[5];
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* name, float* out, long long* cyc, const float* in, int per) {
    long long c = 0;
    for (int rep = 0; rep < 3; ++rep) { k<MODE><<<256, 512>>>(out, cyc, in); (void)hipDeviceSynchronize(); (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); }
    printf("%-52s %7lld cycles / %d k-blocks of 16 -> %.1f cycles per k-block per wave (two waves per SIMD)\n", name, c, NB * per, (double)c / (NB * per));
}
int main() {
    float *out, *in; long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 8); (void)hipMalloc(&in, 65536);
    float h[16384]; for (int i = 0; i < 16384; ++i) h[i] = 0.37f + 0.001f * (i % 977);
    (void)hipMemcpy(in, h, 65536, hipMemcpyHostToDevice);
    run<0>("fp32: 8 x v_mfma_f32_32x32x2_f32", out, cyc, in, 1);
    run<1>("x3: 2 splits + 6 bf16 MFMAs, one tile", out, cyc, in, 1);
    run<2>("x3: 3 splits + 12 bf16 MFMAs, two tiles", out, cyc, in, 2);
    run<3>("the two splits alone", out, cyc, in, 1);
    run<4>("6 dependent bf16 MFMAs alone", out, cyc, in, 1);
    return 0;
}
