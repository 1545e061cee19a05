// Micro-benchmark: issue cost of the vector instructions the dropout generator is made of (two waves per SIMD, 1 workgroup per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 1024
template <int MODE>
__global__ void __launch_bounds__(512) k(unsigned* out, long long* cyc, unsigned a_, unsigned b_) {
    unsigned a = a_ + threadIdx.x, b = b_ * threadIdx.x + 1u;
    unsigned v0 = a, v1 = b, v2 = a ^ b, v3 = a + b, v4 = a * 3u, v5 = b * 5u, v6 = a * 7u, v7 = b * 9u;
    float f0 = (float)a, f1 = (float)b, f2 = f0 * 1.5f, f3 = f1 * 2.5f, f4 = f0 + 3.f, f5 = f1 + 5.f, f6 = f0 - 7.f, f7 = f1 - 9.f;
    __syncthreads();
    const long long t0 = clock64();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 16
    for (int i = 0; i < N / 8; ++i) {
        if (MODE == 0) { v0 = v0 * 0xD2511F53u + b; v1 = v1 * 0xD2511F53u + b; v2 = v2 * 0xD2511F53u + b; v3 = v3 * 0xD2511F53u + b; v4 = v4 * 0xD2511F53u + b; v5 = v5 * 0xD2511F53u + b; v6 = v6 * 0xD2511F53u + b; v7 = v7 * 0xD2511F53u + b; }
        if (MODE == 1) { v0 = __umulhi(v0, 0xD2511F53u) ^ b; v1 = __umulhi(v1, 0xD2511F53u) ^ b; v2 = __umulhi(v2, 0xD2511F53u) ^ b; v3 = __umulhi(v3, 0xD2511F53u) ^ b; v4 = __umulhi(v4, 0xD2511F53u) ^ b; v5 = __umulhi(v5, 0xD2511F53u) ^ b; v6 = __umulhi(v6, 0xD2511F53u) ^ b; v7 = __umulhi(v7, 0xD2511F53u) ^ b; }
        if (MODE == 2) { v0 ^= v0 >> 15; v1 ^= v1 >> 15; v2 ^= v2 >> 15; v3 ^= v3 >> 15; v4 ^= v4 >> 15; v5 ^= v5 >> 15; v6 ^= v6 >> 15; v7 ^= v7 >> 15; v0 += b; v1 += b; v2 += b; v3 += b; v4 += b; v5 += b; v6 += b; v7 += b; }
        if (MODE == 3) { f0 = fmaf(f0, f1, f2); f1 = fmaf(f1, f2, f3); f2 = fmaf(f2, f3, f4); f3 = fmaf(f3, f4, f5); f4 = fmaf(f4, f5, f6); f5 = fmaf(f5, f6, f7); f6 = fmaf(f6, f7, f0); f7 = fmaf(f7, f0, f1); }
        if (MODE == 4) { v0 = ((v0 & 0xFFFFFFu) * 0x511F53u) + b; v1 = ((v1 & 0xFFFFFFu) * 0x511F53u) + b; v2 = ((v2 & 0xFFFFFFu) * 0x511F53u) + b; v3 = ((v3 & 0xFFFFFFu) * 0x511F53u) + b; v4 = ((v4 & 0xFFFFFFu) * 0x511F53u) + b; v5 = ((v5 & 0xFFFFFFu) * 0x511F53u) + b; v6 = ((v6 & 0xFFFFFFu) * 0x511F53u) + b; v7 = ((v7 & 0xFFFFFFu) * 0x511F53u) + b; }
    }
    asm volatile("" :: "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7), "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7));
    __builtin_amdgcn_sched_barrier(0);
    const long long t1 = clock64();
    out[blockIdx.x * 512 + threadIdx.x] = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7 ^ (unsigned)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* name, unsigned* out, long long* cyc, int per) {
    long long c = 0;
    for (int rep = 0; rep < 3; ++rep) { k<MODE><<<256, 512>>>(out, cyc, 12345u, 7u); (void)hipDeviceSynchronize(); (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); }
    printf("%-40s %7lld cycles / %d ops  -> %.2f cycles per op per wave (two waves per SIMD)\n", name, c, N * per, (double)c / (N * per));
}
int main() {
    unsigned* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 8);
    run<0>("v_mad_u32 / mul_lo_u32 (+add)", out, cyc, 1);
    run<1>("v_mul_hi_u32 (+xor)", out, cyc, 1);
    run<2>("shift + xor + add (3 simple ops)", out, cyc, 3);
    run<3>("v_fma_f32", out, cyc, 1);
    run<4>("v_mad_u32_u24", out, cyc, 1);
    return 0;
}
