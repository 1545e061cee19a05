// Micro-benchmark: cost of executing straight-line code for the FIRST time on a CU (instruction fetch from L2 / memory)
// against the same code once it is in the instruction cache.  The fused per-sequence kernels run one workgroup per CU and
// every phase of them once, i.e. all of their 50-70 KB of ISA is "first time" code in every launch.
// Each kernel is KB kilobytes of independent 8-byte v_fma_f32 (8 accumulators) executed PASSES times in a loop; the
// time of every pass is recorded by wave 0 of workgroup 0.  A different kernel (the flusher) runs between launches so that
// nothing survives in the instruction cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#define PASSES 3
#define REPT_BODY "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
                  "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
template <int KB>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, float a, float b) {
    float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    long long t[PASSES + 1];
    __syncthreads();
#pragma unroll 1
    for (int p = 0; p < PASSES; ++p) {
        t[p] = clock64();
        __builtin_amdgcn_sched_barrier(0);
        // KB * 1024 / 64 repetitions of 8 instructions x 8 bytes
        asm volatile(".rept %c10\n" REPT_BODY ".endr\n"
                     : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(a), "v"(b), "n"(KB * 16));
        __builtin_amdgcn_sched_barrier(0);
    }
    t[PASSES] = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int p = 0; p < PASSES; ++p) cyc[p] = t[p + 1] - t[p];
}
template <int KB> void run(int waves, float* out, long long* cyc) {
    long long c[PASSES];
    for (int rep = 0; rep < 2; ++rep) {
        k<112><<<256, 64>>>(out, cyc, 0.5f, 0.25f);          // flusher: 112 KB of other code through every instruction cache
        (void)hipDeviceSynchronize();
        k<KB><<<256, 64 * waves>>>(out, cyc, 0.999f, 0.001f);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    }
    const int n = KB * 128;
    printf("%3d KB straight-line, %d wave(s)/CU: pass 0 %8lld cyc = %5.1f cyc/instr   pass 1 %8lld = %5.1f   pass 2 %8lld = %5.1f\n",
           KB, waves, c[0], (double)c[0] / n, c[1], (double)c[1] / n, c[2], (double)c[2] / n);
}
int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 64);
    run<4>(1, out, cyc); run<16>(1, out, cyc); run<48>(1, out, cyc); run<96>(1, out, cyc);
    run<16>(4, out, cyc); run<48>(4, out, cyc);
    run<16>(8, out, cyc); run<48>(8, out, cyc); run<96>(8, out, cyc);
    return 0;
}
