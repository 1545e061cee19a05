// Microbenchmark: cost of a dependent kernel boundary on this box (trivial kernels, eager vs graph).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k_one(unsigned long long* s) { s[1] += 1; }
__global__ void __launch_bounds__(256) k_wide(float* p, int n) {
    extern __shared__ float sm[];
    int i = blockIdx.x * 256 + threadIdx.x;
    sm[threadIdx.x] = (float)i;
    __syncthreads();
    if (i < n) p[i] = sm[255 - threadIdx.x];
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    unsigned long long* st; float* buf;
    CK(hipMalloc(&st, 64)); CK(hipMemset(st, 0, 64)); CK(hipMalloc(&buf, 64 << 20));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int N = 2000;
    for (int variant = 0; variant < 3; ++variant) {
        auto launch = [&]() {
            if (variant == 0) hipLaunchKernelGGL(k_one, dim3(1), dim3(1), 0, s, st);
            else if (variant == 1) hipLaunchKernelGGL(k_wide, dim3(200), dim3(256), 37 * 1024, s, buf, 200 * 256);
            else hipLaunchKernelGGL(k_wide, dim3(12800), dim3(256), 1024, s, buf, 12800 * 256);
        };
        for (int i = 0; i < 10; ++i) launch();
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(a, s));
        for (int i = 0; i < N; ++i) launch();
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("variant %d eager: %.3f us per kernel\n", variant, ms * 1e3 / N);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 100; ++i) launch();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(a, s));
        for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
        printf("variant %d graph: %.3f us per kernel\n", variant, ms * 1e3 / 2000);
    }
    return 0;
}
