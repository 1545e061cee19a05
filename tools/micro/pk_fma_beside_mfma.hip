// Micro-reproducer attempt for the round-3 fault: the FrequencyLayer's pruned DFT of waves 0..3 (float4 arithmetic,
// which the back end turns into v_pk_fma_f32 with op_sel unless packed fp32 operations are disabled) while the SIMD partner
// waves 4..7 run the x3 product sequence (LDS fragment reads, three-way bf16 splits, six v_mfma_f32_32x32x16_bf16 per k-block).
// Inputs are constant, so every repetition must reproduce the spectrum of repetition 0 bit for bit; the kernel counts the
// elements that do not.   Build twice:  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize [-Xclang -target-feature -Xclang
// -packed-fp32-ops] tools/micro/pk_fma_beside_mfma.hip -o tools/micro/pk_fma_beside_mfma[_nopk]
#include "../../bsarec_amd/csrc/fused_layer.h"
#include <cstdio>
#include <vector>

template <bool PARTNER>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
dft_beside_mfma(const float* __restrict__ xin, const float* __restrict__ tw, const float* __restrict__ w, int L, int cb, int reps,
                unsigned* __restrict__ mismatches, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int TS = 64 * FS;
    float* sX = sm;
    float* part = sm + TS;                 // [16][4][2][64]
    float* sTab = sm + 3 * TS;             // [8][64][2]
    float* sSpec = sTab + 8 * 128;         // [8][2][64]
    float* sRef = sSpec + 8 * 128;         // spectrum of repetition 0
    float* sQ = sRef + 8 * 128;            // partner's output tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = wave >> 2;
    const int l31 = lane & 31, half = lane >> 5, wm = (wave >> 1) & 1, wn = wave & 1;
    for (int i = tid; i < 64 * 16; i += 512) {
        const int r = i >> 4, c4 = (i & 15) << 2;
        f32x4 v = gld4(xin + ((long)blockIdx.x * 64 + r) * 64 + c4);
        if (r >= L) v = f32x4{0, 0, 0, 0};
        st4(sX + r * FS + c4, v);
    }
    build_twiddle_table(tw, L, cb, sTab);
    WFrag<2, 64> wA;
    const int col = wn * 32 + l31, KH = 8 * half;
    load_w<2, 64>(w, (long)col * 64 + KH, wA);
    lds_barrier();
    const int lr = (tid & 255) >> 4, lc = (tid & 15) << 2;
    unsigned bad = 0;
    float keep = 0.f;
    for (int rep = 0; rep < reps; ++rep) {
        if (grp == 0) {
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
                        if (j < cb) {
                            const float c = sTab[2 * (j * 64 + t)], sn = sTab[2 * (j * 64 + t) + 1];
                            re[j] += x * c; im[j] -= x * sn;
                        }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                st4(part + ((lr * 4 + j) * 2 + 0) * 64 + lc, re[j]);
                st4(part + ((lr * 4 + j) * 2 + 1) * 64 + lc, im[j]);
            }
        } else if (PARTNER) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            mma_w<2, 64>(sX + (wm * 32 + l31) * FS + KH, wA, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) sQ[(wm * 32 + rho(r) + 4 * half) * FS + col] = acc[r];
            keep += acc[rep & 15];
        }
        lds_barrier();
        if (grp == 0) {
            for (int i = tid; i < 512; i += 256) {
                const int j = i >> 7, rc = i & 127;
                if (j < cb) {
                    float a = 0.f;
#pragma unroll
                    for (int g = 0; g < 16; ++g) a += part[g * 512 + i];
                    if (rep == 0) sRef[j * 128 + rc] = a;
                    else if (__float_as_uint(a) != __float_as_uint(sRef[j * 128 + rc])) ++bad;
                    sSpec[j * 128 + rc] = a;
                }
            }
        } else if (PARTNER) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            mma_w<2, 64>(sX + (wm * 32 + l31) * FS + KH, wA, acc);
            keep += acc[(rep + 3) & 15];
        }
        lds_barrier();
    }
    if (bad) atomicAdd(mismatches, bad);
    if (keep == 12345.678f) sink[tid] = keep;
}

int main(int argc, char** argv) {
    const int nblk = 1024, L = 50, cb = 3, reps = argc > 1 ? atoi(argv[1]) : 200;
    std::vector<float> x((size_t)nblk * 64 * 64), tw(2 * L), w(64 * 64);
    unsigned seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (auto& v : x) v = rnd();
    for (auto& v : w) v = rnd() * 0.1f;
    for (int j = 0; j < L; ++j) { tw[2 * j] = (float)cos(2 * M_PI * j / L); tw[2 * j + 1] = (float)sin(2 * M_PI * j / L); }
    float *dx, *dtw, *dw, *sink; unsigned* dm;
    (void)hipMalloc(&dx, x.size() * 4); (void)hipMalloc(&dtw, tw.size() * 4); (void)hipMalloc(&dw, w.size() * 4);
    (void)hipMalloc(&sink, 512 * 4); (void)hipMalloc(&dm, 4);
    (void)hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dtw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    const size_t smem = (size_t)(4 * 64 * FS + 3 * 8 * 128 + 64 * FS) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dft_beside_mfma<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dft_beside_mfma<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    for (int partner = 0; partner < 2; ++partner) {
        unsigned total = 0;
        for (int it = 0; it < 5; ++it) {
            (void)hipMemset(dm, 0, 4);
            if (partner) dft_beside_mfma<true><<<nblk, 512, smem>>>(dx, dtw, dw, L, cb, reps, dm, sink);
            else dft_beside_mfma<false><<<nblk, 512, smem>>>(dx, dtw, dw, L, cb, reps, dm, sink);
            hipError_t e = hipDeviceSynchronize();
            unsigned m = 0;
            (void)hipMemcpy(&m, dm, 4, hipMemcpyDeviceToHost);
            if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
            total += m;
        }
        printf("partner waves %s: %u spectrum elements differed from repetition 0 (5 launches x %d workgroups x %d repetitions x 384 elements)\n",
               partner ? "running x3 products (bf16 MFMA + splits)" : "idle", total, nblk, reps);
    }
    return 0;
}
