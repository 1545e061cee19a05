// Peer-to-peer gradient exchange helpers (include/bsarec_comm.h): IPC allocation plumbing and the cross-GPU barrier
// kernel.  The reduce itself lives in adam_kernel (kernels.h): every rank sums the W gradient arenas in rank order.
#pragma once
#include "../../include/bsarec_comm.h"
#include <hip/hip_runtime.h>
#include <string.h>

// One wave.  Lane p < world talks to rank p.  All flag traffic is system-scope: the flags live in another GPU's memory.
__global__ void __launch_bounds__(64)
comm_barrier_kernel(const bsarec_comm_t C) {
    __shared__ uint64_t s_epoch;
    const int lane = threadIdx.x;
    if (lane == 0) {
        const uint64_t e = *C.epoch + 1;
        *C.epoch = e;
        s_epoch = e;
    }
    __syncthreads();
    const uint64_t e = s_epoch;
    // everything this GPU wrote before the barrier (previous kernels of the stream) must be visible system-wide before
    // any peer sees the flag
    __threadfence_system();
    if (lane < C.world)
        __hip_atomic_store(C.flags[lane] + C.rank, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (lane < C.world) {
        const uint64_t* mine = C.flags[C.rank] + lane;
        const long long t0 = wall_clock64();                                   // 100 MHz constant clock
        const long long limit = (long long)(C.timeout_ms > 0 ? C.timeout_ms : 5000) * 100000LL;
        while (__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < e) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > limit) { atomicExch(C.error, 1u); break; }   // never hang the GPU on a lost peer
        }
    }
    __threadfence_system();
}

extern "C" int bsarec_comm_alloc(void** dev_ptr, size_t bytes, int uncached) {
    if (!dev_ptr || bytes == 0) return -10;
    hipError_t e = uncached ? hipExtMallocWithFlags(dev_ptr, bytes, hipDeviceMallocUncached) : hipMalloc(dev_ptr, bytes);
    if (e != hipSuccess) return (int)e;
    e = hipMemset(*dev_ptr, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return (int)e;
}
extern "C" int bsarec_comm_free(void* dev_ptr) { return dev_ptr ? (int)hipFree(dev_ptr) : -10; }

extern "C" int bsarec_comm_export(void* dev_ptr, unsigned char handle[BSAREC_IPC_HANDLE_BYTES]) {
    static_assert(sizeof(hipIpcMemHandle_t) == BSAREC_IPC_HANDLE_BYTES, "IPC handle size");
    if (!dev_ptr || !handle) return -10;
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, dev_ptr);
    if (e != hipSuccess) return (int)e;
    memcpy(handle, &h, sizeof(h));
    return 0;
}
extern "C" int bsarec_comm_import(const unsigned char handle[BSAREC_IPC_HANDLE_BYTES], void** dev_ptr) {
    if (!handle || !dev_ptr) return -10;
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    return (int)hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess);
}
extern "C" int bsarec_comm_release(void* dev_ptr) { return dev_ptr ? (int)hipIpcCloseMemHandle(dev_ptr) : -10; }

extern "C" int bsarec_comm_barrier(const bsarec_comm_t* c, void* stream) {
    if (!c || c->world < 1 || c->world > BSAREC_MAX_PEERS || c->rank < 0 || c->rank >= c->world || !c->epoch || !c->error) return -10;
    for (int p = 0; p < c->world; ++p) if (!c->flags[p]) return -10;
    hipLaunchKernelGGL(comm_barrier_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *c);
    return (int)hipGetLastError();
}
