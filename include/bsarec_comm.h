/* bsarec_comm.h -- peer-to-peer gradient exchange for the data-parallel BSARec step on one xGMI node.
 *
 * New functionality: the reference is single-device (src/main.py:19, src/trainers.py:105-107), so there is nothing
 * to match; this is the MI355X-native form of SURVEY 8(e)'s "one exchange per step".  One process per GPU.  The step's
 * gradient message is small (1.3 MB at C1 .. 5.5 MB on Yelp) and latency-bound, so instead of a ring all-reduce every
 * rank READS its peers' gradient arenas directly over xGMI (all 7 links at once, IPC-mapped hipMalloc memory) inside
 * the fused Adam kernel: one cross-GPU barrier (a 1-workgroup kernel) + one kernel per step, no collective library in
 * the data path.  Same library, same conventions as bsarec_hip.h (plain pointers, caller's stream, graph-capturable);
 * the allocation entry points below are the ONLY ones in the library that allocate device memory (IPC export needs
 * whole hipMalloc allocations, which a caching allocator does not hand out).
 */
#ifndef BSAREC_COMM_H
#define BSAREC_COMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSAREC_MAX_PEERS 8
#define BSAREC_IPC_HANDLE_BYTES 64

/* Device memory that can be exported to the other ranks of the node, zero-filled.  uncached = 0: hipMalloc (gradient
 * arenas: bulk data, made visible by kernel boundaries + the barrier's system-scope fences); uncached = 1:
 * hipExtMallocWithFlags(hipDeviceMallocUncached), fine-grained memory for the barrier flags that peers write and this
 * GPU polls.  bsarec_comm_free releases either. */
int bsarec_comm_alloc(void **dev_ptr, size_t bytes, int uncached);
int bsarec_comm_free(void *dev_ptr);
/* hipIpcGetMemHandle / hipIpcOpenMemHandle(lazy peer access) / hipIpcCloseMemHandle.  Handles are 64 opaque bytes the
 * host exchanges out of band (torch.distributed all_gather_object in the shipped host code). */
int bsarec_comm_export(void *dev_ptr, unsigned char handle[BSAREC_IPC_HANDLE_BYTES]);
int bsarec_comm_import(const unsigned char handle[BSAREC_IPC_HANDLE_BYTES], void **dev_ptr);
int bsarec_comm_release(void *dev_ptr);

/* One rank's view of the node: flags[p] is rank p's flag array (uint64[BSAREC_MAX_PEERS], in memory rank p allocated
 * with bsarec_comm_alloc; flags[rank] is the local one), `epoch` a local device uint64 the barrier kernel increments,
 * `error` a local device uint32 that is set to 1 if a wait gives up (peers more than timeout_ms late). */
typedef struct {
    int rank, world;
    uint64_t *flags[BSAREC_MAX_PEERS];
    uint64_t *epoch;
    uint32_t *error;
    int timeout_ms;           /* 0: 5000 */
} bsarec_comm_t;

/* Cross-GPU barrier as ONE 64-thread kernel on `stream`: epoch += 1; system-scope release; flags[p][rank] = epoch for
 * every p; wait until flags[rank][p] >= epoch for every p (bounded); system-scope acquire.  Everything enqueued on the
 * stream before it is complete and visible to the peers' kernels that run after THEIR barrier returns.  Never blocks
 * the host; capturable; every wave reaches its exit (the wait is bounded by timeout_ms). */
int bsarec_comm_barrier(const bsarec_comm_t *comm, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BSAREC_COMM_H */
