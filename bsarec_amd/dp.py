"""Data-parallel pieces of the training path (new functionality: the reference is single-device,
src/main.py:19).  One process per GPU; user-sequence minibatches shard across ranks; every rank holds a full replica
and runs the same fused Adam on the summed gradient / world.  Three forms of the one exchange per step:

``rccl``           ONE summing all-reduce of the flat fp32 gradient arena after the backward (RCCL over xGMI when the
                   backend is "nccl"), captured inside the step's hipGraph.
``rccl_bucketed``  two buckets: the dense item-table gradient dE = dlogits^T . h_last (V*d floats, 68 % of the bytes at
                   C1, complete right after the logits backward) is all-reduced on a side stream UNDER the encoder
                   backward; the lookup-path rows go to their own [V, d] buffer and travel with the encoder gradients
                   in the second bucket after the backward.
``p2p``            no collective library in the data path: gradient arenas live in IPC-exported hipMalloc memory; after
                   ONE cross-GPU barrier kernel every rank's fused Adam reads all W arenas directly over xGMI (all
                   links at once) and sums them in rank order (bsarec_comm.h).  Two arenas alternate by step parity, so
                   the barrier of step k+1 is also the guarantee that nobody still reads the arena of step k-1.

Loss semantics: the reference's loss is the mean CE over the batch.  With equal shards of size B the
mean over the global batch of W*B samples equals the mean of the W local means, so each rank
back-propagates its local mean and the summed gradient is scaled by 1/W.  A short last global batch
would break the equal-shard premise and is dropped on every rank (DeviceBatches, world > 1).
"""
from __future__ import annotations

import ctypes as C

import torch


def shard_of_global_batch(index: torch.Tensor, batch_size: int, rank: int, world: int) -> torch.Tensor:
    """Rank's slice of one global batch of world*batch_size sample indices (same permutation on every rank)."""
    return index[rank * batch_size:(rank + 1) * batch_size]


def allreduce_sum_(flat_grads: torch.Tensor, group=None, force: bool = False) -> float:
    """In-place summing all-reduce of the flat gradient arena; returns the scale (1/world) that the
    fused Adam applies to the sum.  ``force`` issues the collective even in a 1-rank group (tests, bench --dp:
    exercises the RCCL launch and its graph capture on a single GPU)."""
    world = torch.distributed.get_world_size(group)
    if world > 1 or force:
        torch.distributed.all_reduce(flat_grads, op=torch.distributed.ReduceOp.SUM, group=group)
    return 1.0 / world


class _DevMem:
    """A raw device pointer as a torch tensor (``__cuda_array_interface__``); keeps ``owner`` alive."""

    def __init__(self, ptr: int, nbytes: int, typestr: str, itemsize: int, owner=None):
        self.__cuda_array_interface__ = {"shape": (nbytes // itemsize,), "typestr": typestr, "data": (ptr, False),
                                         "version": 2, "strides": None}
        self.owner = owner


def _as_tensor(ptr: int, numel: int, dtype: torch.dtype, device) -> torch.Tensor:
    ts, isz = {torch.float32: ("<f4", 4), torch.int64: ("<i8", 8), torch.int32: ("<i4", 4)}[dtype]
    return torch.as_tensor(_DevMem(ptr, numel * isz, ts, isz), device=device)


class PeerExchange:
    """The ``p2p`` exchange: two gradient arenas (step parity) + barrier flags of every rank, IPC-mapped into this
    process.  Two allocations per rank: [arena 0 | arena 1] (hipMalloc) and [flags u64[8] | epoch u64 | error u32]
    (uncached, fine-grained: peers write the flags, this GPU polls them)."""

    @classmethod
    def create(cls, numel: int, group, device, logger=None, timeout_ms: int = 5000):
        """Build the exchange on every rank and prove it (self_test); returns None on EVERY rank if any rank failed any
        step -- the collective calls below are reached by all ranks whatever happened locally, so a failure cannot strand
        the others."""
        px, err = None, None
        try:
            px = cls(numel, group, device, timeout_ms)
        except Exception as e:                                 # noqa: BLE001 -- any local failure means "fall back"
            err = f"{type(e).__name__}: {e}"
        ok = torch.tensor([1.0 if (px is not None and px.ok) else 0.0], device=device)
        torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN, group=group)
        if ok.item() == 1.0 and px.self_test():
            return px
        if logger is not None:
            logger.info(f"peer-to-peer gradient exchange unavailable ({err or 'a peer failed / self-test failed'}); using RCCL")
        return None

    def __init__(self, numel: int, group, device, timeout_ms: int = 5000):
        from . import _lib as L
        self.lib = lib = L.load()
        self.L = L
        self.group, self.device = group, device
        self.rank = torch.distributed.get_rank(group)
        self.world = torch.distributed.get_world_size(group)
        self.numel = numel
        self.arena_bytes = (numel * 4 + 255) // 256 * 256
        self.ok, self.base, self.fbase, raw = self.world <= 8, None, None, None     # one xGMI node
        self._shared = []
        if self.ok:
            base, fbase = C.c_void_p(), C.c_void_p()
            h1, h2 = C.create_string_buffer(64), C.create_string_buffer(64)
            self.ok = lib.bsarec_comm_alloc(C.byref(base), 2 * self.arena_bytes, 0) == 0 and \
                lib.bsarec_comm_alloc(C.byref(fbase), 256, 1) == 0
            if self.ok:
                self.base, self.fbase = base.value, fbase.value
                self.ok = lib.bsarec_comm_export(self.base, h1) == 0 and lib.bsarec_comm_export(self.fbase, h2) == 0
                raw = (bytes(h1.raw), bytes(h2.raw))
        handles = [None] * self.world
        torch.distributed.all_gather_object(handles, raw if self.ok else None, group=group)     # reached by every rank
        self.peer_base, self.peer_flags = [], []
        self.ok = self.ok and all(h is not None for h in handles)
        for r in range(self.world):
            if r == self.rank or not self.ok:
                self.peer_base.append(self.base)
                self.peer_flags.append(self.fbase)
            else:
                p, q = C.c_void_p(), C.c_void_p()
                if lib.bsarec_comm_import(handles[r][0], C.byref(p)) != 0 or lib.bsarec_comm_import(handles[r][1], C.byref(q)) != 0:
                    self.ok = False
                self.peer_base.append(p.value)
                self.peer_flags.append(q.value)
        if not self.ok:
            return
        self.comm = L.Comm()
        self.comm.rank, self.comm.world, self.comm.timeout_ms = self.rank, self.world, int(timeout_ms)
        for r in range(self.world):
            self.comm.flags[r] = self.peer_flags[r]
        self.comm.epoch = self.fbase + 64
        self.comm.error = self.fbase + 72
        self.arenas = [_as_tensor(self.base + q * self.arena_bytes, numel, torch.float32, device) for q in range(2)]
        self._err = _as_tensor(self.fbase + 72, 1, torch.int32, device)
        # (create() follows with a collective that every rank reaches: nobody uses a mapping before all ranks have theirs)

    def grad_srcs(self, parity: int):
        """Every rank's arena of this parity, in rank order (the local one at index ``rank``)."""
        return [self.peer_base[r] + parity * self.arena_bytes for r in range(self.world)]

    def barrier(self, stream: int):
        self.L.check(self.lib.bsarec_comm_barrier(C.byref(self.comm), stream), "bsarec_comm_barrier")

    def timed_out(self) -> bool:
        return bool(self._err.item() != 0)

    def self_test(self) -> bool:
        """Every rank fills arena 1 with a rank-dependent pattern, one barrier, then reads every peer's arena through the
        mapping: proves the IPC mappings and the barrier on the hardware it runs on before training depends on them."""
        n = min(self.numel, 4096)
        self.arenas[1][:n] = float(self.rank + 1)
        st = torch.cuda.current_stream(self.device).cuda_stream
        self.barrier(st)
        ok = not self.timed_out()
        for r in range(self.world):
            t = _as_tensor(self.peer_base[r] + self.arena_bytes, n, torch.float32, self.device)
            ok = ok and bool((t == float(r + 1)).all().item())
        self.barrier(st)                                       # nobody clears while a peer still reads
        self.arenas[1][:n] = 0.0
        flag = torch.tensor([1.0 if ok else 0.0], device=self.device)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN, group=self.group)
        return bool(flag.item() == 1.0)

    def share(self, nbytes: int):
        """One more IPC-exported allocation of ``nbytes`` per rank (the catalogue shards of bsarec_amd/catalogue.py):
        returns (local pointer, [every rank's pointer as mapped into this process, rank order]).  Collective: every rank
        calls it with the same size; raises on every rank if any rank failed."""
        C_ = C
        base, h = C_.c_void_p(), C_.create_string_buffer(64)
        ok = self.lib.bsarec_comm_alloc(C_.byref(base), int(nbytes), 0) == 0 and \
            self.lib.bsarec_comm_export(base.value, h) == 0
        handles = [None] * self.world
        torch.distributed.all_gather_object(handles, bytes(h.raw) if ok else None, group=self.group)
        ok = ok and all(x is not None for x in handles)
        ptrs = []
        for r in range(self.world):
            if r == self.rank or not ok:
                ptrs.append(base.value)
            else:
                q = C_.c_void_p()
                ok = self.lib.bsarec_comm_import(handles[r], C_.byref(q)) == 0 and ok
                ptrs.append(q.value)
        flag = torch.tensor([1.0 if ok else 0.0], device=self.device)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN, group=self.group)
        if flag.item() != 1.0:
            raise RuntimeError("peer-to-peer shared allocation failed on some rank (hipIpc export / import)")
        self._shared.append((base.value, [q for r, q in enumerate(ptrs) if r != self.rank]))
        return base.value, ptrs

    def close(self):
        if getattr(self, "base", None) is None:
            return
        try:
            torch.cuda.synchronize(self.device)
            for own, peers in self._shared:
                for q in peers:
                    self.lib.bsarec_comm_release(q)
                self.lib.bsarec_comm_free(own)
            self._shared = []
            for r in range(self.world):
                if r != self.rank:
                    self.lib.bsarec_comm_release(self.peer_base[r])
                    self.lib.bsarec_comm_release(self.peer_flags[r])
            self.lib.bsarec_comm_free(self.base)
            self.lib.bsarec_comm_free(self.fbase)
        finally:
            self.base = None

    # (no __del__: tensors viewing this memory may outlive the object; the process exit releases everything)
