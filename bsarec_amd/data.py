"""Input side of the hot path: the reference's dataset format restated as device-resident tables.

The reference feeds the model through a Python ``Dataset`` + ``DataLoader`` (src/dataset.py:9-117,
207-221) that tops out near 44 k samples/s; here the same samples are tensorised once and live in
HBM, and a batch is an index gather on the device.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch

from .dp import shard_of_global_batch


def read_user_seqs(path: str) -> Tuple[List[List[int]], int, int]:
    """``user item item ...`` per line (src/dataset.py:184-197) -> (sequences, max_item, num_users)."""
    seqs: List[List[int]] = []
    max_item = 0
    with open(path) as fh:
        for line in fh:
            parts = line.strip().split(" ")
            items = [int(x) for x in parts[1:]]
            seqs.append(items)
            if items:
                max_item = max(max_item, max(items))
    return seqs, max_item, len(seqs)


def synth_ml1m_like(seed: int = 42, n_users: int = 6040, n_items: int = 3416) -> List[List[int]]:
    """ML-1M-shaped synthetic interactions (the real ML-1M.txt is absent from the reference mount,
    SURVEY 8d C1): lengths max(20, lognormal(4.6, 0.9)) clipped to 2314, items Zipf(1.0) over a
    fixed random permutation of 1..n_items."""
    rng = np.random.default_rng(seed)
    lens = np.clip(np.maximum(20, np.rint(rng.lognormal(4.6, 0.9, size=n_users))), 20, 2314).astype(np.int64)
    w = 1.0 / np.arange(1, n_items + 1)
    cdf = np.cumsum(w / w.sum())
    perm = rng.permutation(n_items) + 1
    out = []
    for n in lens:
        out.append(perm[np.minimum(np.searchsorted(cdf, rng.random(n)), n_items - 1)].tolist())
    return out


def write_user_seqs(path: str, seqs: List[List[int]]) -> None:
    with open(path, "w") as fh:
        for u, s in enumerate(seqs, 1):
            fh.write(str(u) + " " + " ".join(map(str, s)) + "\n")


def train_table(user_seqs: List[List[int]], L: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """All training prefixes (src/dataset.py:18-23,61-72): per user t = s[-(L+2):-2]; sample i is
    (t[:i] left-padded to L, t[i]); i = 0 is an all-padding input.  Returns (user, inputs, answers)."""
    users, ins, ans = [], [], []
    for u, s in enumerate(user_seqs):
        t = np.asarray(s[-(L + 2):-2], dtype=np.int64)
        n = len(t)
        if n == 0:
            continue
        padded = np.concatenate([np.zeros(L, dtype=np.int64), t])
        win = np.lib.stride_tricks.sliding_window_view(padded, L)[:n]       # window i = L items before t[i]
        ins.append(win)
        ans.append(t)
        users.append(np.full(n, u, dtype=np.int64))
    return np.concatenate(users), np.concatenate(ins), np.concatenate(ans)


def eval_table(user_seqs: List[List[int]], L: int, split: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """valid: (s[:-2], s[-2]); test: (s[:-1], s[-1]) (src/dataset.py:24-28,69-81)."""
    cut = 2 if split == "valid" else 1
    n = len(user_seqs)
    ins = np.zeros((n, L), dtype=np.int64)
    ans = np.zeros(n, dtype=np.int64)
    for u, s in enumerate(user_seqs):
        h = s[:-cut][-L:]
        if h:
            ins[u, L - len(h):] = h
        ans[u] = s[-cut]
    return np.arange(n, dtype=np.int64), ins, ans


def seen_csr(user_seqs: List[List[int]], split: str) -> Tuple[np.ndarray, np.ndarray]:
    """Items masked at evaluation: s[:-2] for valid, s[:-1] for test (src/dataset.py:126-160),
    as CSR (indptr, cols) with duplicates removed."""
    cut = 2 if split == "valid" else 1
    indptr = [0]
    cols = []
    for s in user_seqs:
        u = np.unique(np.asarray(s[:-cut], dtype=np.int64))
        cols.append(u)
        indptr.append(indptr[-1] + len(u))
    return np.asarray(indptr, dtype=np.int64), (np.concatenate(cols) if cols else np.zeros(0, dtype=np.int64))


class DeviceBatches:
    """Device-resident replacement of the reference's train DataLoader (RandomSampler, batch_size,
    no drop_last; src/dataset.py:209-211).  Iterating yields the reference's 5-tuples
    (user_ids, input_ids, answers, neg_answer, same_target) with tensors already on the GPU.

    Data parallel: every rank draws the same per-epoch permutation (seed + epoch) and takes the
    rank-th slice of each global batch of world*batch_size samples."""

    def __init__(self, users, inputs, answers, batch_size: int, device, shuffle: bool = True, seed: int = 42,
                 rank: int = 0, world: int = 1, drop_last: bool = False):
        self.users = torch.as_tensor(users, dtype=torch.int64, device=device)
        self.inputs = torch.as_tensor(np.ascontiguousarray(inputs), dtype=torch.int64, device=device)
        self.answers = torch.as_tensor(answers, dtype=torch.int64, device=device)
        self.batch_size, self.shuffle, self.seed = batch_size, shuffle, seed
        # data parallel: a short last global batch would give ranks unequal (or empty) shards and a
        # mis-weighted gradient mean, so it is dropped on every rank; single-GPU keeps it, as the reference does
        self.rank, self.world, self.drop_last = rank, world, (drop_last or world > 1)
        self.epoch = 0
        self.device = device
        self._empty = torch.zeros((0,), dtype=torch.int64, device=device)
        self._neg_V = None                         # enable_negatives(): catalogue size of the negative draw

    def enable_negatives(self, user_seq, item_size: int):
        """Per-sample negative items with the reference's semantics (src/dataset.py:63-67,120-124): a uniformly random
        item id in [1, item_size) that is NOT in the sample's own prefix -- ``set(items)`` of the training sample, i.e.
        its input row plus its answer (items the user touches later, or in the valid / test tail, may be drawn).
        Drawn on the device.  Needed by the sibling models with a pairwise loss (SASRec, FMLPRec); BSARec ignores
        neg_answer.  The stream is torch's device generator, not Python's ``random``: same distribution, different
        draws (no reference fixture pins negative draws)."""
        self._neg_V = int(item_size)
        self._neg_gen = torch.Generator(device=self.device)
        self._neg_gen.manual_seed(self.seed + 7919)
        return self

    def sample_negatives(self, inputs: torch.Tensor, answers: torch.Tensor) -> torch.Tensor:
        V = self._neg_V
        neg = torch.randint(1, V, answers.shape, device=self.device, generator=self._neg_gen)
        nbad = -1
        for _ in range(256):                       # rejection loop (src/dataset.py:121-123)
            bad = (inputs == neg[:, None]).any(1) | (answers == neg)
            nbad = int(bad.sum().item())
            if nbad == 0:
                break
            neg[bad] = torch.randint(1, V, (nbad,), device=self.device, generator=self._neg_gen)
        assert nbad == 0, "negative sampling did not converge (catalogue smaller than a prefix?)"
        return neg

    def enable_same_target(self):
        """Semantic augmentation of the contrastive sibling models (DuoRec): for every training sample another training
        sequence with the SAME target item (src/dataset.py:41-56,82-96), drawn per batch on the device.  The reference
        draws ``random.choice`` among the sequences of that target and re-draws while it got the sample's own items and
        the group holds anything else; here: a uniform member of the group, stepped to the next member when it is the
        sample itself and the group has more than one."""
        order = torch.argsort(self.answers, stable=True)
        sorted_ans = self.answers[order]
        uniq, counts = torch.unique_consecutive(sorted_ans, return_counts=True)
        V = int(self.answers.max().item()) + 1
        self._st_order = order
        self._st_start = torch.zeros(V, dtype=torch.int64, device=self.device)
        self._st_count = torch.zeros(V, dtype=torch.int64, device=self.device)
        self._st_start[uniq] = torch.cumsum(counts, 0) - counts
        self._st_count[uniq] = counts
        self._st_gen = torch.Generator(device=self.device)
        self._st_gen.manual_seed(self.seed + 104729)
        return self

    def sample_same_target(self, idx: torch.Tensor) -> torch.Tensor:
        ans = self.answers[idx]
        start, cnt = self._st_start[ans], self._st_count[ans]
        r = (torch.rand(idx.shape, device=self.device, generator=self._st_gen) * cnt).long().clamp_(max=cnt.max() - 1)
        r = torch.minimum(r, cnt - 1)
        pick = self._st_order[start + r]
        own = (pick == idx) & (cnt > 1)
        pick = torch.where(own, self._st_order[start + (r + 1) % cnt], pick)
        return self.inputs[pick]

    def __len__(self):
        g = self.batch_size * self.world
        n = self.answers.shape[0]
        return n // g if self.drop_last else (n + g - 1) // g

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def local_permutation(self) -> torch.Tensor:
        """This epoch's sample order as seen by this rank: consecutive runs of batch_size entries are the
        rank's slice of consecutive global batches (all of the epoch's samples when world == 1)."""
        n = self.answers.shape[0]
        if self.shuffle:
            gen = torch.Generator(device="cpu")
            gen.manual_seed(self.seed + self.epoch)
            perm = torch.randperm(n, generator=gen).to(self.device)
        else:
            perm = torch.arange(n, device=self.device)
        if self.world > 1:
            g = self.batch_size * self.world
            nb = n // g
            perm = perm[:nb * g].view(nb, self.world, self.batch_size)[:, self.rank, :].reshape(-1)
        return perm.contiguous()

    def __iter__(self):
        n = self.answers.shape[0]
        if self.shuffle:
            gen = torch.Generator(device="cpu")
            gen.manual_seed(self.seed + self.epoch)
            perm = torch.randperm(n, generator=gen).to(self.device)
        else:
            perm = torch.arange(n, device=self.device)
        g = self.batch_size * self.world
        for i in range(len(self)):
            idx = perm[i * g:(i + 1) * g]
            if self.world > 1:
                idx = shard_of_global_batch(idx, self.batch_size, self.rank, self.world)
            users, ins, ans = self.users[idx], self.inputs[idx], self.answers[idx]
            neg = self.sample_negatives(ins, ans) if self._neg_V is not None else self._empty
            same = self.sample_same_target(idx) if getattr(self, "_st_order", None) is not None else self._empty.view(0)
            yield (users, ins, ans, neg, same)
        self.epoch += 1
