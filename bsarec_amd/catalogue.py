"""Catalogue-sharded BSARec training step (the C5 variant; include/bsarec_shard.h).

New functionality -- the reference is single-device (src/main.py:19).  At V = 10 M items and d = 256 a replicated item
table costs 41 GB per GPU (weights, gradient, Adam moments) and a dense 10.24 GB gradient all-reduce per step; here rank
r OWNS rows [r*rows_per, (r+1)*rows_per) of ``item_embeddings.weight`` -- their gradient and their Adam moments never
leave the GPU -- and the encoder (everything else, ~1.6 M parameters) stays a data-parallel replica.

What the reference does in `calculate_loss` (src/model/bsarec.py:30-37) maps to one step like this (per rank, B local
sequences, Bg = W*B):

  lookup           token rows are READ out of the owners' shards over xGMI (IPC-mapped hipMalloc memory) into a staging
                   table [B*L + 1, d]; the ordinary encoder plan runs over it with item_size = B*L + 1
  all-gather       h_last [Bg, d], answers [Bg], token ids [W, B*L]
  head             partial logits of ALL Bg sequences against the owned rows; per-row (max, sum exp, target logit)
  all-gather       the statistics [W, 3, Bg] -> lse, loss (identical on every rank), d loss / d logits of the owned slice
  head backward    dE of the owned rows: complete and local; partial d h_last -> all-reduce [Bg, d], keep the own rows
  encoder backward bsarec_backward_seq; the staging table's gradient holds one row per token
  barrier          (bsarec_comm_barrier) every rank's gradients are complete
  lookup gradient  owners PULL the token rows of their items out of every rank's staging gradient
  Adam             encoder: the fused Adam sums every rank's gradient arena in rank order (replicas stay bit-identical);
                   shard: local dE, local moments

The loss is the mean over the global batch: d loss / d logits carries 1 / Bg, so every gradient is already the
global-batch one and the encoder's Adam takes the plain SUM over ranks (grad_scale = 1).
"""
from __future__ import annotations

import copy
import ctypes as C

import torch

from . import _lib as L
from .dp import PeerExchange, _as_tensor
from .model import BSARecModel


class ShardedCatalogue:
    """One rank of the catalogue-sharded step.  ``args`` are the reference's (global ``item_size``); ``batch`` is the
    per-rank batch size (fixed: the staging table and the encoder plan are sized by it)."""

    def __init__(self, args, batch: int, group, device):
        import torch.distributed as dist
        self.args, self.group, self.device, self.B = args, group, torch.device(device), int(batch)
        self.rank, self.W = dist.get_rank(group), dist.get_world_size(group)
        if self.W > 8:
            raise ValueError("catalogue sharding runs inside one xGMI node (<= 8 ranks)")
        if getattr(args, "storage", None) == "bf16":
            raise ValueError("catalogue sharding is fp32 only")
        V, d, Lq = int(args.item_size), int(args.hidden_size), int(args.max_seq_length)
        self.V, self.d, self.Lq = V, d, Lq
        self.rows_per = (V + self.W - 1) // self.W
        self.lo = self.rank * self.rows_per
        self.Vs = max(0, min(self.rows_per, V - self.lo))
        self.n = self.B * Lq
        self.Bg = self.W * self.B
        self.lib = L.load()
        # encoder replica over the staging table
        enc_args = copy.copy(args)
        enc_args.item_size = self.n + 1
        enc_args.plan_options = dict(getattr(args, "plan_options", None) or {})
        self.encoder = BSARecModel(enc_args).to(self.device)
        off, self.stage_n, _ = self.encoder._slices["item_embeddings.weight"]
        assert off == 0 and self.stage_n == (self.n + 1) * d
        dist.broadcast(self.encoder._arena, src=dist.get_global_rank(group, 0), group=group)       # identical replicas
        self.encoder.set_seed(int(getattr(args, "seed", 42)), self.rank)
        # peer-to-peer plumbing: gradient arenas + the table shards in IPC-exported memory
        px = PeerExchange.create(self.encoder._numel, group, self.device)
        if px is None:
            raise RuntimeError("catalogue sharding needs the peer-to-peer mappings (hipIpc) between the ranks of the node")
        self.px = px
        self.encoder.use_grad_arenas(px.arenas)
        e_ptr, self.shard_ptrs = px.share(max(self.rows_per, 1) * d * 4)
        self.E = _as_tensor(e_ptr, self.rows_per * d, torch.float32, self.device).view(self.rows_per, d)
        gen = torch.Generator(device="cpu").manual_seed(int(getattr(args, "seed", 42)) * 1000003 + self.rank)
        self.E.copy_(torch.empty(self.rows_per, d).normal_(0.0, float(args.initializer_range), generator=gen))
        self.dE = torch.zeros_like(self.E)
        self.m, self.v = torch.zeros_like(self.E), torch.zeros_like(self.E)
        self.encoder.configure_adam(lr=float(getattr(args, "lr", 1e-3)),
                                    betas=(float(getattr(args, "adam_beta1", 0.9)), float(getattr(args, "adam_beta2", 0.999))),
                                    weight_decay=float(getattr(args, "weight_decay", 0.0)))
        # head buffers
        self.ld = (max(self.Vs, 1) + 3) // 4 * 4
        self.logits = torch.zeros(self.Bg, self.ld, dtype=torch.float32, device=self.device)
        self.stats = torch.zeros(3, self.Bg, dtype=torch.float32, device=self.device)
        self.stats_all = torch.zeros(self.W, 3, self.Bg, dtype=torch.float32, device=self.device)
        self.h_all = torch.zeros(self.Bg, d, dtype=torch.float32, device=self.device)
        self.ans_all = torch.zeros(self.Bg, dtype=torch.int64, device=self.device)
        self.ids_all = torch.zeros(self.W, self.n, dtype=torch.int64, device=self.device)
        self.local_ids = torch.zeros(self.B, Lq, dtype=torch.int64, device=self.device)
        self.dh = torch.zeros(self.Bg, d, dtype=torch.float32, device=self.device)
        self.scratch = torch.zeros(max(1, self.lib.bsarec_shard_head_bwd_scratch_floats(self.Bg, self.Vs, d)),
                                   dtype=torch.float32, device=self.device)
        self.d_out = torch.zeros(self.B, Lq, d, dtype=torch.float32, device=self.device)
        self.loss_rows = torch.zeros(self.Bg, dtype=torch.float32, device=self.device)
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._shards8 = L.PTRS8(*([int(p) for p in self.shard_ptrs] + [None] * (8 - self.W)))
        self._grads8 = L.PTRS8(*([int(p) for p in px.grad_srcs(0)] + [None] * (8 - self.W)))
        dist.barrier(group=group)

    # ---- weights ---------------------------------------------------------------------------------------------------
    def load_full_state_dict(self, sd):
        """Reference-keyed state dict with the FULL item table: this rank keeps its rows and the encoder parameters."""
        full = sd["item_embeddings.weight"].to(device=self.device, dtype=torch.float32)
        assert tuple(full.shape) == (self.V, self.d)
        self.E.zero_()
        if self.Vs:
            self.E[:self.Vs].copy_(full[self.lo:self.lo + self.Vs])
        own = self.encoder.state_dict()
        for k in own:
            if k != "item_embeddings.weight":
                own[k].copy_(sd[k].to(device=self.device, dtype=torch.float32))
        torch.cuda.synchronize(self.device)
        torch.distributed.barrier(group=self.group)

    def check_exchange(self):
        """The barrier kernel gives up after 5 s and sets a STICKY error word instead of hanging; a step that ran past a
        timed-out barrier read incomplete peer memory.  Call once per epoch / before a checkpoint (``full_state_dict``
        does): the flag is MAX-reduced so that every rank raises."""
        bad = torch.tensor([1.0 if self.px.timed_out() else 0.0], device=self.device)
        torch.distributed.all_reduce(bad, op=torch.distributed.ReduceOp.MAX, group=self.group)
        if bad.item() != 0.0:
            raise RuntimeError("catalogue-sharded step: a cross-GPU barrier timed out on some rank (sticky error word); "
                               "shards and replicas can no longer be trusted")

    def full_state_dict(self):
        """The reference's state dict (full item table gathered from the owners; checkpoints of small catalogues, tests)."""
        self.check_exchange()
        parts = [torch.empty_like(self.E) for _ in range(self.W)]
        torch.distributed.all_gather(parts, self.E.contiguous(), group=self.group)
        sd = {k: v.detach().clone() for k, v in self.encoder.state_dict().items()}
        sd["item_embeddings.weight"] = torch.cat(parts, 0)[:self.V].clone()
        return sd

    # ---- one training step -----------------------------------------------------------------------------------------
    def _adam(self, params, grads, m, v, n, srcs=None):
        a = self.encoder._adam
        s = L.Adam(params, grads, m, v, n, a["lr"], a["b1"], a["b2"], a["eps"], a["wd"], 1.0, None, 0)
        if srcs:
            s.n_grad_srcs = len(srcs)
            for i, p in enumerate(srcs):
                s.grad_srcs[i] = p
        return s

    def train_step(self, input_ids, answers) -> torch.Tensor:
        """input_ids [B, L], answers [B]: this rank's slice of the global batch.  Returns the device loss scalar (mean
        over the GLOBAL batch, the same value on every rank)."""
        import torch.distributed as dist
        lib, enc, g = self.lib, self.encoder, self.group
        B, Lq, d, W, Bg, n = self.B, self.Lq, self.d, self.W, self.Bg, self.n
        ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        ans = answers.to(device=self.device, dtype=torch.int64).contiguous()
        assert tuple(ids.shape) == (B, Lq) and tuple(ans.shape) == (B,)
        st = enc._stream()
        # peers finished the previous step: their shards are current, nobody reads my old staging gradient any more
        self.px.barrier(st)
        L.check(lib.bsarec_shard_gather_rows(ids.data_ptr(), n, C.byref(self._shards8), W, self.rows_per, self.V, d,
                                             enc._arena.data_ptr(), self.local_ids.data_ptr(), st), "bsarec_shard_gather_rows")
        enc.train()
        plan = enc._run_forward(self.local_ids, train=True, new_step=True)
        h_last = plan.view(L.BUF_LAYER_OUT, self.args.num_hidden_layers, (B, Lq, d))[:, Lq - 1, :].float().contiguous()
        dist.all_gather(list(self.h_all.view(W, B, d).unbind(0)), h_last, group=g)
        dist.all_gather(list(self.ans_all.view(W, B).unbind(0)), ans, group=g)
        dist.all_gather(list(self.ids_all.unbind(0)), ids.view(-1), group=g)
        L.check(lib.bsarec_shard_logits(self.h_all.data_ptr(), d, Bg, self.E.data_ptr(), self.Vs, d, self.logits.data_ptr(),
                                        self.ld, st), "bsarec_shard_logits")
        L.check(lib.bsarec_shard_ce_stats(self.logits.data_ptr(), self.ld, Bg, self.Vs, self.ans_all.data_ptr(), self.lo, self.V,
                                          self.stats.data_ptr(), st), "bsarec_shard_ce_stats")
        dist.all_gather(list(self.stats_all.unbind(0)), self.stats, group=g)
        L.check(lib.bsarec_shard_ce_grad(self.logits.data_ptr(), self.ld, Bg, self.Vs, self.ans_all.data_ptr(), self.lo, self.V,
                                         self.stats_all.data_ptr(), W, self.loss_rows.data_ptr(), self.loss.data_ptr(), st),
                "bsarec_shard_ce_grad")
        L.check(lib.bsarec_shard_head_bwd(self.logits.data_ptr(), self.ld, Bg, self.Vs, self.h_all.data_ptr(), d,
                                          self.E.data_ptr(), d, self.dE.data_ptr(), self.dh.data_ptr(), self.scratch.data_ptr(), st),
                "bsarec_shard_head_bwd")
        dist.all_reduce(self.dh, op=dist.ReduceOp.SUM, group=g)
        self.d_out[:, Lq - 1, :] = self.dh[self.rank * B:(self.rank + 1) * B]
        L.check(lib.bsarec_backward_seq(plan.handle, self.d_out.data_ptr(), st), "bsarec_backward_seq")
        self.px.barrier(st)                 # every rank's gradient arena (staging rows + encoder) is complete
        L.check(lib.bsarec_shard_scatter_rows(self.ids_all.data_ptr(), n, W, C.byref(self._grads8), self.lo, self.Vs, self.V, d,
                                              self.dE.data_ptr(), st), "bsarec_shard_scatter_rows")
        sn, a = self.stage_n, enc._adam
        ad = self._adam(enc._arena.data_ptr() + 4 * sn, enc._garena.data_ptr() + 4 * sn, a["m"].data_ptr() + 4 * sn,
                        a["v"].data_ptr() + 4 * sn, enc._numel - sn, [p + 4 * sn for p in self.px.grad_srcs(0)])
        L.check(lib.bsarec_adam_step(C.byref(ad), enc._state.data_ptr(), st), "bsarec_adam_step")
        if self.Vs:
            ae = self._adam(self.E.data_ptr(), self.dE.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.Vs * d)
            L.check(lib.bsarec_adam_apply(C.byref(ae), enc._state.data_ptr(), st), "bsarec_adam_apply")
        return self.loss[0]

    def train_step_graph(self, input_ids, answers) -> torch.Tensor:
        """:meth:`train_step` replayed from ONE hipGraph: the kernels AND the collectives between them (all-gathers of
        h_last / answers / ids / statistics, the all-reduce of d h_last -- torch.distributed over RCCL enqueues them on the
        capture stream, as the data-parallel step's in-graph all-reduce does) are captured once for this rank's static
        input buffers; every later call copies the batch in and launches the graph: no host work between the step's ~25
        launches.  The first call runs eagerly (plans, kernel attributes, communicators) and captures; if the backend's
        collectives cannot be captured (gloo in the tests) the step stays eager -- ``graph_captured`` says which."""
        ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        ans = answers.to(device=self.device, dtype=torch.int64).contiguous()
        if getattr(self, "_graph", None) is None and not getattr(self, "_graph_failed", False):
            self._sid, self._sans = ids.clone(), ans.clone()
            loss = self.train_step(self._sid, self._sans).clone()
            torch.cuda.synchronize(self.device)
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    gl = self.train_step(self._sid, self._sans)
                self._graph = (g, gl)
            except Exception as e:                    # capture of a collective refused: eager from now on
                self._graph_failed, self._graph_error = True, f"{type(e).__name__}: {e}"
                torch.cuda.synchronize(self.device)
            return loss
        if getattr(self, "_graph", None) is not None:
            self._sid.copy_(ids)
            self._sans.copy_(ans)
            self._graph[0].replay()
            return self._graph[1]
        return self.train_step(ids, ans)

    @property
    def graph_captured(self) -> bool:
        return getattr(self, "_graph", None) is not None

    # ---- evaluation ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def topk(self, input_ids, k: int = 20, seen=None):
        """Full-catalogue top-k of this rank's B sequences over the SHARDED table (the reference's eval step,
        src/trainers.py:118-141: scores of the last position against every item, items the user already interacted
        with set to 0 -- not -inf --, top 20): local scores of all Bg sequences against the owned rows, local top-k,
        all-gather of the W x k candidates, merge.  ``seen``: optional int64 [B, S] of item ids per sequence (padded with
        -1).  Returns (scores [B, k], item ids [B, k])."""
        import torch.distributed as dist
        lib, enc, g = self.lib, self.encoder, self.group
        B, Lq, d, W, Bg, n = self.B, self.Lq, self.d, self.W, self.Bg, self.n
        ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        assert tuple(ids.shape) == (B, Lq)
        st = enc._stream()
        self.px.barrier(st)
        L.check(lib.bsarec_shard_gather_rows(ids.data_ptr(), n, C.byref(self._shards8), W, self.rows_per, self.V, d,
                                             enc._arena.data_ptr(), self.local_ids.data_ptr(), st), "bsarec_shard_gather_rows")
        was_training = enc.training
        enc.eval()
        plan = enc._run_forward(self.local_ids, train=False, new_step=False)
        enc.train(was_training)
        h_last = plan.view(L.BUF_LAYER_OUT, self.args.num_hidden_layers, (B, Lq, d))[:, Lq - 1, :].float().contiguous()
        dist.all_gather(list(self.h_all.view(W, B, d).unbind(0)), h_last, group=g)
        L.check(lib.bsarec_shard_logits(self.h_all.data_ptr(), d, Bg, self.E.data_ptr(), self.Vs, d, self.logits.data_ptr(),
                                        self.ld, st), "bsarec_shard_logits")
        scores = self.logits[:, :self.Vs]
        if seen is not None:
            sl = seen.to(device=self.device, dtype=torch.int64).contiguous()
            S = sl.shape[1]
            seen_all = torch.empty(W, B, S, dtype=torch.int64, device=self.device)
            dist.all_gather(list(seen_all.unbind(0)), sl, group=g)
            loc = seen_all.view(Bg, S) - self.lo
            ok = (seen_all.view(Bg, S) >= 0) & (loc >= 0) & (loc < self.Vs)
            rows = torch.arange(Bg, device=self.device).view(Bg, 1).expand(Bg, S)
            scores[rows[ok], loc[ok]] = 0.0
        kk = min(k, max(self.Vs, 1))
        cand_v = torch.full((Bg, k), -float("inf"), device=self.device)
        cand_i = torch.zeros(Bg, k, dtype=torch.int64, device=self.device)
        if self.Vs:
            v, i = self._topk_rows(scores, self.logits.stride(0), kk)
            cand_v[:, :kk], cand_i[:, :kk] = v, i + self.lo
        all_v = torch.empty(W, Bg, k, device=self.device)
        all_i = torch.empty(W, Bg, k, dtype=torch.int64, device=self.device)
        dist.all_gather(list(all_v.unbind(0)), cand_v, group=g)
        dist.all_gather(list(all_i.unbind(0)), cand_i, group=g)
        mv = all_v.permute(1, 0, 2).reshape(Bg, W * k)
        mi = all_i.permute(1, 0, 2).reshape(Bg, W * k)
        top_v, sel = self._topk_rows(mv.contiguous(), W * k, k)
        top_i = torch.gather(mi, 1, sel)
        r0 = self.rank * B
        return top_v[r0:r0 + B].clone(), top_i[r0:r0 + B].clone()

    def _topk_rows(self, scores, ld, k):
        """k best columns of every row, descending (``bsarec_topk_seen`` without a seen-item mask: the HIP top-k of the
        evaluation path; equal scores go to the smaller column)."""
        rows, V = scores.shape[0], scores.shape[1]
        idx = torch.empty(rows, k, dtype=torch.int64, device=self.device)
        val = torch.empty(rows, k, dtype=torch.float32, device=self.device)
        L.check(self.lib.bsarec_topk_seen(scores.data_ptr(), ld, rows, V, None, None, None, k, idx.data_ptr(), val.data_ptr(),
                                          self.encoder._stream()), "bsarec_topk_seen")
        return val, idx

    @torch.no_grad()
    def full_sort_scores(self, batches, epoch: int = 0, k: int = 20):
        """The reference's evaluation bookkeeping (src/trainers.py:118-158 + get_full_sort_score, :70-83) over the SHARDED
        table: ``batches`` yields this rank's (input_ids [B, L], answers [B], seen [B, S] or None) per step (every rank the
        same number of steps); the top-20 of each sequence comes from :meth:`topk`, hits and DCG sums are all-reduced, so
        every rank returns the metrics of the GLOBAL evaluation set: ([HR@5, NDCG@5, HR@10, NDCG@10, HR@20, NDCG@20], str)."""
        import torch.distributed as dist
        from .trainer import ndcg_at_k, recall_at_k
        ks = (5, 10, 20)
        sums = torch.zeros(2 * len(ks) + 1, dtype=torch.float64, device=self.device)
        for ids, answers, seen in batches:
            _, top_i = self.topk(ids, k=k, seen=seen)
            hit = top_i == answers.to(device=self.device, dtype=torch.int64).view(-1, 1)
            n = hit.shape[0]
            for j, kk in enumerate(ks):
                sums[2 * j] += recall_at_k(hit, kk) * n
                sums[2 * j + 1] += ndcg_at_k(hit, kk) * n
            sums[-1] += n
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.group)
        vals = (sums[:-1] / sums[-1].clamp(min=1)).tolist()
        post_fix = {"Epoch": epoch, "HR@5": '{:.4f}'.format(vals[0]), "NDCG@5": '{:.4f}'.format(vals[1]),
                    "HR@10": '{:.4f}'.format(vals[2]), "NDCG@10": '{:.4f}'.format(vals[3]),
                    "HR@20": '{:.4f}'.format(vals[4]), "NDCG@20": '{:.4f}'.format(vals[5])}
        return vals, str(post_fix)

    def close(self):
        torch.cuda.synchronize(self.device)
        torch.distributed.barrier(group=self.group)
        self.px.close()
