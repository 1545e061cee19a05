"""Host-side mirror of the reference model API over the gfx950 C ABI.

``BSARecModel(args)`` keeps the reference's constructor, attribute and state_dict contract
(src/model/bsarec.py:7-37, src/model/_abstract_model.py:6-79; the 4 + 19 N state_dict keys of
SURVEY 8b) so the reference's ``main.py`` / ``Trainer`` / checkpoints drop in, while every FLOP runs
in ``libbsarec_hip.so``.  PyTorch is only the allocator, the stream and the nn.Module shell:

* all parameters are views of ONE flat fp32 arena, their gradients views of a second one, so the
  optimiser is a single fused kernel and the data-parallel exchange a single all-reduce;
* activations / scratch live in one workspace tensor carved by the library's launch plan;
* there is no fallback: constructing works on CPU (state_dict plumbing), running needs the GPU.
"""
from __future__ import annotations

import ctypes as C
import math
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L


def param_shapes(args) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict keys and shapes in the reference's registration order."""
    d, Lq, V = args.hidden_size, args.max_seq_length, args.item_size
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["item_embeddings.weight"] = (V, d)
    s["position_embeddings.weight"] = (Lq, d)
    s["LayerNorm.weight"] = (d,)
    s["LayerNorm.bias"] = (d,)
    for l in range(args.num_hidden_layers):
        p = f"item_encoder.blocks.{l}."
        s[p + "layer.filter_layer.sqrt_beta"] = (1, 1, d)
        s[p + "layer.filter_layer.LayerNorm.weight"] = (d,)
        s[p + "layer.filter_layer.LayerNorm.bias"] = (d,)
        for nm in ("query", "key", "value", "dense"):
            s[p + f"layer.attention_layer.{nm}.weight"] = (d, d)
            s[p + f"layer.attention_layer.{nm}.bias"] = (d,)
        s[p + "layer.attention_layer.LayerNorm.weight"] = (d,)
        s[p + "layer.attention_layer.LayerNorm.bias"] = (d,)
        s[p + "feed_forward.dense_1.weight"] = (4 * d, d)
        s[p + "feed_forward.dense_1.bias"] = (4 * d,)
        s[p + "feed_forward.dense_2.weight"] = (d, 4 * d)
        s[p + "feed_forward.dense_2.bias"] = (d,)
        s[p + "feed_forward.LayerNorm.weight"] = (d,)
        s[p + "feed_forward.LayerNorm.bias"] = (d,)
    if getattr(args, "filter_kind", 0) == 1:           # sibling model FMLPRec: the learnable complex filter, appended
        for l in range(args.num_hidden_layers):
            s[f"item_encoder.blocks.{l}.layer.filter_layer.complex_weight"] = (1, Lq // 2 + 1, d, 2)
    return s


class _Bag(nn.Module):
    """Attribute container so that parameter paths match the reference's module tree."""


def _twiddle(Lq: int) -> torch.Tensor:
    j = np.arange(Lq, dtype=np.float64)
    tw = np.stack([np.cos(2.0 * np.pi * j / Lq), np.sin(2.0 * np.pi * j / Lq)], axis=1)
    return torch.from_numpy(tw.astype(np.float32).reshape(-1).copy())


class _Plan:
    """One launch plan (fixed batch size) + the workspace tensor it carves."""

    def __init__(self, model: "BSARecModel", batch: int, garena=None, own_state: bool = False):
        lib = L.load()
        a = model.args
        self.cfg = L.Config(batch, a.max_seq_length, a.hidden_size, a.num_attention_heads, a.num_hidden_layers,
                            a.item_size, model.cutoff_bins, float(a.alpha), 1e-12, float(a.hidden_dropout_prob),
                            float(a.attention_probs_dropout_prob), int(getattr(a, "filter_kind", 0)))
        self.cfg.hidden_act = L.HIDDEN_ACTS[getattr(a, "hidden_act", "gelu")]
        opts = L.default_options()
        opts.update(model.options)                      # per-model overrides win over host defaults / environment
        for k in L.OPTION_FIELDS:
            setattr(self.cfg, k, int(opts.get(k, 0)))
        self.options = {k: int(getattr(self.cfg, k)) for k in L.OPTION_FIELDS}
        nbytes = lib.bsarec_workspace_bytes(C.byref(self.cfg))
        if nbytes == 0:
            raise ValueError("configuration not supported by libbsarec_hip (see include/bsarec_hip.h limits: "
                             "L <= 256, hidden <= 256 and % 4 == 0, head size % 4 == 0, cutoff_bins*hidden <= 8192)")
        dev = model._arena.device
        self.ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
        off = (-self.ws.data_ptr()) % 256
        self.ws = self.ws[off:off + nbytes]
        self.ws.zero_()
        self.batch = batch
        self.handle = C.c_void_p()
        # storage = bf16: bf16 TENSORS (and a bf16 shadow of the weights) exist at the fused shape only; elsewhere the generic
        # kernels keep fp32 tensors and multiply in bf16 (include/bsarec_hip.h, bsarec_config_t.storage)
        self.bf16 = bool(self.cfg.storage) and lib.bsarec_config_is_fused(C.byref(self.cfg)) == 1
        self.garena = model._garena if garena is None else garena
        pt, gt = model._tensor_struct(model._arena), model._tensor_struct(self.garena)
        st = model._tensor_struct(model._shadow_arena()) if self.bf16 else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        # own_state: a snapshot of the model's step state (dropout seed / step) that stays put while OTHER forwards run --
        # a forward retained for autograd must regenerate ITS masks in its backward
        self.state = torch.zeros_like(model._state) if own_state else model._state
        L.check(lib.bsarec_plan_create(C.byref(self.handle), C.byref(self.cfg), C.byref(pt), C.byref(gt),
                                       C.byref(st) if st is not None else None,
                                       self.ws.data_ptr(), nbytes, self.state.data_ptr(),
                                       model._twiddle.data_ptr(), stream), "bsarec_plan_create")
        self.lib = lib

    def view(self, buf: int, layer: int, shape) -> torch.Tensor:
        """A workspace buffer as a tensor: fp32, or bfloat16 where the plan stores it so (cfg.storage = 1)."""
        off = self.lib.bsarec_buffer_offset(self.handle, buf, layer)
        if off < 0:
            raise IndexError("no such buffer")
        if self.lib.bsarec_buffer_is_bf16(self.handle, buf, layer):
            n = int(np.prod(shape)) * 2
            return self.ws[off:off + n].view(torch.bfloat16).view(*shape)
        n = int(np.prod(shape)) * 4
        return self.ws[off:off + n].view(torch.float32).view(*shape)

    def __del__(self):
        try:
            if self.handle:
                self.lib.bsarec_plan_destroy(self.handle)
        except Exception:
            pass


class _SlotToken:
    """Holds one retained-forward plan slot; released by the backward, or when autograd drops the graph."""

    def __init__(self, model, key):
        self.model, self.key = model, key

    def release(self):
        if self.key is not None:
            self.model._slots_busy.discard(self.key)
            self.key = None

    def __del__(self):
        self.release()


class _SeqFn(torch.autograd.Function):
    """BSARecModel.forward as an autograd node (src/model/bsarec.py:16-28): the last layer's output on all positions,
    differentiable w.r.t. every parameter.  Each call keeps its own plan (activations + a snapshot of the dropout
    step) until its backward ran, so several forwards can be alive in one graph (DuoRec's three, duorec.py:95-127)."""

    @staticmethod
    def forward(ctx, model, ids, *params):
        B = ids.shape[0]
        slot = 1
        while (B, slot) in model._slots_busy:
            slot += 1
        model._slots_busy.add((B, slot))
        ctx.token = _SlotToken(model, (B, slot))
        plan = model._run_forward(ids, train=model.training, new_step=model.training, slot=slot)
        ctx.model, ctx.plan = model, plan
        N, Lq, d = model.args.num_hidden_layers, model.args.max_seq_length, model.args.hidden_size
        return plan.view(L.BUF_LAYER_OUT, N, (B, Lq, d)).clone()

    @staticmethod
    def backward(ctx, gout):
        model, plan = ctx.model, ctx.plan
        g = gout.to(torch.float32).contiguous()
        L.check(plan.lib.bsarec_backward_seq(plan.handle, g.data_ptr(), model._stream()), "bsarec_backward_seq")
        grads = model._garena.clone()
        ctx.token.release()
        return (None, None) + model._grads_in_param_order(grads)


class _SeqAllFn(torch.autograd.Function):
    """BSARecModel.forward(all_sequence_output=True) as ONE autograd node with N + 1 outputs (src/model/bsarec.py:46-54: the
    embedding output and every block's output): each element of the list carries gradients, as the reference's list does
    -- ``bsarec_backward_seq_multi`` joins an intermediate output's upstream gradient with the one flowing down from above."""

    @staticmethod
    def forward(ctx, model, ids, *params):
        B = ids.shape[0]
        slot = 1
        while (B, slot) in model._slots_busy:
            slot += 1
        model._slots_busy.add((B, slot))
        ctx.token = _SlotToken(model, (B, slot))
        plan = model._run_forward(ids, train=model.training, new_step=model.training, slot=slot)
        ctx.model, ctx.plan = model, plan
        N, Lq, d = model.args.num_hidden_layers, model.args.max_seq_length, model.args.hidden_size
        return tuple(plan.view(L.BUF_LAYER_OUT, l, (B, Lq, d)).float().clone() for l in range(N + 1))

    @staticmethod
    def backward(ctx, *gouts):
        import ctypes as C
        model, plan = ctx.model, ctx.plan
        N = model.args.num_hidden_layers
        gs = [None if g is None else g.to(torch.float32).contiguous() for g in gouts]
        if gs[N] is None:
            gs[N] = torch.zeros_like(next(g for g in gs if g is not None))
        ptrs = (C.c_void_p * (N + 1))(*[None if g is None else g.data_ptr() for g in gs])
        L.check(plan.lib.bsarec_backward_seq_multi(plan.handle, ptrs, model._stream()), "bsarec_backward_seq_multi")
        grads = model._garena.clone()
        ctx.token.release()
        return (None, None) + model._grads_in_param_order(grads)


class _LossFn(torch.autograd.Function):
    """calculate_loss as one autograd node: forward+loss kernels now, the whole backward chain when
    autograd calls back (src/trainers.py:103-106)."""

    @staticmethod
    def forward(ctx, model, ids, answers, *params):
        plan = model._run_forward(ids, train=model.training, new_step=model.training, last_only=True)
        model._run_loss(plan, answers)
        ctx.model, ctx.plan = model, plan
        return plan.view(L.BUF_LOSS, 0, (1,))[0].clone()

    @staticmethod
    def backward(ctx, gout):
        model, plan = ctx.model, ctx.plan
        model._run_backward(plan)
        g = model._garena * gout
        return (None, None, None) + model._grads_in_param_order(g)


class BSARecModel(nn.Module):
    """Drop-in for ``MODEL_DICT['bsarec'](args=args)`` (src/main.py:31)."""

    def __init__(self, args):
        super().__init__()
        if args.hidden_size % args.num_attention_heads != 0:           # src/model/_modules.py:79-82
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (args.hidden_size, args.num_attention_heads))
        if getattr(args, "hidden_act", "gelu") not in L.HIDDEN_ACTS:       # src/model/_modules.py:38-45
            raise KeyError(getattr(args, "hidden_act"))
        self.args = args
        # per-model plan options (include/bsarec_hip.h, bsarec_config_t tail), e.g. {"storage": 1} for bf16 storage
        self.options = dict(getattr(args, "plan_options", None) or {})
        if getattr(args, "storage", None) == "bf16":
            self.options["storage"] = 1
        self.batch_size = getattr(args, "batch_size", 256)              # stored, unused (as in the reference)
        self.cutoff_bins = min(args.c // 2 + 1, args.max_seq_length // 2 + 1)   # src/model/bsarec.py:87,96
        if getattr(args, "filter_kind", 0) == 1:
            self.cutoff_bins = args.max_seq_length // 2 + 1                     # src/model/fmlprec.py:99
        shapes = param_shapes(args)
        self._slices: "OrderedDict[str, Tuple[int, int, Tuple[int, ...]]]" = OrderedDict()
        off = 0
        for k, shp in shapes.items():
            n = int(np.prod(shp))
            self._slices[k] = (off, n, shp)
            off += n
        self._numel = off
        arena = torch.zeros(off, dtype=torch.float32)
        self._build_tree(arena)
        self._install_arena(arena)
        self.init_weights()
        self._plans: Dict[Tuple[int, str], _Plan] = {}
        self._slots_busy = set()
        self._seed = int(getattr(args, "seed", 42))

    # ---- module tree with the reference's parameter paths -------------------------------------
    def _build_tree(self, arena):
        for key, (o, n, shp) in self._slices.items():
            parts = key.split(".")
            mod = self
            for name in parts[:-1]:
                if not hasattr(mod, name):
                    mod.add_module(name, _Bag())
                mod = getattr(mod, name)
            mod.register_parameter(parts[-1], nn.Parameter(arena[o:o + n].view(shp)))

    def _grads_in_param_order(self, g):
        """Slices of the flat gradient ``g`` in the order ``self.parameters()`` yields them (module registration order;
        equal to the arena order except for tensors appended to the arena by a sibling model)."""
        out = []
        for name, _ in self.named_parameters():
            o, n, shp = self._slices[name]
            out.append(g[o:o + n].view(shp))
        return tuple(out)

    def _param_by_key(self, key):
        mod = self
        parts = key.split(".")
        for name in parts[:-1]:
            mod = getattr(mod, name)
        return mod._parameters[parts[-1]]

    def _install_arena(self, arena):
        """(Re)point every parameter at its slice of ``arena``; allocate grad arena + device state."""
        self._arena = arena
        self._garena = torch.zeros_like(arena)
        for key, (o, n, shp) in self._slices.items():
            p = self._param_by_key(key)
            p.data = arena[o:o + n].view(shp)
            p.grad = None
        self._state = torch.zeros(8, dtype=torch.int64, device=arena.device)
        self._twiddle = _twiddle(self.args.max_seq_length).to(arena.device)
        self._plans = {}
        self._adam = None
        self._shadow = None               # bf16 mirror of the arena (plans with storage = 1), allocated on first use
        self._shadow_stale = True
        self._garena_alt = None           # second gradient arena (peer-to-peer exchange: arenas alternate by step parity)
        self._lookup = None               # [V*d] lookup-path rows of the item-table gradient (bucketed exchange)
        self._dense_hook = None

    def use_grad_arenas(self, arenas):
        """Data parallel, ``p2p`` exchange: gradients are written into caller-provided arenas (IPC-exported memory);
        ``arenas[parity]`` is the target of the steps with that parity."""
        assert all(a.numel() == self._numel and a.dtype == torch.float32 for a in arenas)
        self._garena, self._garena_alt = arenas[0], arenas[1]
        self._garena.zero_(); self._garena_alt.zero_()
        self._plans = {}

    def enable_lookup_grad(self):
        """Data parallel, bucketed exchange: one buffer [gradient arena | lookup rows of the item table] so that the
        second bucket (encoder gradients + lookup rows) is one contiguous message."""
        vd = self._slices["item_embeddings.weight"][1]
        self._gbuf = torch.zeros(self._numel + vd, dtype=torch.float32, device=self._arena.device)
        self._garena, self._lookup = self._gbuf[:self._numel], self._gbuf[self._numel:]
        self._plans = {}

    def set_dense_grad_hook(self, fn):
        """``fn(stream)`` is called as soon as the dense item-table gradient is enqueued (bsarec_plan_set_dense_grad_hook);
        applies to the plans created from now on and to the existing ones."""
        self._dense_hook_py = fn
        self._dense_hook = L.HOOK(lambda user, stream: fn(stream)) if fn is not None else None
        for plan in self._plans.values():
            self._install_hook(plan)

    def _install_hook(self, plan):
        lk = self._lookup.data_ptr() if (self._lookup is not None and plan.garena is self._garena) else None
        hook = self._dense_hook if self._dense_hook is not None else L.HOOK(0)
        L.check(plan.lib.bsarec_plan_set_dense_grad_hook(plan.handle, hook, None, lk), "bsarec_plan_set_dense_grad_hook")

    def _shadow_arena(self):
        if self._shadow is None:
            self._shadow = self._arena.to(torch.bfloat16)
            self._shadow_stale = True
        return self._shadow

    def _refresh_shadow(self, plan, force=False):
        """bf16 storage: the MFMA products read a bf16 shadow of the Linear weights.  The fused Adam keeps it current;
        after anything else that changes the fp32 masters (load_state_dict, init, an external optimiser) it is rebuilt."""
        if plan.bf16 and (force or self._shadow_stale):
            L.check(plan.lib.bsarec_shadow_refresh(plan.handle, self._stream()), "bsarec_shadow_refresh")
            self._shadow_stale = False

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._shadow_stale = True
        return out

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        # .cuda()/.cpu()/.to() re-allocated every parameter separately: gather them back into one arena
        ref = self._param_by_key("item_embeddings.weight")
        arena = torch.empty(self._numel, dtype=torch.float32, device=ref.device)
        for key, (o, n, shp) in self._slices.items():
            arena[o:o + n] = self._param_by_key(key).data.reshape(-1).to(torch.float32)
        seed = self._seed if hasattr(self, "_seed") else 42
        self._install_arena(arena)
        self._seed = seed
        return out

    def init_weights(self, module=None):
        """src/model/_abstract_model.py:26-39 + src/model/bsarec.py:88."""
        std = self.args.initializer_range
        with torch.no_grad():
            for key, (o, n, shp) in self._slices.items():
                p = self._param_by_key(key)
                if key.endswith("sqrt_beta"):
                    p.normal_(0.0, 1.0)
                elif key.endswith("complex_weight"):
                    p.normal_(0.0, 1.0).mul_(0.02)          # src/model/fmlprec.py:99 (randn * 0.02)
                elif "LayerNorm.weight" in key:
                    p.fill_(1.0)
                elif key.endswith(".bias"):
                    p.zero_()
                else:
                    p.normal_(0.0, std)          # Linear / Embedding weights, padding row 0 included

    # ---- C ABI plumbing -----------------------------------------------------------------------
    def _tensor_struct(self, arena) -> L.Tensors:
        t = L.Tensors()
        base, esz = arena.data_ptr(), arena.element_size()       # fp32 arenas, or the bf16 shadow (same element offsets)

        def ptr(key):
            return base + esz * self._slices[key][0]
        for f, key in L.TOP_KEYS.items():
            setattr(t, f, ptr(key))
        for l in range(self.args.num_hidden_layers):
            for f, suffix in L.LAYER_KEYS.items():
                setattr(t.layer[l], f, ptr(f"item_encoder.blocks.{l}.{suffix}"))
            for f, suffix in L.OPTIONAL_LAYER_KEYS.items():
                key = f"item_encoder.blocks.{l}.{suffix}"
                setattr(t.layer[l], f, ptr(key) if key in self._slices else None)
        return t

    def _require_gpu(self):
        if self._arena.device.type != "cuda":
            raise RuntimeError("bsarec_amd runs on MI355X only: move the model with .cuda() "
                               "(there is no CPU fallback; the CPU oracle lives under oracle/ for tests)")

    def set_seed(self, seed: int, rank: int = 0):
        """Seed of the device-side Philox dropout stream (rank-decorrelated for data parallel)."""
        self._seed = int(seed)
        s = (int(seed) ^ (rank * 0x9E3779B97F4A7C15)) & 0x7FFFFFFFFFFFFFFF
        self._state[0] = s

    def _plan(self, batch: int, parity: int = 0, slot: int = 0) -> _Plan:
        self._require_gpu()
        opts = L.default_options()
        opts.update(self.options)
        key = (batch, str(self._arena.device), tuple(sorted(opts.items())), parity, slot)
        if key not in self._plans:
            if int(self._state[0].item()) == 0:
                self.set_seed(self._seed)
            if slot > 0:        # a retained forward of the autograd API: own workspace, own snapshot of the step state
                self._plans[key] = _Plan(self, batch, own_state=True)
            elif parity == 0:
                self._plans[key] = _Plan(self, batch)
            else:       # the other gradient arena; its own workspace (the reduction job table inside it names the arena)
                self._plans[key] = _Plan(self, batch, garena=self._garena_alt)
            if self._lookup is not None or self._dense_hook is not None:
                self._install_hook(self._plans[key])
        return self._plans[key]

    def _stream(self):
        return torch.cuda.current_stream(self._arena.device).cuda_stream

    def _run_forward(self, input_ids, train: bool, new_step: bool, last_only: bool = False, slot: int = 0) -> _Plan:
        if input_ids.dim() != 2 or input_ids.shape[1] != self.args.max_seq_length:
            raise ValueError("input_ids must be [B, max_seq_length]")
        ids = input_ids.to(device=self._arena.device, dtype=torch.int64).contiguous()
        plan = self._plan(ids.shape[0], slot=slot)
        lib, st = plan.lib, self._stream()
        if new_step:
            if slot:
                self._state[1:2].add_(1)     # (the plan of a retained forward has a private state: advance the model's)
            else:
                L.check(lib.bsarec_step_begin(plan.handle, st), "bsarec_step_begin")
            self._step_begun = True          # the counter now holds a value in use (see _fresh_step_counter)
        if slot:
            plan.state.copy_(self._state)    # this forward's own dropout step, whatever runs before its backward
        # the eager API may follow an update of the masters this module cannot see (torch.optim over model.parameters()):
        # always rebuild the bf16 shadow here (storage = 1 only; one small launch per layer)
        self._refresh_shadow(plan, force=True)
        self._shadow_stale = True
        # last_only: the caller consumes position L-1 of the last layer only (loss / logits / backward)
        fwd = lib.bsarec_forward_last if last_only else lib.bsarec_forward
        L.check(fwd(plan.handle, ids.data_ptr(), 1 if train else 0, st), "bsarec_forward")
        plan._ids_keepalive = ids
        return plan

    def _fresh_step_counter(self, plan: _Plan):
        """The eager calls advance the dropout step counter BEFORE using it (bsarec_step_begin); the indexed steps use
        the counter as it stands and advance it when the step closes (no extra launch in the captured graph).  A
        begin-style step followed by an indexed one would therefore draw the same masks twice: advance once in between."""
        if getattr(self, "_step_begun", False):
            L.check(plan.lib.bsarec_step_begin(plan.handle, self._stream()), "bsarec_step_begin")
            self._step_begun = False

    def _run_loss(self, plan: _Plan, answers):
        ans = answers.to(device=self._arena.device, dtype=torch.int64).contiguous()
        L.check(plan.lib.bsarec_loss(plan.handle, ans.data_ptr(), self._stream()), "bsarec_loss")
        plan._ans_keepalive = ans

    def _run_backward(self, plan: _Plan):
        L.check(plan.lib.bsarec_backward(plan.handle, self._stream()), "bsarec_backward")

    # ---- reference model API --------------------------------------------------------------------
    def forward(self, input_ids, user_ids=None, all_sequence_output=False):
        """src/model/bsarec.py:16-28.  In train mode with autograd enabled the output is differentiable w.r.t. every parameter (an
        autograd node over bsarec_forward / bsarec_backward_seq); with ``all_sequence_output=True`` EVERY element of the
        returned list is (one node with N + 1 outputs, bsarec_backward_seq_multi), as in the reference where the whole list
        is one autograd graph (bsarec.py:46-54).  :meth:`calculate_loss` stays the fast path of the trainer."""
        B, Lq, d = input_ids.shape[0], self.args.max_seq_length, self.args.hidden_size
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if all_sequence_output:
                return list(_SeqAllFn.apply(self, input_ids, *self.parameters()))
            return _SeqFn.apply(self, input_ids, *self.parameters())
        plan = self._run_forward(input_ids, train=self.training, new_step=self.training)
        if all_sequence_output:
            return [plan.view(L.BUF_LAYER_OUT, l, (B, Lq, d)).float().clone() for l in range(self.args.num_hidden_layers + 1)]
        return plan.view(L.BUF_LAYER_OUT, self.args.num_hidden_layers, (B, Lq, d)).clone()

    def predict(self, input_ids, user_ids=None, all_sequence_output=False):
        """src/model/_abstract_model.py:74-75."""
        return self.forward(input_ids, user_ids, all_sequence_output)

    def calculate_loss(self, input_ids, answers, neg_answers=None, same_target=None, user_ids=None):
        """src/model/bsarec.py:30-37: 0-d loss tensor; ``.backward()`` fills every parameter's grad."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _LossFn.apply(self, input_ids, answers, *self.parameters())
        plan = self._run_forward(input_ids, train=self.training, new_step=self.training, last_only=True)
        self._run_loss(plan, answers)
        return plan.view(L.BUF_LOSS, 0, (1,))[0].clone()

    def full_logits(self, input_ids) -> torch.Tensor:
        """Last-position scores over the whole catalogue (Trainer.predict_full of the last position,
        src/trainers.py:62-68,126-129) without leaving the device."""
        plan = self._run_forward(input_ids, train=False, new_step=False, last_only=True)
        L.check(plan.lib.bsarec_logits(plan.handle, self._stream()), "bsarec_logits")
        V = self.args.item_size
        Vp = (V + 3) // 4 * 4
        return plan.view(L.BUF_LOGITS, 0, (input_ids.shape[0], Vp))[:, :V]

    # ---- fused training step (used by bsarec_amd.trainer.Trainer) ------------------------------
    def configure_adam(self, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self._require_gpu()
        self._adam = dict(lr=float(lr), b1=float(betas[0]), b2=float(betas[1]), eps=float(eps), wd=float(weight_decay),
                          m=torch.zeros_like(self._arena), v=torch.zeros_like(self._arena))
        self._state[2] = 0

    def _adam_struct(self, grad_scale: float = 1.0, grad_srcs=None) -> L.Adam:
        a = self._adam
        s = L.Adam(self._arena.data_ptr(), self._garena.data_ptr(), a["m"].data_ptr(), a["v"].data_ptr(), self._numel,
                   a["lr"], a["b1"], a["b2"], a["eps"], a["wd"], float(grad_scale), None, 0)
        if self._lookup is not None:            # bucketed exchange: the lookup rows of the item table sit in their own buffer
            s.grads2, s.grads2_n = self._lookup.data_ptr(), self._lookup.numel()
        if grad_srcs:                           # peer-to-peer exchange: sum of every rank's arena, in rank order
            s.n_grad_srcs = len(grad_srcs)
            for i, p in enumerate(grad_srcs):
                s.grad_srcs[i] = p
        if self._shadow is not None:           # keep the bf16 shadow of everything behind the item table current
            s.shadow_bf16 = self._shadow.data_ptr()
            s.shadow_from = self._slices["position_embeddings.weight"][0]
        return s

    def train_step(self, input_ids, answers) -> torch.Tensor:
        """step_begin + forward + loss + backward + Adam in one C call (src/trainers.py:100-107).
        Returns a view of the device loss scalar (no host sync)."""
        if self._adam is None:
            raise RuntimeError("call configure_adam() first")
        ids = input_ids.to(device=self._arena.device, dtype=torch.int64).contiguous()
        ans = answers.to(device=self._arena.device, dtype=torch.int64).contiguous()
        plan = self._plan(ids.shape[0])
        self._refresh_shadow(plan)
        ad = self._adam_struct()
        L.check(plan.lib.bsarec_train_step(plan.handle, ids.data_ptr(), ans.data_ptr(), C.byref(ad), self._stream()),
                "bsarec_train_step")
        self._step_begun = True
        plan._ids_keepalive, plan._ans_keepalive = ids, ans
        return plan.view(L.BUF_LOSS, 0, (1,))[0]

    def train_step_indexed(self, table, answers_table, perm, cursor, batch: int) -> torch.Tensor:
        """train_step fed from a device-resident sample table: the batch perm[cursor : cursor+batch] is gathered
        on the device and the cursor advanced by the same C call, so a captured graph replays a whole epoch
        without any host-side tensor work."""
        if self._adam is None:
            raise RuntimeError("call configure_adam() first")
        plan = self._plan(batch)
        if not hasattr(plan, "ids_buf"):
            dev = self._arena.device
            plan.ids_buf = torch.zeros((batch, self.args.max_seq_length), dtype=torch.int64, device=dev)
            plan.ans_buf = torch.zeros((batch,), dtype=torch.int64, device=dev)
        self._fresh_step_counter(plan)
        self._refresh_shadow(plan)
        ad = self._adam_struct()
        L.check(plan.lib.bsarec_train_step_indexed(
            plan.handle, table.data_ptr(), answers_table.data_ptr(), perm.data_ptr(), perm.shape[0], cursor.data_ptr(),
            plan.ids_buf.data_ptr(), plan.ans_buf.data_ptr(), C.byref(ad), self._stream()), "bsarec_train_step_indexed")
        return plan.view(L.BUF_LOSS, 0, (1,))[0]

    def grad_step_indexed(self, table, answers_table, perm, cursor, batch: int, tick_adam: bool = False,
                          parity: int = 0) -> torch.Tensor:
        """Data-parallel half of train_step_indexed: gather + forward + loss + backward into the gradient arena
        (no Adam).  Follow with an all-reduce of ``_garena`` and :meth:`adam_step` -- or, with ``tick_adam``, the step's
        closing block already advances Adam's t / bias corrections and :meth:`adam_step(tick=False)` applies the update.
        ``parity`` selects the gradient arena (:meth:`use_grad_arenas`)."""
        plan = self._plan(batch, parity)
        if not hasattr(plan, "ids_buf"):
            dev = self._arena.device
            plan.ids_buf = torch.zeros((batch, self.args.max_seq_length), dtype=torch.int64, device=dev)
            plan.ans_buf = torch.zeros((batch,), dtype=torch.int64, device=dev)
        self._fresh_step_counter(plan)
        self._refresh_shadow(plan)
        L.check(plan.lib.bsarec_grad_step_indexed(
            plan.handle, table.data_ptr(), answers_table.data_ptr(), perm.data_ptr(), perm.shape[0], cursor.data_ptr(),
            plan.ids_buf.data_ptr(), plan.ans_buf.data_ptr(), self._adam["lr"] if tick_adam else 0.0,
            self._adam["b1"] if tick_adam else 0.0, self._adam["b2"] if tick_adam else 0.0, self._stream()),
            "bsarec_grad_step_indexed")
        return plan.view(L.BUF_LOSS, 0, (1,))[0]

    def adam_step(self, grad_scale: float = 1.0, tick: bool = True, grad_srcs=None):
        """Fused Adam over the flat arenas (after an external gradient exchange, or -- ``grad_srcs`` -- reading every
        rank's gradient arena itself)."""
        ad = self._adam_struct(grad_scale, grad_srcs)
        fn = L.load().bsarec_adam_step if tick else L.load().bsarec_adam_apply
        L.check(fn(C.byref(ad), self._state.data_ptr(), self._stream()), "bsarec_adam_step" if tick else "bsarec_adam_apply")

    def grad_views(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((k, self._garena[o:o + n].view(shp)) for k, (o, n, shp) in self._slices.items())


class _BCEFn(torch.autograd.Function):
    """SASRec's loss head as one autograd node (bsarec_forward_last + bsarec_loss_bce; backward: bsarec_backward)."""

    @staticmethod
    def forward(ctx, model, ids, pos, neg, *params):
        plan = model._run_forward(ids, train=model.training, new_step=model.training, last_only=True)
        model._run_loss_bce(plan, pos, neg)
        ctx.model, ctx.plan = model, plan
        return plan.view(L.BUF_LOSS, 0, (1,))[0].clone()

    @staticmethod
    def backward(ctx, gout):
        model, plan = ctx.model, ctx.plan
        model._run_backward(plan)
        g = model._garena * gout
        return (None, None, None, None) + model._grads_in_param_order(g)


class SASRecModel(BSARecModel):
    """Sibling model on the same kernels: ``MODEL_DICT['sasrec'](args=args)``, src/model/sasrec.py.

    The reference's TransformerBlock (src/model/_modules.py:142-151) is a BSARecBlock with alpha = 0 -- the mix
    ``alpha * dsp + (1 - alpha) * gsp`` returns the attention branch -- so the encoder runs the BSARec plan with
    alpha = 0; the frequency-layer tensors exist in the arena but receive exactly zero gradient and are neither
    saved nor loaded.  What differs is the loss head (one positive / one negative logit, BCE, sasrec.py:41-63) and the
    state_dict key names (``...blocks.{l}.layer.query.weight`` instead of ``...layer.attention_layer.query.weight``;
    36 keys)."""

    needs_negatives = True                       # Trainer: feed neg_answer (DeviceBatches.enable_negatives)

    def __init__(self, args):
        import copy
        a = copy.copy(args)
        a.alpha = 0.0
        if not hasattr(a, "c"):
            a.c = 3
        super().__init__(a)

    @staticmethod
    def _ref_key(k: str):
        if ".filter_layer." in k:
            return None
        return k.replace(".layer.attention_layer.", ".layer.")

    def state_dict(self, *a, **kw):
        sd = super().state_dict(*a, **kw)
        return OrderedDict((self._ref_key(k), v) for k, v in sd.items() if self._ref_key(k) is not None)

    def load_state_dict(self, state_dict, strict=True):
        own = super().state_dict()
        full = OrderedDict((k, v) for k, v in own.items())            # frequency-layer tensors keep their values
        for k, v in state_dict.items():
            kk = k.replace(".layer.", ".layer.attention_layer.") if (".layer." in k and ".attention_layer." not in k) else k
            if kk not in own:
                if strict:
                    raise KeyError(f"unexpected key {k}")
                continue
            full[kk] = v
        return super().load_state_dict(full, strict=strict)

    def _run_loss_bce(self, plan, pos, neg):
        dev = self._arena.device
        p_ = pos.to(device=dev, dtype=torch.int64).contiguous()
        n_ = neg.to(device=dev, dtype=torch.int64).contiguous()
        L.check(plan.lib.bsarec_loss_bce(plan.handle, p_.data_ptr(), n_.data_ptr(), self._stream()), "bsarec_loss_bce")
        plan._bce_keepalive = (p_, n_)

    def calculate_loss(self, input_ids, answers, neg_answers=None, same_target=None, user_ids=None):
        """src/model/sasrec.py:41-63: 0-d loss tensor; ``.backward()`` fills every parameter's grad."""
        if neg_answers is None:
            raise ValueError("SASRec's loss needs neg_answers (src/dataset.py:67)")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _BCEFn.apply(self, input_ids, answers, neg_answers, *self.parameters())
        plan = self._run_forward(input_ids, train=self.training, new_step=self.training, last_only=True)
        self._run_loss_bce(plan, answers, neg_answers)
        return plan.view(L.BUF_LOSS, 0, (1,))[0].clone()

    def grad_step(self, input_ids, answers, neg_answers=None) -> torch.Tensor:
        """step_begin + forward + BCE head + backward into the gradient arena (the data-parallel half of train_step: the
        caller exchanges ``_garena`` and calls :meth:`adam_step`)."""
        plan = self._run_forward(input_ids, train=True, new_step=True, last_only=True)
        self._run_loss_bce(plan, answers, neg_answers)
        self._run_backward(plan)
        return plan.view(L.BUF_LOSS, 0, (1,))[0]

    def train_step(self, input_ids, answers, neg_answers=None) -> torch.Tensor:
        """step_begin + forward + BCE head + backward + fused Adam, all on the device (five C calls)."""
        if self._adam is None:
            raise RuntimeError("call configure_adam() first")
        loss = self.grad_step(input_ids, answers, neg_answers)
        self.adam_step()
        return loss


class _LogSigFn(_BCEFn):
    @staticmethod
    def forward(ctx, model, ids, pos, neg, *params):
        plan = model._run_forward(ids, train=model.training, new_step=model.training, last_only=True)
        model._run_loss_pair(plan, pos, neg)
        ctx.model, ctx.plan = model, plan
        return plan.view(L.BUF_LOSS, 0, (1,))[0].clone()


class FMLPRecModel(BSARecModel):
    """Sibling model (the "learnable complex filter" of the north star): ``MODEL_DICT['fmlprec']``,
    src/model/fmlprec.py.  An FMLPRecBlock is a BSARecBlock with alpha = 1 (only the frequency branch reaches the
    feed-forward) whose filter is ``irfft(rfft(x) * complex_weight)`` instead of BSARec's low-pass / beta^2 mix
    (``filter_kind = 1``: the generic frequency kernels multiply the spectrum by the weight, all L//2+1 bins kept).
    The attention tensors and sqrt_beta exist in the arena, receive exactly zero gradient and are neither saved nor
    loaded; state_dict uses the reference's names (``...layer.complex_weight``, ``...layer.LayerNorm.*``; 4 + 9 N keys).
    Loss head: -log(sigmoid(x_pos) + 1e-24) - log(1 - sigmoid(x_neg) + 1e-24), mean over the batch (fmlprec.py:41-62).
    At the fused shape (hidden = 64, L <= 64) the block runs in the FM instantiation of the per-sequence block kernels (filter
    + feed-forward, no attention branch); elsewhere on the generic tiled kernels."""

    needs_negatives = True

    def __init__(self, args):
        import copy
        a = copy.copy(args)
        a.alpha = 1.0
        a.filter_kind = 1
        if not hasattr(a, "c"):
            a.c = 3
        super().__init__(a)

    @staticmethod
    def _ref_key(k: str):
        if ".attention_layer." in k or k.endswith("sqrt_beta"):
            return None
        return k.replace(".layer.filter_layer.", ".layer.")

    def _ref_order(self):
        """The reference registers complex_weight first inside a block (fmlprec.py:99-101)."""
        keys = [k for k in self._slices if self._ref_key(k) is not None]
        top = [k for k in keys if not k.startswith("item_encoder.")]
        out = list(top)
        for l in range(self.args.num_hidden_layers):
            pre = f"item_encoder.blocks.{l}."
            blk = [k for k in keys if k.startswith(pre)]
            cw = [k for k in blk if k.endswith("complex_weight")]
            out += cw + [k for k in blk if not k.endswith("complex_weight")]
        return out

    def state_dict(self, *a, **kw):
        sd = super().state_dict(*a, **kw)
        return OrderedDict((self._ref_key(k), sd[k]) for k in self._ref_order())

    def load_state_dict(self, state_dict, strict=True):
        own = super().state_dict()
        full = OrderedDict((k, v) for k, v in own.items())
        for k, v in state_dict.items():
            kk = k.replace(".layer.", ".layer.filter_layer.") if (".layer." in k and ".filter_layer." not in k) else k
            if kk not in own:
                if strict:
                    raise KeyError(f"unexpected key {k}")
                continue
            full[kk] = v
        return super().load_state_dict(full, strict=strict)

    def _run_loss_pair(self, plan, pos, neg):
        dev = self._arena.device
        p_ = pos.to(device=dev, dtype=torch.int64).contiguous()
        n_ = neg.to(device=dev, dtype=torch.int64).contiguous()
        L.check(plan.lib.bsarec_loss_logsig(plan.handle, p_.data_ptr(), n_.data_ptr(), self._stream()), "bsarec_loss_logsig")
        plan._pair_keepalive = (p_, n_)

    def calculate_loss(self, input_ids, answers, neg_answers=None, same_target=None, user_ids=None):
        """src/model/fmlprec.py:41-62."""
        if neg_answers is None:
            raise ValueError("FMLPRec's loss needs neg_answers (src/dataset.py:67)")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _LogSigFn.apply(self, input_ids, answers, neg_answers, *self.parameters())
        plan = self._run_forward(input_ids, train=self.training, new_step=self.training, last_only=True)
        self._run_loss_pair(plan, answers, neg_answers)
        return plan.view(L.BUF_LOSS, 0, (1,))[0].clone()

    def grad_step(self, input_ids, answers, neg_answers=None) -> torch.Tensor:
        plan = self._run_forward(input_ids, train=True, new_step=True, last_only=True)
        self._run_loss_pair(plan, answers, neg_answers)
        self._run_backward(plan)
        return plan.view(L.BUF_LOSS, 0, (1,))[0]

    def train_step(self, input_ids, answers, neg_answers=None) -> torch.Tensor:
        if self._adam is None:
            raise RuntimeError("call configure_adam() first")
        loss = self.grad_step(input_ids, answers, neg_answers)
        self.adam_step()
        return loss


class DuoRecModel(SASRecModel):
    """Sibling model: ``MODEL_DICT['duorec']``, src/model/duorec.py.  The encoder is SASRec's
    TransformerEncoder (a BSARec plan with alpha = 0, same 36 state_dict keys); the loss is the full-catalogue CE of the
    last position plus InfoNCE terms between the last-position outputs of up to three forward passes of one step --
    the input, the input again under another dropout draw, and a same-target sequence (duorec.py:95-127).  The three
    passes are three live nodes of the differentiable :meth:`forward` (each keeps its own activations and dropout step
    until its backward); the head itself -- two small matmuls and a softmax over 2B x 2B similarities -- is the
    reference's torch code restated, on the device."""

    needs_negatives = False
    needs_same_target = True                     # Trainer / DeviceBatches: feed same_target rows (src/dataset.py:82-96)
    torch_optim = True                           # several backward passes accumulate through autograd: torch.optim.Adam steps

    def __init__(self, args):
        super().__init__(args)
        self.tau = float(getattr(args, "tau", 1.0))
        self.ssl = getattr(args, "ssl", "us_x")
        self.sim = getattr(args, "sim", "dot")
        self.lmd = float(getattr(args, "lmd", 0.1))
        self.lmd_sem = float(getattr(args, "lmd_sem", 0.1))

    @staticmethod
    def _mask_correlated(batch_size, device):
        n = 2 * batch_size
        mask = torch.ones((n, n), dtype=torch.bool, device=device)
        mask.fill_diagonal_(False)
        idx = torch.arange(batch_size, device=device)
        mask[idx, batch_size + idx] = False
        mask[batch_size + idx, idx] = False
        return mask

    def info_nce(self, z_i, z_j, temp, batch_size, sim="dot"):
        """duorec.py:47-78: positives = the two views of a sequence, negatives = the other 2(B - 1) rows of the batch."""
        n = 2 * batch_size
        z = torch.cat((z_i, z_j), dim=0)
        if sim == "cos":
            s = torch.nn.functional.cosine_similarity(z.unsqueeze(1), z.unsqueeze(0), dim=2) / temp
        else:
            s = torch.mm(z, z.T) / temp
        pos = torch.cat((torch.diag(s, batch_size), torch.diag(s, -batch_size)), dim=0).reshape(n, 1)
        neg = s[self._mask_correlated(batch_size, z.device)].reshape(n, -1)
        labels = torch.zeros(n, dtype=torch.long, device=z.device)
        return torch.cat((pos, neg), dim=1), labels

    def calculate_loss(self, input_ids, answers, neg_answers=None, same_target=None, user_ids=None):
        """src/model/duorec.py:95-127."""
        ce = torch.nn.functional.cross_entropy
        B = input_ids.shape[0]
        seq_output = self.forward(input_ids)[:, -1, :]
        logits = torch.matmul(seq_output, self.item_embeddings.weight.transpose(0, 1))
        loss = ce(logits, answers.to(logits.device))
        if self.ssl in ("us", "un"):
            aug = self.forward(input_ids)[:, -1, :]
            lg, lb = self.info_nce(seq_output, aug, self.tau, B, self.sim)
            loss = loss + self.lmd * ce(lg, lb)
        if self.ssl in ("us", "su"):
            sem = self.forward(same_target)[:, -1, :]
            lg, lb = self.info_nce(seq_output, sem, self.tau, B, self.sim)
            loss = loss + self.lmd_sem * ce(lg, lb)
        if self.ssl == "us_x":
            aug = self.forward(input_ids)[:, -1, :]
            sem = self.forward(same_target)[:, -1, :]
            lg, lb = self.info_nce(aug, sem, self.tau, B, self.sim)
            loss = loss + self.lmd_sem * ce(lg, lb)
        return loss

    def train_step(self, input_ids, answers, neg_answers=None, same_target=None):
        raise RuntimeError("DuoRec steps through calculate_loss / backward / torch.optim.Adam (Trainer does)")


# src/model/__init__.py:10-19: the hot-path entry and its siblings on the same kernels
MODEL_DICT = {"bsarec": BSARecModel, "sasrec": SASRecModel, "fmlprec": FMLPRecModel, "duorec": DuoRecModel}
