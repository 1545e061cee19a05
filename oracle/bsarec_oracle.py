"""CPU oracle for the BSARec training hot path -- TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference algorithm (Sun-Sir/BSARec,
``/root/reference/src``) with a hand-derived backward pass.  It is the checker
for the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under
``bsarec_amd/`` imports it and the product path never falls back to it.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference in
the build container and stores its outputs (per-op, end-to-end forward /
gradients / 3 Adam steps, the two shipped checkpoints' known answers and the
data-pipeline facts) under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks every function here against those vectors.

Each function cites the reference file:line it follows (paths relative to
``/root/reference/``).  Dropout is the one place the oracle cannot follow the
reference bit-for-bit (torch's bernoulli stream is not reproducible): both the
oracle and the HIP kernels draw keep-masks from the same counter-based
Philox4x32-10 stream defined in :func:`dropout_keep`, so masks are identical on
both sides; the reference only pins the 1/(1-p) scaling statistically.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
from scipy.special import erf as _erf

# --------------------------------------------------------------------------------------
# configuration / parameter inventory
# --------------------------------------------------------------------------------------


@dataclass
class Config:
    """Hyper-parameters the reference model reads from ``args``
    (src/utils.py:83-96, src/model/_abstract_model.py:10-12, src/model/bsarec.py:71-87)."""

    item_size: int
    hidden_size: int = 64
    max_seq_length: int = 50
    num_hidden_layers: int = 2
    num_attention_heads: int = 2
    c: int = 3
    alpha: float = 0.9
    hidden_dropout_prob: float = 0.5
    attention_probs_dropout_prob: float = 0.5
    initializer_range: float = 0.02
    eps: float = 1e-12
    hidden_act: str = "gelu"          # src/model/_modules.py:38-45 (ACT2FN): gelu | relu | swish | tanh | sigmoid

    @property
    def head_size(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def cutoff_bins(self) -> int:
        # src/model/bsarec.py:87 (self.c = args.c // 2 + 1); the slice in :96 clamps at L//2+1
        return min(self.c // 2 + 1, self.max_seq_length // 2 + 1)


def param_shapes(cfg: Config) -> "Dict[str, Tuple[int, ...]]":
    """The 42 (= 4 + 19 N) state_dict keys of the reference model, in registration order
    (src/model/_abstract_model.py:10-11, src/model/bsarec.py:11-13,43-44,59-60,71-73,85-88,
    src/model/_modules.py:13-14,29-34,89-97).  Linear weights are [out, in]."""
    d, L, V = cfg.hidden_size, cfg.max_seq_length, cfg.item_size
    s: Dict[str, Tuple[int, ...]] = {}
    s["item_embeddings.weight"] = (V, d)
    s["position_embeddings.weight"] = (L, d)
    s["LayerNorm.weight"] = (d,)
    s["LayerNorm.bias"] = (d,)
    for l in range(cfg.num_hidden_layers):
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
    return s


def init_params(cfg: Config, seed: int = 0, dtype=np.float32) -> Dict[str, np.ndarray]:
    """N(0, initializer_range) for Linear/Embedding weights *including padding row 0*, zero
    biases, LayerNorm gamma=1 beta=0, sqrt_beta ~ N(0,1)
    (src/model/_abstract_model.py:26-39, src/model/bsarec.py:88)."""
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shp in param_shapes(cfg).items():
        if name.endswith("sqrt_beta"):
            a = rng.standard_normal(shp)
        elif "LayerNorm.weight" in name:
            a = np.ones(shp)
        elif name.endswith(".bias"):
            a = np.zeros(shp)
        else:
            a = rng.standard_normal(shp) * cfg.initializer_range
        out[name] = a.astype(dtype)
    return out


# --------------------------------------------------------------------------------------
# Philox4x32-10 dropout stream (shared definition with bsarec_amd/csrc/philox.h)
# --------------------------------------------------------------------------------------

_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = 0x9E3779B9
_PHILOX_W1 = 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0: int, k1: int):
    """Vectorised Philox4x32-10 (Salmon et al., SC'11).  Counters are uint32 arrays, the
    key is two python ints.  Returns four uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint64)
    c1 = np.broadcast_to(np.asarray(c1, dtype=np.uint64), c0.shape)
    c2 = np.broadcast_to(np.asarray(c2, dtype=np.uint64), c0.shape)
    c3 = np.broadcast_to(np.asarray(c3, dtype=np.uint64), c0.shape)
    k0 &= 0xFFFFFFFF
    k1 &= 0xFFFFFFFF
    for _ in range(10):
        p0 = _PHILOX_M0 * c0
        p1 = _PHILOX_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
        k0 = (k0 + _PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + _PHILOX_W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32))


def dropout_threshold(p: float) -> int:
    """An element is kept iff its 32-bit draw >= threshold; threshold = floor(p * 2^32)."""
    return max(0, min(int(p * 4294967296.0), 0xFFFFFFFF))


def dropout_keep(n_elems: int, p: float, seed: int, step: int, site: int) -> np.ndarray:
    """Keep-mask (bool[n_elems]) of dropout site ``site`` at optimisation step ``step``.

    Element ``i`` uses Philox counter (i >> 2, 0, site, step) keyed by the 64-bit seed
    (lo, hi) and takes output word ``i & 3``.  Sites, in the reference's draw order
    (SURVEY A.9): 0 = embedding dropout (src/model/_abstract_model.py:22); for layer l:
    1+4l = FrequencyLayer.out_dropout (src/model/bsarec.py:101), 2+4l = attn_dropout
    (src/model/_modules.py:131), 3+4l = out_dropout (:137), 4+4l = FeedForward.dropout (:66).
    Attention-probability elements are indexed with the key axis padded to a multiple of 4
    (see :func:`attn_dropout_keep`)."""
    if p <= 0.0:
        return np.ones(n_elems, dtype=bool)
    ngrp = (n_elems + 3) // 4
    g = np.arange(ngrp, dtype=np.uint64)
    w = philox4x32_10(g & _MASK32, g >> np.uint64(32), site, step & 0xFFFFFFFF,
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    draws = np.stack(w, axis=1).reshape(-1)[:n_elems]
    return draws >= np.uint32(dropout_threshold(p))


def attn_dropout_keep(B: int, h: int, L: int, p: float, seed: int, step: int, site: int) -> np.ndarray:
    """Keep-mask bool[B,h,L,L]; flat element index ((b*h+head)*L+q)*Lp + k with Lp = 4*ceil(L/4)."""
    Lp = (L + 3) // 4 * 4
    m = dropout_keep(B * h * L * Lp, p, seed, step, site).reshape(B, h, L, Lp)
    return m[..., :L]


# --------------------------------------------------------------------------------------
# primitive ops (forward + backward)
# --------------------------------------------------------------------------------------


def layer_norm_fwd(x, gamma, beta, eps=1e-12):
    """TF-style LayerNorm, eps inside the sqrt, biased variance (src/model/_modules.py:16-20);
    nn.LayerNorm(eps=1e-12) in MultiHeadAttention (:97) is the same function."""
    u = x.mean(-1, keepdims=True)
    xc = x - u
    s = (xc * xc).mean(-1, keepdims=True)
    denom = np.sqrt(s + x.dtype.type(eps))
    xhat = xc / denom
    return gamma * xhat + beta, (xhat, (1.0 / denom).astype(x.dtype))


def layer_norm_bwd(dy, cache, gamma):
    xhat, rstd = cache
    d = dy.shape[-1]
    g = dy * gamma
    dx = rstd * (g - g.mean(-1, keepdims=True) - xhat * (g * xhat).mean(-1, keepdims=True))
    return dx, (dy * xhat).reshape(-1, d).sum(0), dy.reshape(-1, d).sum(0)


def gelu(x):
    """erf-GELU exactly as written in src/model/_modules.py:56."""
    t = x.dtype.type
    return x * t(0.5) * (t(1.0) + _erf(x / t(math.sqrt(2.0))).astype(x.dtype))


def gelu_grad(x):
    t = x.dtype.type
    cdf = t(0.5) * (t(1.0) + _erf(x / t(math.sqrt(2.0))).astype(x.dtype))
    pdf = np.exp(-(x * x) * t(0.5)) * t(1.0 / math.sqrt(2.0 * math.pi))
    return cdf + x * pdf


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def hidden_act_fn(name: str):
    """FeedForward.get_hidden_act (src/model/_modules.py:38-59)."""
    return {"gelu": gelu, "relu": lambda x: np.maximum(x, 0), "swish": lambda x: x * _sigmoid(x), "tanh": np.tanh,
            "sigmoid": _sigmoid}[name]


def hidden_act_grad(name: str):
    def swish_grad(x):
        sg = _sigmoid(x)
        return sg * (1 + x * (1 - sg))
    return {"gelu": gelu_grad, "relu": lambda x: (x > 0).astype(x.dtype), "swish": swish_grad,
            "tanh": lambda x: 1 - np.tanh(x) ** 2, "sigmoid": lambda x: _sigmoid(x) * (1 - _sigmoid(x))}[name]


def attention_mask(ids):
    """Additive causal+padding mask in {0, -10000}, f32[B,1,L,L]
    (src/model/_abstract_model.py:53-69)."""
    B, L = ids.shape
    key_ok = (ids > 0)[:, None, None, :]
    causal = np.tril(np.ones((L, L), dtype=bool))[None, None]
    return ((1.0 - (key_ok & causal).astype(np.float64)) * -10000.0)


def lowpass_matrix(L: int, cb: int, dtype=np.float64) -> np.ndarray:
    """Closed form of irfft(trunc_cb(rfft(., ortho)), n=L, ortho) along the sequence axis:
    the real symmetric idempotent circulant P[i,j] = (1/L) sum_k w_k cos(2 pi k (i-j)/L),
    w_0 = 1, w_k = 2, w_{L/2} = 1 for even L (verified against
    src/model/bsarec.py:93-97 in tests/test_oracle_golden.py)."""
    i = np.arange(L)
    diff = (i[:, None] - i[None, :]).astype(np.float64)
    P = np.zeros((L, L), dtype=np.float64)
    for k in range(min(cb, L // 2 + 1)):
        w = 1.0 if (k == 0 or (L % 2 == 0 and k == L // 2)) else 2.0
        P += w * np.cos(2.0 * np.pi * k * diff / L)
    return (P / L).astype(dtype)


def fmlp_filter(x, cw):
    """FMLPRecLayer's filter (src/model/fmlprec.py:103-108): irfft(rfft(x, dim=1, ortho) * complex_weight, n=L, ortho).
    ``cw`` is the reference's real view [1, L//2+1, d, 2].  Returns (y, spectrum of x) in x's dtype / complex128."""
    L = x.shape[1]
    W = cw[0, :, :, 0].astype(np.float64) + 1j * cw[0, :, :, 1].astype(np.float64)
    X = np.fft.rfft(x.astype(np.float64), axis=1, norm="ortho")
    return np.fft.irfft(X * W[None], n=L, axis=1, norm="ortho").astype(x.dtype), X


def fmlp_filter_bwd(dy, X, cw):
    """Backward of fmlp_filter by the real-matrix form of the two transforms: with C[k,t] = cos(2 pi k t / L),
    S[k,t] = sin(.), w_k = 1 for DC / Nyquist else 2 and the ortho factor 1/sqrt(L) on each side,
      Xr = C x / sqrt(L), Xi = -S x / sqrt(L), Y = X W, y = (C^T (w Yr) - S^T (w Yi)) / sqrt(L).
    Returns (dx, d complex_weight in the reference's [1, K, d, 2] layout)."""
    B, L, d = dy.shape
    K = L // 2 + 1
    k = np.arange(K)[:, None]
    tt = np.arange(L)[None, :]
    ang = 2.0 * np.pi * k * tt / L
    C, S = np.cos(ang), np.sin(ang)
    w = np.full(K, 2.0)
    w[0] = 1.0
    if L % 2 == 0:
        w[-1] = 1.0
    rs = 1.0 / math.sqrt(L)
    Wr, Wi = cw[0, :, :, 0].astype(np.float64), cw[0, :, :, 1].astype(np.float64)
    dy64 = dy.astype(np.float64)
    gYr = rs * w[None, :, None] * np.einsum("kt,btc->bkc", C, dy64)
    gYi = -rs * w[None, :, None] * np.einsum("kt,btc->bkc", S, dy64)
    Xr, Xi = X.real, X.imag
    gW = np.stack([(gYr * Xr + gYi * Xi).sum(0), (-gYr * Xi + gYi * Xr).sum(0)], axis=-1)[None]
    gXr = gYr * Wr[None] + gYi * Wi[None]
    gXi = -gYr * Wi[None] + gYi * Wr[None]
    dx = rs * (np.einsum("kt,bkc->btc", C, gXr) - np.einsum("kt,bkc->btc", S, gXi))
    return dx.astype(dy.dtype), gW.astype(cw.dtype)


def lowpass(x, cb: int):
    """low = irfft(rfft(x, dim=1, ortho)[:, :cb] zero-extended, n=L, dim=1, ortho)
    (src/model/bsarec.py:93-97).  Self-adjoint, so the same call is its own backward."""
    L = x.shape[1]
    spec = np.fft.rfft(x.astype(np.float64), axis=1, norm="ortho")
    spec[:, cb:, :] = 0
    return np.fft.irfft(spec, n=L, axis=1, norm="ortho").astype(x.dtype)


# --------------------------------------------------------------------------------------
# model forward / backward
# --------------------------------------------------------------------------------------


@dataclass
class DropoutSpec:
    """Philox stream selector for one forward pass; ``train=False`` disables dropout
    (model.eval(), src/trainers.py:119)."""

    train: bool = False
    seed: int = 0
    step: int = 0


def _drop(x, p, spec: DropoutSpec, site: int):
    if not spec.train or p <= 0.0:
        return x, None
    keep = dropout_keep(x.size, p, spec.seed, spec.step, site).reshape(x.shape)
    scale = x.dtype.type(1.0 / (1.0 - p))
    return np.where(keep, x * scale, x.dtype.type(0)), keep


def forward(params: Dict[str, np.ndarray], cfg: Config, ids: np.ndarray,
            drop: Optional[DropoutSpec] = None, dtype=np.float32, keep_cache: bool = True):
    """BSARecModel.forward(input_ids, all_sequence_output=True)
    (src/model/bsarec.py:16-28).  Returns (list of N+1 layer outputs, cache)."""
    drop = drop or DropoutSpec()
    P = {k: v.astype(dtype) for k, v in params.items()}
    t = dtype
    B, L = ids.shape
    d, h, dh, N = cfg.hidden_size, cfg.num_attention_heads, cfg.head_size, cfg.num_hidden_layers
    cb = cfg.cutoff_bins
    ph, pa = cfg.hidden_dropout_prob, cfg.attention_probs_dropout_prob
    mask = attention_mask(ids).astype(dtype)                        # _abstract_model.py:53-69
    cache: Dict[str, object] = {"ids": ids, "mask": mask}

    # add_position_embedding: Drop(LN(E[ids] + Pos[0:L]))            _abstract_model.py:14-24
    e = P["item_embeddings.weight"][ids] + P["position_embeddings.weight"][None, :L]
    x, cache["ln0"] = layer_norm_fwd(e, P["LayerNorm.weight"], P["LayerNorm.bias"], cfg.eps)
    x, cache["keep0"] = _drop(x, ph, drop, 0)
    outs = [x]

    for l in range(N):
        p = f"item_encoder.blocks.{l}."
        lc: Dict[str, object] = {"x": x}
        # FrequencyLayer                                              bsarec.py:90-104
        beta = P[p + "layer.filter_layer.sqrt_beta"]
        cwk = p + "layer.filter_layer.complex_weight"
        if cwk in P:                                   # sibling model FMLPRec: learnable complex filter instead
            f, lc["Xspec"] = fmlp_filter(x, P[cwk])
            low = np.zeros_like(x)
        else:
            low = lowpass(x, cb)
            f = low + (beta ** 2) * (x - low)
        fd, lc["keep_f"] = _drop(f, ph, drop, 1 + 4 * l)
        dsp, lc["ln_f"] = layer_norm_fwd(fd + x, P[p + "layer.filter_layer.LayerNorm.weight"],
                                         P[p + "layer.filter_layer.LayerNorm.bias"], cfg.eps)
        lc["low"] = low
        # MultiHeadAttention                                          _modules.py:108-140
        ap = p + "layer.attention_layer."
        q = x @ P[ap + "query.weight"].T + P[ap + "query.bias"]
        k = x @ P[ap + "key.weight"].T + P[ap + "key.bias"]
        v = x @ P[ap + "value.weight"].T + P[ap + "value.bias"]
        qh = q.reshape(B, L, h, dh).transpose(0, 2, 1, 3)
        kh = k.reshape(B, L, h, dh).transpose(0, 2, 1, 3)
        vh = v.reshape(B, L, h, dh).transpose(0, 2, 1, 3)
        s = (qh @ kh.transpose(0, 1, 3, 2)) / t(math.sqrt(dh))      # scale after the product (:121)
        s = s + mask                                                # (:125)
        s = s - s.max(-1, keepdims=True)
        ex = np.exp(s)
        a = ex / ex.sum(-1, keepdims=True)                          # nn.Softmax(dim=-1) (:128)
        if drop.train and pa > 0.0:
            keep_a = attn_dropout_keep(B, h, L, pa, drop.seed, drop.step, 2 + 4 * l)
            ad = np.where(keep_a, a * t(1.0 / (1.0 - pa)), t(0))
        else:
            keep_a, ad = None, a
        ctx = (ad @ vh).transpose(0, 2, 1, 3).reshape(B, L, d)
        o = ctx @ P[ap + "dense.weight"].T + P[ap + "dense.bias"]
        od, lc["keep_o"] = _drop(o, ph, drop, 3 + 4 * l)
        gsp, lc["ln_a"] = layer_norm_fwd(od + x, P[ap + "LayerNorm.weight"], P[ap + "LayerNorm.bias"], cfg.eps)
        lc.update(qh=qh, kh=kh, vh=vh, a=a, ad=ad, keep_a=keep_a, ctx=ctx)
        # alpha mix                                                   bsarec.py:78
        hmix = t(cfg.alpha) * dsp + t(1 - cfg.alpha) * gsp
        # FeedForward                                                 _modules.py:61-69
        fp = p + "feed_forward."
        u = hmix @ P[fp + "dense_1.weight"].T + P[fp + "dense_1.bias"]
        g = hidden_act_fn(cfg.hidden_act)(u)
        y2 = g @ P[fp + "dense_2.weight"].T + P[fp + "dense_2.bias"]
        y2d, lc["keep_ff"] = _drop(y2, ph, drop, 4 + 4 * l)
        y, lc["ln_ff"] = layer_norm_fwd(y2d + hmix, P[fp + "LayerNorm.weight"], P[fp + "LayerNorm.bias"], cfg.eps)
        lc.update(hmix=hmix, u=u, g=g, dsp=dsp, gsp=gsp)
        cache[f"layer{l}"] = lc
        x = y
        outs.append(x)
    cache["params"] = P
    return outs, (cache if keep_cache else None)


def logits_and_loss(h_last, E, answers):
    """logits = h_last @ E^T over the whole catalogue (row 0 included), mean CE
    (src/model/bsarec.py:32-35)."""
    logits = h_last @ E.T
    m = logits.max(-1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(logits - m).sum(-1))
    B = logits.shape[0]
    nll = lse - logits[np.arange(B), answers]
    return logits, nll.mean(dtype=np.float64), lse


def bce_head(pos_ids, neg_ids):
    """SASRecModel.calculate_loss's head (src/model/sasrec.py:41-63): one positive / one negative logit at the last
    position, BCEWithLogits over the rows with pos_ids != 0 (both terms averaged over those rows)."""
    def head(h_last, E, dtype):
        t = dtype
        pos_ids_, neg_ids_ = np.asarray(pos_ids), np.asarray(neg_ids)
        keep = pos_ids_ != 0
        n = max(int(keep.sum()), 1)
        xp = (E[pos_ids_] * h_last).sum(-1).astype(np.float64)
        xn = (E[neg_ids_] * h_last).sum(-1).astype(np.float64)
        softplus = lambda z: np.maximum(z, 0) + np.log1p(np.exp(-np.abs(z)))
        loss = (softplus(-xp)[keep].sum() + softplus(xn)[keep].sum()) / n
        gp = np.where(keep, -1.0 / (1.0 + np.exp(xp)), 0.0) / n          # d loss / d xp = -sigmoid(-xp) / n
        gn = np.where(keep, 1.0 / (1.0 + np.exp(-xn)), 0.0) / n          # d loss / d xn = sigmoid(xn) / n
        gp, gn = gp.astype(t), gn.astype(t)
        dE = np.zeros_like(E)
        np.add.at(dE, pos_ids_, gp[:, None] * h_last)
        np.add.at(dE, neg_ids_, gn[:, None] * h_last)
        dh = gp[:, None] * E[pos_ids_] + gn[:, None] * E[neg_ids_]
        return float(loss), None, dE, dh.astype(t)
    return head


def fmlp_head(pos_ids, neg_ids):
    """FMLPRecModel.calculate_loss's head (src/model/fmlprec.py:41-62): mean over ALL rows of
    -log(sigmoid(x_pos) + 1e-24) - log(1 - sigmoid(x_neg) + 1e-24)."""
    def head(h_last, E, dtype):
        t = dtype
        pos_ids_, neg_ids_ = np.asarray(pos_ids), np.asarray(neg_ids)
        n = len(pos_ids_)
        xp = (E[pos_ids_] * h_last).sum(-1).astype(np.float64)
        xn = (E[neg_ids_] * h_last).sum(-1).astype(np.float64)
        sp, sn = 1.0 / (1.0 + np.exp(-xp)), 1.0 / (1.0 + np.exp(-xn))
        loss = (-np.log(sp + 1e-24) - np.log(1.0 - sn + 1e-24)).mean()
        gp = (-(sp * (1 - sp)) / (sp + 1e-24) / n).astype(t)
        gn = ((sn * (1 - sn)) / (1.0 - sn + 1e-24) / n).astype(t)
        dE = np.zeros_like(E)
        np.add.at(dE, pos_ids_, gp[:, None] * h_last)
        np.add.at(dE, neg_ids_, gn[:, None] * h_last)
        dh = gp[:, None] * E[pos_ids_] + gn[:, None] * E[neg_ids_]
        return float(loss), None, dE, dh.astype(t)
    return head


def loss_and_grads(params, cfg: Config, ids, answers, drop: Optional[DropoutSpec] = None,
                   dtype=np.float32, d_outs: Optional[List[np.ndarray]] = None, head=None):
    """BSARecModel.calculate_loss + autograd backward (src/model/bsarec.py:30-37,
    src/trainers.py:103-106), derived by hand.  Returns
    (loss, logits, grads, layer_outputs).  ``d_outs`` optionally adds upstream gradients on the
    layer outputs (used by per-op golden tests)."""
    drop = drop or DropoutSpec()
    outs, cache = forward(params, cfg, ids, drop, dtype)
    P = cache["params"]
    t = dtype
    B, L = ids.shape
    d, h, dh, N = cfg.hidden_size, cfg.num_attention_heads, cfg.head_size, cfg.num_hidden_layers
    cb = cfg.cutoff_bins
    ph, pa = cfg.hidden_dropout_prob, cfg.attention_probs_dropout_prob
    sc_h = t(1.0 / (1.0 - ph)) if ph < 1 else t(0)
    sc_a = t(1.0 / (1.0 - pa)) if pa < 1 else t(0)
    E = P["item_embeddings.weight"]
    G: Dict[str, np.ndarray] = {k: np.zeros_like(v) for k, v in P.items()}

    h_last = outs[-1][:, -1, :]
    dx = np.zeros_like(outs[-1])
    if head is None:
        logits, loss, lse = logits_and_loss(h_last, E, answers)
        dlog = np.exp(logits - lse[:, None])
        dlog[np.arange(B), answers] -= 1
        dlog = (dlog / t(B)).astype(dtype)
        G["item_embeddings.weight"] += dlog.T @ h_last                  # dense dE incl. row 0 (A.8)
        dx[:, -1, :] = dlog @ E
    else:                               # sibling models: another loss head on the same encoder (``answers`` unused)
        loss, logits, dE_head, dh_head = head(h_last, E, dtype)
        G["item_embeddings.weight"] += dE_head
        dx[:, -1, :] = dh_head
    if d_outs is not None:
        dx = dx + d_outs[-1]

    def undrop(g, keep, sc):
        return g if keep is None else np.where(keep, g * sc, t(0))

    for l in reversed(range(N)):
        p = f"item_encoder.blocks.{l}."
        lc = cache[f"layer{l}"]
        x = lc["x"]
        fp = p + "feed_forward."
        # FeedForward backward
        dz, G[fp + "LayerNorm.weight"], G[fp + "LayerNorm.bias"] = layer_norm_bwd(dx, lc["ln_ff"], P[fp + "LayerNorm.weight"])
        dy2 = undrop(dz, lc["keep_ff"], sc_h)
        G[fp + "dense_2.weight"] = dy2.reshape(-1, d).T @ lc["g"].reshape(-1, 4 * d)
        G[fp + "dense_2.bias"] = dy2.reshape(-1, d).sum(0)
        du = (dy2 @ P[fp + "dense_2.weight"]) * hidden_act_grad(cfg.hidden_act)(lc["u"])
        G[fp + "dense_1.weight"] = du.reshape(-1, 4 * d).T @ lc["hmix"].reshape(-1, d)
        G[fp + "dense_1.bias"] = du.reshape(-1, 4 * d).sum(0)
        dh_ = du @ P[fp + "dense_1.weight"] + dz
        ddsp = t(cfg.alpha) * dh_
        dgsp = t(1 - cfg.alpha) * dh_
        # attention backward
        ap = p + "layer.attention_layer."
        dza, G[ap + "LayerNorm.weight"], G[ap + "LayerNorm.bias"] = layer_norm_bwd(dgsp, lc["ln_a"], P[ap + "LayerNorm.weight"])
        do = undrop(dza, lc["keep_o"], sc_h)
        G[ap + "dense.weight"] = do.reshape(-1, d).T @ lc["ctx"].reshape(-1, d)
        G[ap + "dense.bias"] = do.reshape(-1, d).sum(0)
        dctx = (do @ P[ap + "dense.weight"]).reshape(B, L, h, dh).transpose(0, 2, 1, 3)
        dad = dctx @ lc["vh"].transpose(0, 1, 3, 2)
        dvh = lc["ad"].transpose(0, 1, 3, 2) @ dctx
        da = dad if lc["keep_a"] is None else np.where(lc["keep_a"], dad * sc_a, t(0))
        a = lc["a"]
        ds = a * (da - (da * a).sum(-1, keepdims=True)) / t(math.sqrt(dh))
        dqh = ds @ lc["kh"]
        dkh = ds.transpose(0, 1, 3, 2) @ lc["qh"]
        dq = dqh.transpose(0, 2, 1, 3).reshape(B, L, d)
        dk = dkh.transpose(0, 2, 1, 3).reshape(B, L, d)
        dv = dvh.transpose(0, 2, 1, 3).reshape(B, L, d)
        dxl = dza.copy()
        for nm, gg in (("query", dq), ("key", dk), ("value", dv)):
            G[ap + nm + ".weight"] = gg.reshape(-1, d).T @ x.reshape(-1, d)
            G[ap + nm + ".bias"] = gg.reshape(-1, d).sum(0)
            dxl += gg @ P[ap + nm + ".weight"]
        # FrequencyLayer backward: dX = b^2 G + P((1-b^2) G), dbeta = 2 b sum G (X - low)  (A.4)
        flp = p + "layer.filter_layer."
        dzf, G[flp + "LayerNorm.weight"], G[flp + "LayerNorm.bias"] = layer_norm_bwd(ddsp, lc["ln_f"], P[flp + "LayerNorm.weight"])
        df = undrop(dzf, lc["keep_f"], sc_h)
        beta = P[flp + "sqrt_beta"]
        b2 = beta ** 2
        if flp + "complex_weight" in P:
            dxf, G[flp + "complex_weight"] = fmlp_filter_bwd(df, lc["Xspec"], P[flp + "complex_weight"])
            dxl += dzf + dxf
            G[flp + "sqrt_beta"] = np.zeros_like(beta)
        else:
            dxl += dzf + b2 * df + lowpass((t(1) - b2) * df, cb)
            G[flp + "sqrt_beta"] = (t(2) * beta * (df * (x - lc["low"])).reshape(-1, d).sum(0)).reshape(1, 1, d)
        dx = dxl
        if d_outs is not None:
            dx = dx + d_outs[l]

    # embedding front-end backward
    de = undrop(dx, cache["keep0"], sc_h)
    de, G["LayerNorm.weight"], G["LayerNorm.bias"] = layer_norm_bwd(de, cache["ln0"], P["LayerNorm.weight"])
    G["position_embeddings.weight"][:L] += de.sum(0)
    flat_ids = ids.reshape(-1)
    nz = flat_ids != 0                                              # padding_idx=0: lookup grad only
    np.add.at(G["item_embeddings.weight"], flat_ids[nz], de.reshape(-1, d)[nz])
    return loss, logits, G, outs


# --------------------------------------------------------------------------------------
# Adam (torch.optim.Adam semantics; src/trainers.py:27-28,105-107)
# --------------------------------------------------------------------------------------


@dataclass
class AdamState:
    lr: float = 1e-3
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    weight_decay: float = 0.0
    t: int = 0
    m: Dict[str, np.ndarray] = field(default_factory=dict)
    v: Dict[str, np.ndarray] = field(default_factory=dict)


def adam_step(params: Dict[str, np.ndarray], grads: Dict[str, np.ndarray], st: AdamState) -> None:
    """In-place Adam update of every tensor, embedding row 0 included."""
    st.t += 1
    bc1 = 1.0 - st.beta1 ** st.t
    bc2 = 1.0 - st.beta2 ** st.t
    for k, w in params.items():
        t = w.dtype.type
        g = grads[k].astype(w.dtype)
        if st.weight_decay != 0.0:
            g = g + t(st.weight_decay) * w
        if k not in st.m:
            st.m[k] = np.zeros_like(w)
            st.v[k] = np.zeros_like(w)
        st.m[k] = t(st.beta1) * st.m[k] + t(1 - st.beta1) * g
        st.v[k] = t(st.beta2) * st.v[k] + t(1 - st.beta2) * g * g
        denom = np.sqrt(st.v[k]) / t(math.sqrt(bc2)) + t(st.eps)
        w -= t(st.lr / bc1) * (st.m[k] / denom)


# --------------------------------------------------------------------------------------
# data pipeline and evaluation (the callers either side of the hot path)
# --------------------------------------------------------------------------------------


def read_user_seqs(path: str):
    """``user item item ...`` per line, ids >= 1 (src/dataset.py:184-197)."""
    seqs = []
    max_item = 0
    with open(path) as fh:
        for line in fh:
            parts = line.strip().split(" ")
            items = [int(x) for x in parts[1:]]
            seqs.append(items)
            max_item = max(max_item, max(items))
    return seqs, max_item, len(seqs)


def left_pad(items: List[int], L: int) -> List[int]:
    """src/dataset.py:69-72."""
    items = items[-L:] if len(items) > L else items
    return [0] * (L - len(items)) + items


def train_samples(user_seqs, L: int):
    """All training prefixes (src/dataset.py:18-23,61-72): per user t = s[-(L+2):-2];
    sample i has input t[:i] left-padded (i = 0 is all padding) and answer t[i]."""
    users, inputs, answers = [], [], []
    for u, s in enumerate(user_seqs):
        t = s[-(L + 2):-2]
        for i in range(len(t)):
            users.append(u)
            inputs.append(left_pad(t[:i], L))
            answers.append(t[i])
    return (np.asarray(users, dtype=np.int64), np.asarray(inputs, dtype=np.int64).reshape(-1, L),
            np.asarray(answers, dtype=np.int64))


def eval_samples(user_seqs, L: int, split: str):
    """valid: input s[:-2], answer s[-2]; test: input s[:-1], answer s[-1]
    (src/dataset.py:24-28,61-81)."""
    cut = 2 if split == "valid" else 1
    inputs = [left_pad(s[:-cut], L) for s in user_seqs]
    answers = [s[-cut] for s in user_seqs]
    return (np.arange(len(user_seqs), dtype=np.int64), np.asarray(inputs, dtype=np.int64).reshape(-1, L),
            np.asarray(answers, dtype=np.int64))


def seen_items(user_seqs, split: str):
    """Items treated as already seen: s[:-2] for valid, s[:-1] for test (src/dataset.py:126-160)."""
    cut = 2 if split == "valid" else 1
    return [sorted(set(s[:-cut])) for s in user_seqs]


def topk_after_seen(scores: np.ndarray, seen: List[List[int]], k: int = 20) -> np.ndarray:
    """Seen items' scores are set to 0 (not -inf), then top-k by score, best first
    (src/trainers.py:134-149)."""
    scores = scores.copy()
    for r, items in enumerate(seen):
        scores[r, items] = 0
    ind = np.argpartition(scores, -k)[:, -k:]
    vals = np.take_along_axis(scores, ind, axis=1)
    order = np.argsort(vals, axis=1)[:, ::-1]
    return np.take_along_axis(ind, order, axis=1)


def hr_ndcg(answers: np.ndarray, pred: np.ndarray) -> List[float]:
    """[HR@5, NDCG@5, HR@10, NDCG@10, HR@20, NDCG@20] for single-target lists
    (src/metrics.py:3-31, src/trainers.py:70-83)."""
    out = []
    hit = pred == answers[:, None]
    for k in (5, 10, 20):
        hk = hit[:, :k]
        out.append(float(hk.any(1).mean()))
        pos = np.argmax(hk, axis=1)
        out.append(float(np.where(hk.any(1), 1.0 / np.log2(pos + 2.0), 0.0).mean()))
    return out


def train_flops_per_seq(cfg: Config) -> float:
    """Algorithmic training FLOPs per sequence, as the reference executes it:
    F_train = 3 * [N L (24 d^2 + 4 L d + 8 cb d) + 2 d V]."""
    d, L, N, V, cb = cfg.hidden_size, cfg.max_seq_length, cfg.num_hidden_layers, cfg.item_size, cfg.cutoff_bins
    return 3.0 * (N * L * (24 * d * d + 4 * L * d + 8 * cb * d) + 2 * d * V)
