#!/usr/bin/env python3
"""Reproducer of the x3_products parity failure (tests/test_gpu_parity.py [6-2-50-33]): layer outputs of the full block
kernels with x3_products = 1 against the default fp32 products and the oracle, several runs in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from oracle import bsarec_oracle as O
from bsarec_amd import _lib as Lb
if os.environ.get("DBG_LIB"):        # an older build of the library (ABI 7: no bsarec_config_is_fused): bisecting
    Lb.LIB_PATH = os.environ["DBG_LIB"]
    if os.environ.get("DBG_ABI", "7") == "7":
        Lb.ABI_VERSION = 7
        Lb.EXPORTS.pop("bsarec_config_is_fused", None)
        _l = Lb.load()
        _l.bsarec_config_is_fused = lambda cfg: 1
from test_gpu_parity import build_model
heads, L, B = 2, 50, 33
cfg = O.Config(item_size=131, hidden_size=64, max_seq_length=L, num_hidden_layers=2, num_attention_heads=heads,
               c=5, alpha=0.7, hidden_dropout_prob=0.4, attention_probs_dropout_prob=0.3)
params = O.init_params(cfg, seed=heads + L)
rng = np.random.default_rng(L)
for k in params:
    if k.endswith(".bias"):
        params[k] = (rng.standard_normal(params[k].shape) * 0.05).astype(np.float32)
    elif "LayerNorm.weight" in k:
        params[k] = (1 + rng.standard_normal(params[k].shape) * 0.1).astype(np.float32)
ids = np.zeros((B, L), dtype=np.int64)
for b in range(B):
    n = 0 if b == 0 else (L if b == 1 else int(rng.integers(1, L + 1)))
    if n:
        ids[b, L - n:] = rng.integers(1, 131, size=n)
ans = rng.integers(1, 131, size=B).astype(np.int64)
oloss, _, G, outs = O.loss_and_grads(params, cfg, ids, ans, O.DropoutSpec(True, 77, 1))
REPS = int(os.environ.get("DBG_REPS", "10"))
h = heads
def snap(plan):
    Lp = (L + 3) // 4 * 4
    d = {"dsp0": plan.view(Lb.BUF_DSP, 0, (B, L, 64)), "probs0": plan.view(Lb.BUF_PROBS, 0, (B, h, L, Lp)), "ctx0": plan.view(Lb.BUF_CTX, 0, (B, L, 64)),
         "hmix0": plan.view(Lb.BUF_HMIX, 0, (B, L, 64)), "out1": plan.view(Lb.BUF_LAYER_OUT, 1, (B, L, 64)),
         "probs1": plan.view(Lb.BUF_PROBS, 1, (B, h, L, Lp)), "ctx1": plan.view(Lb.BUF_CTX, 1, (B, L, 64)),
         "hmix1": plan.view(Lb.BUF_HMIX, 1, (B, L, 64)), "out2": plan.view(Lb.BUF_LAYER_OUT, 2, (B, L, 64))}
    return {k: v.float().cpu().numpy().copy() for k, v in d.items()}
real = ids > 0
ref = None
for x3, prune, fwd_only in ((0, 0, 1), (1, 0, 1), (1, 0, 0)):
    nbad = 0
    for rep in range(REPS):
        Lb.set_default_options(no_fused=0, no_prune_top=1 - prune, chain_kernels=0, x3_products=x3)
        model = build_model(cfg, params)
        model.train(); model.set_seed(77)
        loss = model.calculate_loss(torch.from_numpy(ids).cuda(), torch.from_numpy(ans).cuda(), None, None, None)
        if not fwd_only:
            loss.backward()
        torch.cuda.synchronize()
        S = snap(model._plan(B))
        if ref is None:
            ref = S
            continue
        msg = []
        for k in S:
            e = np.abs(S[k] - ref[k])
            if k.startswith("probs"):
                e = e * real[:, None, :, None]
            else:
                e = e * real[:, :, None]
            if e.max() > 1e-3:
                bad = np.argwhere(e > 1e-3)
                msg.append(f"{k}: max {e.max():.3f} seqs {sorted(set(bad[:, 0].tolist()))[:5]}")
        if msg:
            nbad += 1
            print(f"  x3={x3} fwd_only={fwd_only} rep {rep}:", " | ".join(msg), flush=True)
    print(f"x3={x3} prune={prune} fwd_only={fwd_only}: {nbad}/{REPS} runs differ from the fp32 run", flush=True)
