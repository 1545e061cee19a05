#!/usr/bin/env python3
"""Golden fixture for the FMLPRec sibling model (the "learnable complex filter" of the north star),
made by IMPORTING the reference (src/model/fmlprec.py).  Build container only; what is committed is data: weights,
ids, positive / negative answers -> all layer outputs, the loss, every gradient, parameters after 3 Adam steps
(dropout p = 0).

    python tests/golden/make_golden_fmlprec.py         # -> tests/golden/fmlprec_A_d64_L50.npz, fmlprec_B_d64_L21.npz
"""
import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, HERE)

from model.fmlprec import FMLPRecModel  # noqa: E402
from make_golden import mk_args, mixed_ids  # noqa: E402


def case(name, seed, B, **kw):
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    args = mk_args(model_type="FMLPRec", **kw)
    model = FMLPRecModel(args)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith(".bias"):
                p.normal_(0.0, 0.05)
            elif "LayerNorm.weight" in n:
                p.add_(torch.randn_like(p) * 0.1)
            elif n.endswith("complex_weight"):
                p.mul_(10.0)                            # 0.02 -> 0.2: make the filter's effect well above rounding
    model.train()
    ids = mixed_ids(rng, B, args.max_seq_length, args.item_size)
    pos = rng.integers(1, args.item_size, size=B).astype(np.int64)
    neg = rng.integers(1, args.item_size, size=B).astype(np.int64)
    tid, tpos, tneg = torch.from_numpy(ids), torch.from_numpy(pos), torch.from_numpy(neg)
    out = {"cfg": json.dumps({k: getattr(args, k) for k in (
        "item_size", "hidden_size", "max_seq_length", "num_hidden_layers", "num_attention_heads",
        "hidden_dropout_prob", "attention_probs_dropout_prob", "initializer_range")}),
        "ids": ids, "pos": pos, "neg": neg}
    for n, p in model.state_dict().items():
        out["p/" + n] = p.detach().numpy().copy()
    layers = model.forward(tid, all_sequence_output=True)
    for i, t in enumerate(layers):
        out[f"out/{i}"] = t.detach().numpy().copy()
    loss = model.calculate_loss(tid, tpos, tneg, None, None)
    out["loss"] = np.float64(loss.item())
    model.zero_grad()
    loss.backward()
    for n, p in model.named_parameters():
        out["g/" + n] = p.grad.detach().numpy().copy()
    opt = torch.optim.Adam(model.parameters(), lr=args.lr, betas=(args.adam_beta1, args.adam_beta2),
                           weight_decay=args.weight_decay)
    losses = []
    for _ in range(3):
        l = model.calculate_loss(tid, tpos, tneg, None, None)
        opt.zero_grad()
        l.backward()
        opt.step()
        losses.append(l.item())
    out["adam_losses"] = np.asarray(losses, dtype=np.float64)
    for n, p in model.state_dict().items():
        out["a/" + n] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, f"fmlprec_{name}.npz"), **out)
    print("wrote", name, "loss", out["loss"], [k for k in out if k.startswith("p/")][:8])


if __name__ == "__main__":
    case("A_d64_L50", 31, 10, item_size=97)
    case("B_d64_L21", 32, 7, item_size=131, max_seq_length=21)      # odd L: no Nyquist bin
