#!/usr/bin/env python3
"""Golden fixture for the DuoRec sibling model, made by IMPORTING the reference
(src/model/duorec.py).  Build container only; what is committed is data: weights, ids, answers, same-target rows ->
the last layer's output, the loss (CE + InfoNCE, ssl = us_x), every gradient, parameters after 3 Adam steps.
Dropout p = 0 (the reference draws masks from torch's CPU generator, which no other implementation reproduces): the
"dropout view" of us_x then equals the input view, the contrastive term is between it and the same-target view.

    python tests/golden/make_golden_duorec.py          # -> tests/golden/duorec_A_d64_L50_h2.npz
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

from model.duorec import DuoRecModel  # noqa: E402
from make_golden import mk_args, mixed_ids  # noqa: E402


def case(name, seed, B, **kw):
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    args = mk_args(model_type="DuoRec", batch_size=B, tau=1.0, lmd=0.1, lmd_sem=0.1, ssl="us_x", sim="dot", **kw)
    model = DuoRecModel(args)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith(".bias"):
                p.normal_(0.0, 0.05)
            elif "LayerNorm.weight" in n:
                p.add_(torch.randn_like(p) * 0.1)
    model.train()
    ids = mixed_ids(rng, B, args.max_seq_length, args.item_size)
    sem = mixed_ids(rng, B, args.max_seq_length, args.item_size)[::-1].copy()
    ans = rng.integers(1, args.item_size, size=B).astype(np.int64)
    tid, tsem, tans = torch.from_numpy(ids), torch.from_numpy(sem), torch.from_numpy(ans)
    out = {"cfg": json.dumps({k: getattr(args, k) for k in (
        "item_size", "hidden_size", "max_seq_length", "num_hidden_layers", "num_attention_heads",
        "hidden_dropout_prob", "attention_probs_dropout_prob", "initializer_range", "tau", "lmd", "lmd_sem", "ssl", "sim")}),
        "ids": ids, "sem": sem, "answers": ans}
    for n, p in model.state_dict().items():
        out["p/" + n] = p.detach().numpy().copy()
    out["out_last"] = model.forward(tid).detach().numpy().copy()
    loss = model.calculate_loss(tid, tans, None, tsem, None)
    out["loss"] = np.float64(loss.item())
    model.zero_grad()
    loss.backward()
    for n, p in model.named_parameters():
        out["g/" + n] = p.grad.detach().numpy().copy()
    opt = torch.optim.Adam(model.parameters(), lr=args.lr, betas=(args.adam_beta1, args.adam_beta2),
                           weight_decay=args.weight_decay)
    losses = []
    for _ in range(3):
        l = model.calculate_loss(tid, tans, None, tsem, None)
        opt.zero_grad()
        l.backward()
        opt.step()
        losses.append(l.item())
    out["adam_losses"] = np.asarray(losses, dtype=np.float64)
    for n, p in model.state_dict().items():
        out["a/" + n] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, f"duorec_{name}.npz"), **out)
    print("wrote", name, "loss", out["loss"], "keys", len([k for k in out if k.startswith('p/')]))


if __name__ == "__main__":
    case("A_d64_L50_h2", 31, 10, item_size=97)
