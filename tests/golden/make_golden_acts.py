#!/usr/bin/env python3
"""Golden fixtures for the reference's non-default FeedForward activations (src/model/_modules.py:38-59: relu, swish,
tanh, sigmoid), made by IMPORTING the reference.  Build container only; committed: weights, ids, answers -> last layer
output, loss, every gradient (dropout p = 0).

    python tests/golden/make_golden_acts.py          # -> tests/golden/acts_{relu,swish,tanh,sigmoid}.npz
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

from model.bsarec import BSARecModel  # noqa: E402
from make_golden import mk_args, mixed_ids  # noqa: E402


def case(act, seed, B=6):
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    args = mk_args(item_size=71, hidden_size=64, max_seq_length=24, num_hidden_layers=2, num_attention_heads=2, c=5, alpha=0.7,
                   hidden_act=act)
    model = BSARecModel(args)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith(".bias"):
                p.normal_(0.0, 0.05)
            elif "LayerNorm.weight" in n:
                p.add_(torch.randn_like(p) * 0.1)
            elif "dense_1.weight" in n:
                p.mul_(8.0)                       # push pre-activations away from 0 so that the activations differ visibly
    model.train()
    ids = mixed_ids(rng, B, args.max_seq_length, args.item_size)
    answers = rng.integers(1, args.item_size, size=B).astype(np.int64)
    tid, tans = torch.from_numpy(ids), torch.from_numpy(answers)
    out = {"cfg": json.dumps({k: getattr(args, k) for k in (
        "item_size", "hidden_size", "max_seq_length", "num_hidden_layers", "num_attention_heads", "c", "alpha",
        "hidden_dropout_prob", "attention_probs_dropout_prob", "initializer_range", "hidden_act")}),
        "ids": ids, "answers": answers}
    for n, p in model.state_dict().items():
        out["p/" + n] = p.detach().numpy().copy()
    out["out_last"] = model.forward(tid).detach().numpy().copy()
    loss = model.calculate_loss(tid, tans, None, None, None)
    out["loss"] = np.float64(loss.item())
    model.zero_grad()
    loss.backward()
    for n, p in model.named_parameters():
        out["g/" + n] = p.grad.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, f"acts_{act}.npz"), **out)
    print("wrote", act, "loss", out["loss"])


if __name__ == "__main__":
    for i, act in enumerate(("relu", "swish", "tanh", "sigmoid")):
        case(act, 40 + i)
