"""The reference's non-default FeedForward activations (src/model/_modules.py:38-59: relu, swish, tanh, sigmoid): the
oracle against golden vectors made by importing the reference (tests/golden/make_golden_acts.py), and the HIP path
(generic tiled kernels; the fused per-sequence class implements the default gelu) against both."""
import argparse
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_l2

ACTS = ["relu", "swish", "tanh", "sigmoid"]


def load(act):
    z = np.load(os.path.join(GOLDEN, f"acts_{act}.npz"))
    return z, json.loads(str(z["cfg"]))


@pytest.mark.parametrize("act", ACTS)
def test_oracle_hidden_act_vs_reference_golden(act):
    from oracle import bsarec_oracle as O
    z, cfg = load(act)
    c = O.Config(**cfg)
    P = {k[2:]: z[k] for k in z.files if k.startswith("p/")}
    loss, _, G, outs = O.loss_and_grads(P, c, z["ids"], z["answers"])
    assert abs(loss - float(z["loss"])) <= 2e-6 * abs(float(z["loss"]))
    real = z["ids"] > 0
    assert np.abs(outs[-1] - z["out_last"])[real].max() <= 2e-5
    for k in z.files:
        if k.startswith("g/") and not k.endswith("key.bias"):
            assert rel_l2(G[k[2:]], z[k]) <= 3e-5, (k, rel_l2(G[k[2:]], z[k]))


def test_unknown_hidden_act_raises_like_the_reference():
    torch = pytest.importorskip("torch")
    from bsarec_amd import BSARecModel
    a = argparse.Namespace(item_size=11, hidden_size=64, max_seq_length=8, batch_size=2, hidden_dropout_prob=0.0,
                           attention_probs_dropout_prob=0.0, num_hidden_layers=1, num_attention_heads=2, hidden_act="softplus",
                           initializer_range=0.02, c=3, alpha=0.9)
    with pytest.raises(KeyError):                         # ACT2FN[act] in the reference
        BSARecModel(a)


@pytest.mark.gpu
@pytest.mark.parametrize("act", ACTS)
def test_hip_hidden_act_vs_reference_golden(act):
    torch = pytest.importorskip("torch")
    from bsarec_amd import BSARecModel
    z, cfg = load(act)
    a = argparse.Namespace(batch_size=6, seed=1, **cfg)
    m = BSARecModel(a)
    m.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p/")})
    m = m.cuda()
    m.train()
    ids, ans = torch.from_numpy(z["ids"]).cuda(), torch.from_numpy(z["answers"]).cuda()
    with torch.no_grad():
        out = m.forward(ids).cpu().numpy()
    real = z["ids"] > 0
    assert np.abs(out - z["out_last"])[real].max() <= 3e-5
    loss = m.calculate_loss(ids, ans, None, None, None)
    assert abs(loss.item() - float(z["loss"])) <= 5e-6 * abs(float(z["loss"]))
    loss.backward()
    assert not m._plan(ids.shape[0]).cfg.no_fused and m._plan(ids.shape[0]).cfg.hidden_act != 0
    for k, g in m.grad_views().items():
        if k.endswith("key.bias"):
            continue
        assert rel_l2(g.cpu().numpy(), z["g/" + k]) <= 1e-4, (k, rel_l2(g.cpu().numpy(), z["g/" + k]))
