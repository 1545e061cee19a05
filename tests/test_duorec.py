"""Differentiable ``forward()`` (src/model/bsarec.py:16-28 as an autograd graph) and the sibling model DuoRec built on it
(src/model/duorec.py): the HIP encoder + the restated contrastive head against golden vectors made by
importing the reference's DuoRecModel (tests/golden/make_golden_duorec.py)."""
import argparse
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_l2


def load():
    z = np.load(os.path.join(GOLDEN, "duorec_A_d64_L50_h2.npz"))
    return z, json.loads(str(z["cfg"]))


def test_duorec_state_dict_contract():
    torch = pytest.importorskip("torch")
    from bsarec_amd import DuoRecModel, MODEL_DICT
    z, cfg = load()
    a = argparse.Namespace(hidden_act="gelu", batch_size=10, c=3, **cfg)
    m = MODEL_DICT["duorec"](args=a)
    assert isinstance(m, DuoRecModel) and m.args.alpha == 0.0 and m.ssl == "us_x"
    keys = [k[2:] for k in z.files if k.startswith("p/")]
    assert list(m.state_dict().keys()) == keys           # the reference's 36 names, in its order
    m.load_state_dict({k: torch.from_numpy(z["p/" + k]) for k in keys})
    for k in keys:
        assert np.array_equal(m.state_dict()[k].numpy(), z["p/" + k]), k


def test_same_target_sampler_draws_another_sequence_with_the_same_answer():
    torch = pytest.importorskip("torch")
    from bsarec_amd import data as D
    rng = np.random.default_rng(0)
    n, L = 500, 8
    inputs = rng.integers(1, 50, size=(n, L))
    answers = rng.integers(1, 12, size=n)
    dl = D.DeviceBatches(np.arange(n), inputs, answers, 64, "cpu", shuffle=True, seed=1).enable_same_target()
    idx = torch.arange(n)
    same = dl.sample_same_target(idx)
    rows = {tuple(r): i for i, r in enumerate(inputs.tolist())}
    picked = np.array([rows[tuple(r)] for r in same.tolist()])
    assert np.array_equal(answers[picked], answers)                 # same target item
    counts = np.bincount(answers)[answers]
    assert np.all((picked != np.arange(n)) | (counts == 1))         # not the sample itself when the group has another


@pytest.mark.gpu
def test_hip_duorec_vs_reference_golden():
    """Last-layer output, loss (CE + InfoNCE, us_x), all 36 gradients, and three optimisation steps of the reference's
    loop (calculate_loss / zero_grad / backward / torch.optim.Adam.step) against the imported reference."""
    torch = pytest.importorskip("torch")
    from bsarec_amd import DuoRecModel
    z, cfg = load()
    a = argparse.Namespace(hidden_act="gelu", batch_size=10, c=3, seed=1, **cfg)
    m = DuoRecModel(a)
    keys = [k[2:] for k in z.files if k.startswith("p/")]
    m.load_state_dict({k: torch.from_numpy(z["p/" + k]) for k in keys})
    m = m.cuda()
    m.train()
    ids, sem, ans = (torch.from_numpy(z[k]).cuda() for k in ("ids", "sem", "answers"))
    out = m.forward(ids)
    assert out.requires_grad
    real = z["ids"] > 0
    assert np.abs(out.detach().cpu().numpy() - z["out_last"])[real].max() <= 2e-5
    del out                                               # drops the graph: its retained plan is released
    loss = m.calculate_loss(ids, ans, None, sem, None)
    assert abs(loss.item() - float(z["loss"])) <= 5e-6 * abs(float(z["loss"]))
    m.zero_grad()
    loss.backward()
    grads = {}
    for name, p in m.named_parameters():
        rk = DuoRecModel._ref_key(name)
        if rk is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name     # unused frequency branch
            continue
        grads[rk] = p.grad.cpu().numpy()
    for k in keys:
        if k.endswith("key.bias"):
            assert np.abs(grads[k]).max() <= 1e-6
            continue
        assert rel_l2(grads[k], z["g/" + k]) <= 1e-4, (k, rel_l2(grads[k], z["g/" + k]))
    assert len(m._slots_busy) == 0                        # every retained forward released its plan
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.0)
    losses = []
    for _ in range(3):
        l = m.calculate_loss(ids, ans, None, sem, None)
        opt.zero_grad()
        l.backward()
        opt.step()
        losses.append(l.item())
    np.testing.assert_allclose(losses, z["adam_losses"], rtol=1e-5)
    sd = m.state_dict()
    for k in keys:
        got, want = sd[k].cpu().numpy(), z["a/" + k]
        if k.endswith("key.bias"):
            continue
        bad = np.abs(got - want) > 2e-5
        assert bad.mean() <= 2e-3, (k, bad.mean(), np.abs(got - want).max())


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [1, 0])
def test_forward_is_differentiable_vs_oracle(fused):
    """BSARecModel.forward as an autograd graph: d/d(params) of sum(out * W) for a random W over ALL positions, dropout
    on (the retained forward regenerates ITS masks in the backward although another forward ran in between), against the
    oracle with the same upstream gradient -- fused block kernels and generic tiled kernels."""
    torch = pytest.importorskip("torch")
    from oracle import bsarec_oracle as O
    from bsarec_amd import BSARecModel, _lib as Lb
    old = Lb.set_default_options(no_fused=1 - fused)
    try:
        B, L, V = 9, 50, 131
        cfg = O.Config(item_size=V, hidden_size=64, max_seq_length=L, num_hidden_layers=2, num_attention_heads=2, c=5, alpha=0.7,
                       hidden_dropout_prob=0.4, attention_probs_dropout_prob=0.3)
        params = O.init_params(cfg, seed=3)
        rng = np.random.default_rng(3)
        ids = np.zeros((B, L), dtype=np.int64)
        for b in range(B):
            n = 0 if b == 0 else (L if b == 1 else int(rng.integers(1, L + 1)))
            if n:
                ids[b, L - n:] = rng.integers(1, V, size=n)
        W = rng.normal(0, 1, (B, L, 64)).astype(np.float32)
        a = argparse.Namespace(item_size=V, hidden_size=64, max_seq_length=L, batch_size=B, hidden_dropout_prob=0.4,
                               attention_probs_dropout_prob=0.3, num_hidden_layers=2, num_attention_heads=2, hidden_act="gelu",
                               initializer_range=0.02, c=5, alpha=0.7, seed=42)
        m = BSARecModel(a)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
        m = m.cuda()
        m.train()
        m.set_seed(77)
        tid = torch.from_numpy(ids).cuda()
        out = m(tid)
        step = int(m._state[1].item())
        other = m(tid)                                       # a second retained forward: new masks, its own plan
        assert not torch.equal(out, other)
        (out * torch.from_numpy(W).cuda()).sum().backward()
        null_head = lambda h_last, E, dtype: (0.0, None, np.zeros_like(E), np.zeros_like(h_last))
        d_outs = [np.zeros_like(W) for _ in range(2)] + [W]
        _, _, G, outs = O.loss_and_grads(params, cfg, ids, None, O.DropoutSpec(True, 77, step), d_outs=d_outs, head=null_head)
        real = ids > 0
        assert np.abs(out.detach().cpu().numpy() - outs[-1])[real].max() <= 2e-4
        for name, p in m.named_parameters():
            if name.endswith("key.bias"):
                continue
            assert rel_l2(p.grad.cpu().numpy(), G[name]) <= 2e-4, (name, rel_l2(p.grad.cpu().numpy(), G[name]))
        del other
    finally:
        Lb.set_default_options(**old)


@pytest.mark.gpu
def test_duorec_trains_through_the_driver():
    """`--model_type DuoRec` through the reference-flag driver on a slice of LastFM: the loss falls and the test metrics
    leave chance level.  No DuoRec log ships with the reference: a sanity band, not a known answer."""
    import logging
    torch = pytest.importorskip("torch")
    from bsarec_amd import main as M
    z = np.load(os.path.join(GOLDEN, "kat_LastFM.npz"))
    off, items = z["seq_offsets"], z["seq_items"]
    seqs = [items[off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]
    losses = []

    class Grab(logging.Handler):
        def emit(self, rec):
            m = str(rec.getMessage())
            if "rec_loss" in m:
                losses.append(float(m.split("'rec_loss': '")[1].split("'")[0]))
    logger = logging.getLogger("duorec_test_train")
    logger.setLevel(logging.INFO)
    logger.addHandler(Grab())
    args = M.parse_args(["--data_name", "LastFM", "--model_type", "DuoRec", "--lr", "0.001", "--num_attention_heads", "2",
                         "--epochs", "8", "--patience", "8"])
    scores, info, epochs, secs = M.run(args, seqs, logger)
    assert losses[-1] < losses[0] - 0.1, losses           # CE over 3,647 classes starts at ~8.2 + the contrastive terms
    assert scores[2] > 0.01, scores                          # HR@10 above chance (0.0027)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [1, 0])
def test_all_sequence_output_list_is_one_autograd_graph(fused):
    """forward(all_sequence_output=True) (src/model/bsarec.py:46-54): every element of the list carries gradients.  With
    dropout off the gradient of  sum_l <g_l, out_l>  decomposes exactly: out_2 is the 2-layer model's output, out_1 the output
    of the SAME weights truncated to one layer, out_0 the embedding front-end (plain torch here) -- the single-output backward
    of each is already pinned by the oracle / reference goldens (bsarec_backward_seq), so the multi-output node must equal
    their sum.  fused = 1: per-sequence block kernels (the embedding gradient joins inside the bottom block's epilogue);
    fused = 0: generic tiled kernels."""
    import argparse
    torch = pytest.importorskip("torch")
    from bsarec_amd import BSARecModel, _lib as Lb
    old = Lb.set_default_options(no_fused=1 - fused)
    try:
        def ns(layers):
            return argparse.Namespace(item_size=151, hidden_size=64, max_seq_length=50, batch_size=8, hidden_dropout_prob=0.0,
                                      attention_probs_dropout_prob=0.0, num_hidden_layers=layers, num_attention_heads=2,
                                      hidden_act="gelu", initializer_range=0.02, c=5, alpha=0.7, seed=1)
        torch.manual_seed(3)
        m2 = BSARecModel(ns(2)).cuda()
        with torch.no_grad():
            for k, p in m2.named_parameters():
                if k.endswith(".bias"):
                    p.normal_(0, 0.05)
        m2.train()
        m1 = BSARecModel(ns(1)).cuda()
        sd2 = m2.state_dict()
        m1.load_state_dict({k: sd2[k] for k in m1.state_dict()})
        m1.train()
        rng = np.random.default_rng(0)
        ids = np.zeros((6, 50), dtype=np.int64)
        for b in range(6):
            n = int(rng.integers(1, 51))
            ids[b, 50 - n:] = rng.integers(1, 151, size=n)
        ids = torch.from_numpy(ids).cuda()
        g = [torch.randn(6, 50, 64, device="cuda") for _ in range(3)]

        outs = m2.forward(ids, all_sequence_output=True)
        assert len(outs) == 3 and all(o.requires_grad for o in outs)
        (sum((gi * oi).sum() for gi, oi in zip(g, outs))).backward()
        got = {k: p.grad.clone() for k, p in m2.named_parameters()}

        for p in m2.parameters():
            p.grad = None
        (g[2] * m2.forward(ids)).sum().backward()                    # the last layer alone
        want = {k: p.grad.clone() for k, p in m2.named_parameters()}
        (g[1] * m1.forward(ids)).sum().backward()                    # block 0's output = the 1-layer model's output
        for k, p in m1.named_parameters():
            want[k] += p.grad
        # embedding output (dropout off): LN(E[ids] + Pos) in torch
        E = sd2["item_embeddings.weight"].clone().requires_grad_(True)
        P = sd2["position_embeddings.weight"].clone().requires_grad_(True)
        gw = sd2["LayerNorm.weight"].clone().requires_grad_(True)
        gb = sd2["LayerNorm.bias"].clone().requires_grad_(True)
        x = E[ids] + P[None]
        mu = x.mean(-1, keepdim=True)
        xh = (x - mu) / torch.sqrt(((x - mu) ** 2).mean(-1, keepdim=True) + 1e-12)
        out0 = gw * xh + gb
        np.testing.assert_allclose(outs[0].detach().cpu().numpy(), out0.detach().cpu().numpy(), atol=2e-5)
        (g[0] * out0).sum().backward()
        E.grad[0] = 0                                                # padding_idx = 0: no lookup gradient for the padding row
        want["item_embeddings.weight"] += E.grad
        want["position_embeddings.weight"] += P.grad
        want["LayerNorm.weight"] += gw.grad
        want["LayerNorm.bias"] += gb.grad
        for k in got:
            a, b = got[k].cpu().numpy(), want[k].cpu().numpy()
            if k.endswith("key.bias"):
                assert np.abs(a).max() <= 1e-4
                continue
            assert rel_l2(a, b) <= 2e-5, (k, rel_l2(a, b))
    finally:
        Lb.set_default_options(**old)
