"""The bf16 half of config C3 ("C3 (fp32 and bf16)"): cfg.storage = 1 OUTSIDE the fused shape class.

There the generic tiled kernels keep every tensor fp32 in HBM and round the operands of every matrix product of the block
stack to bf16 while they are staged into LDS (csrc/gemm.h, BF = true: v_mfma_f32_32x32x16_bf16, fp32 accumulation); the
loss head (logits, cross-entropy, logits backward), LayerNorm, softmax, the FrequencyLayer and Adam stay fp32.  The oracle
is fp32 (the reference cannot run bf16 on CPU), so the mode is judged at the bf16 gates of tests/test_gpu_bf16.py:
logits <= 5e-3 rel-Linf, loss <= 5e-4 rel, gradients <= 2e-2 rel-L2, layer outputs <= 3e-2 abs (of max(1, |ref|max))."""
import numpy as np
import pytest

from conftest import load_e2e, rel_l2
from test_gpu_bf16 import GRAD_GATE, LOGITS_GATE, LOSS_GATE, OUT_GATE, build, grad_errors

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("name", ["D_d128_L200_h4", "C_d32_L12_h4", "B_d16_L20_h1"])
def test_bf16_products_generic_path_vs_reference_golden(name):
    """The reference's own fp32 vectors (imported-reference goldens) at three generic shapes: every layer output, the
    logits, the loss and all gradients through the bf16-product kernels."""
    from bsarec_amd import _lib as Lb
    cfg, params, grads, _, z = load_e2e(name)
    model = build(cfg, params)
    model.train()
    ids = torch.from_numpy(z["ids"]).cuda()
    B, L = z["ids"].shape
    outs = [o.detach() for o in model.forward(ids, all_sequence_output=True)]
    plan = model._plan(B)
    assert plan.options["storage"] == 1 and not plan.lib.bsarec_plan_is_fused(plan.handle)
    assert plan.view(Lb.BUF_HMIX, 0, (B, L, cfg.hidden_size)).dtype == torch.float32      # tensors stay fp32 in this mode
    real = z["ids"] > 0
    for l, o in enumerate(outs):
        r = z[f"out/{l}"]
        err = np.abs(o.cpu().numpy() - r)
        assert err[real].max() <= OUT_GATE * max(1.0, np.abs(r).max()), (l, err[real].max())
    loss = model.calculate_loss(ids, torch.from_numpy(z["answers"]).cuda(), None, None, None)
    loss.backward()
    logits = plan.view(Lb.BUF_LOGITS, 0, (B, (cfg.item_size + 3) // 4 * 4))[:, :cfg.item_size].cpu().numpy()
    lerr = np.abs(logits - z["logits"]).max() / np.abs(z["logits"]).max()
    loss_err = abs(loss.item() - float(z["loss"])) / abs(float(z["loss"]))
    ge = grad_errors(model, grads)
    print(f"bf16 products {name}: logits rel-Linf {lerr:.2e}, loss rel {loss_err:.2e}, worst grad rel-L2 "
          f"{max(ge.values()):.2e} ({max(ge, key=ge.get)}), median {np.median(list(ge.values())):.2e}")
    assert lerr <= LOGITS_GATE and loss_err <= LOSS_GATE
    assert max(ge.values()) <= GRAD_GATE, {k: v for k, v in ge.items() if v > GRAD_GATE}
    # and the mode really multiplies in bf16: an fp32 plan of the same model agrees with the goldens ~100x better
    fp32 = build(cfg, params, storage=None)
    fp32.train()
    o32 = fp32.forward(ids, all_sequence_output=True)[-1].detach().cpu().numpy()
    obf = outs[-1].cpu().numpy()
    r = z[f"out/{len(outs) - 1}"]
    assert np.abs(o32 - r)[real].max() * 4 < np.abs(obf - r)[real].max(), "bf16-product plan is as exact as fp32: mode not taken"


def test_bf16_products_config3_shape_training_step_vs_fp32_oracle():
    """BASELINE config 3's shape (L = 200, hidden = 256, 4 heads, 4 layers), dropout ON (the Philox masks do not depend on
    the product type), ragged batch incl. an empty sequence: loss and all gradients vs the fp32 oracle."""
    from oracle import bsarec_oracle as O
    cfg = O.Config(item_size=301, hidden_size=256, max_seq_length=200, num_hidden_layers=4, num_attention_heads=4, c=9,
                   alpha=0.7, hidden_dropout_prob=0.3, attention_probs_dropout_prob=0.2)
    params = O.init_params(cfg, seed=3)
    rng = np.random.default_rng(3)
    for k in params:
        if k.endswith(".bias"):
            params[k] = rng.normal(0, 0.05, params[k].shape).astype(np.float32)
    B, L = 3, 200
    ids = np.zeros((B, L), dtype=np.int64)
    for b, n in enumerate((200, 37, 0)):
        if n:
            ids[b, L - n:] = rng.integers(1, 301, size=n)
    ans = rng.integers(1, 301, size=B).astype(np.int64)
    model = build(cfg, params)
    model.train()
    model.set_seed(11)
    loss = model.calculate_loss(torch.from_numpy(ids).cuda(), torch.from_numpy(ans).cuda(), None, None, None)
    loss.backward()
    step = int(model._state[1].item())
    oloss, _, G, _ = O.loss_and_grads(params, cfg, ids, ans, O.DropoutSpec(True, 11, step))
    loss_err = abs(loss.item() - oloss) / abs(oloss)
    ge = grad_errors(model, G)
    print(f"bf16 products C3 shape: loss rel {loss_err:.2e}, worst grad rel-L2 {max(ge.values()):.2e} "
          f"({max(ge, key=ge.get)}), median {np.median(list(ge.values())):.2e}")
    assert loss_err <= LOSS_GATE
    assert max(ge.values()) <= GRAD_GATE, {k: v for k, v in ge.items() if v > GRAD_GATE}


def test_bf16_products_training_tracks_fp32_training():
    """Three epochs of Trainer.iteration at a generic shape (d = 128, L = 64, 4 heads) in both arithmetic modes from the same
    initial weights, dropout 0: the bf16-product run's epoch losses follow the fp32 run's to 1 %, and it learns."""
    import argparse
    from bsarec_amd import BSARecModel, data as D
    from bsarec_amd.trainer import Trainer
    seqs = D.synth_ml1m_like(seed=4, n_users=80, n_items=400)
    u, x, a_ = D.train_table(seqs, 64)
    u, x, a_ = u[:1024], x[:1024], a_[:1024]
    res = {}
    for storage in (None, "bf16"):
        a = argparse.Namespace(item_size=401, hidden_size=128, max_seq_length=64, batch_size=256, hidden_dropout_prob=0.0,
                               attention_probs_dropout_prob=0.0, num_hidden_layers=2, num_attention_heads=4, hidden_act="gelu",
                               initializer_range=0.02, c=9, alpha=0.7, seed=42, lr=1e-3, adam_beta1=0.9, adam_beta2=0.999,
                               weight_decay=0.0, no_cuda=False, log_freq=1, storage=storage)
        torch.manual_seed(2)
        model = BSARecModel(a).cuda()
        dl = D.DeviceBatches(u, x, a_, 256, "cuda", shuffle=True, seed=9)
        tr = Trainer(model, dl, None, None, a, None)
        res[storage] = [float(tr.train(e)["rec_loss"]) for e in range(3)]
    f, b = res[None], res["bf16"]
    print("fp32 epochs", f, "bf16-product epochs", b)
    assert b[-1] < b[0]
    assert all(abs(x - y) <= 1e-2 * abs(x) for x, y in zip(f, b)), (f, b)


@pytest.mark.parametrize("act", ["relu", "swish", "tanh", "sigmoid"])
def test_bf16_products_non_default_hidden_act_vs_reference_golden(act):
    """hidden_act != gelu always takes the generic tiled kernels (the activation rides in the operand transform of dense_2 and in
    the epilogue of its backward): with storage = bf16 those products run on the bf16 matrix cores too -- the reference's
    goldens for relu / swish / tanh / sigmoid (tests/golden/acts_*.npz) at the bf16 gates.  relu gets twice the gradient gate:
    its derivative is a step, so a pre-activation that bf16 products move across 0 flips a whole gradient term (measured on this
    6-sequence fixture: dense_1.weight 2.1e-2 rel-L2; the smooth activations stay below 1e-2)."""
    import argparse
    from test_hidden_act import load
    from bsarec_amd import BSARecModel
    z, cfg = load(act)
    a = argparse.Namespace(batch_size=6, seed=1, storage="bf16", **cfg)
    m = BSARecModel(a)
    m.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p/")})
    m = m.cuda()
    m.train()
    ids, ans = torch.from_numpy(z["ids"]).cuda(), torch.from_numpy(z["answers"]).cuda()
    with torch.no_grad():
        out = m.forward(ids).cpu().numpy()
    real = z["ids"] > 0
    assert np.abs(out - z["out_last"])[real].max() <= OUT_GATE * max(1.0, np.abs(z["out_last"]).max())
    loss = m.calculate_loss(ids, ans, None, None, None)
    assert abs(loss.item() - float(z["loss"])) <= LOSS_GATE * abs(float(z["loss"]))
    loss.backward()
    plan = m._plan(ids.shape[0])
    assert plan.options["storage"] == 1 and not plan.lib.bsarec_plan_is_fused(plan.handle)
    worst, gate = 0.0, (2 * GRAD_GATE if act == "relu" else GRAD_GATE)
    for k, g in m.grad_views().items():
        if k.endswith("key.bias"):
            continue
        worst = max(worst, rel_l2(g.cpu().numpy(), z["g/" + k]))
        assert rel_l2(g.cpu().numpy(), z["g/" + k]) <= gate, (k, rel_l2(g.cpu().numpy(), z["g/" + k]))
    print(f"bf16 products, hidden_act {act}: worst grad rel-L2 {worst:.2e}")
