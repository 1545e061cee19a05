#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by IMPORTING the reference.

Runs only in the build container (``/root/reference`` does not exist on the GPU box and
the reference never travels).  What is committed is data: inputs, the reference's outputs,
the weights of its two shipped checkpoints and its dataset files restated as integer
arrays.  Re-run with ``python tests/golden/make_golden.py`` from the repo root.

Fixtures written:
  e2e_*.npz     end-to-end: state_dict, ids, answers -> all layer outputs, logits, loss,
                all 42 grads, parameters after 3 Adam steps (dropout p = 0, train mode)
  freq_ops.npz  FrequencyLayer forward/backward for several (L, c)
  mask_ops.npz  get_attention_mask on mixed-padding ids
  kat_*.npz     shipped checkpoint + dataset -> the six test metrics, top-20 lists and
                last-position logits of the first 8 (top-20: first 64) test users (src/output/*_best.log)
  data_facts.json  sample counts / first+last samples of each split for LastFM
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

from model.bsarec import BSARecModel, FrequencyLayer  # noqa: E402
from dataset import RecDataset, get_user_seqs, get_rating_matrix  # noqa: E402
from trainers import Trainer  # noqa: E402


def mk_args(**kw):
    a = argparse.Namespace(
        item_size=97, hidden_size=64, max_seq_length=50, batch_size=256,
        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, num_hidden_layers=2,
        num_attention_heads=2, hidden_act="gelu", initializer_range=0.02, c=3, alpha=0.9,
        model_type="BSARec", no_cuda=True, lr=1e-3, adam_beta1=0.9, adam_beta2=0.999,
        weight_decay=0.0, log_freq=1)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def mixed_ids(rng, B, L, V):
    """Left-padded id rows: row 0 all padding, row 1 no padding, the rest random lengths."""
    ids = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        n = 0 if b == 0 else (L if b == 1 else int(rng.integers(1, L + 1)))
        if n:
            ids[b, L - n:] = rng.integers(1, V, size=n)
    return ids


def e2e_case(name, seed, B, **kw):
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    args = mk_args(**kw)
    model = BSARecModel(args)
    # make biases / LN params non-trivial so their gradients paths are exercised
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith(".bias"):
                p.normal_(0.0, 0.05)
            elif "LayerNorm.weight" in n:
                p.add_(torch.randn_like(p) * 0.1)
    model.train()
    ids = mixed_ids(rng, B, args.max_seq_length, args.item_size)
    answers = rng.integers(1, args.item_size, size=B).astype(np.int64)
    tid, tans = torch.from_numpy(ids), torch.from_numpy(answers)
    out = {"cfg": json.dumps({k: getattr(args, k) for k in (
        "item_size", "hidden_size", "max_seq_length", "num_hidden_layers", "num_attention_heads",
        "c", "alpha", "hidden_dropout_prob", "attention_probs_dropout_prob", "initializer_range")}),
        "ids": ids, "answers": answers}
    for n, p in model.state_dict().items():
        out["p/" + n] = p.detach().numpy().copy()
    layers = model.forward(tid, all_sequence_output=True)
    for i, t in enumerate(layers):
        out[f"out/{i}"] = t.detach().numpy().copy()
    seq = layers[-1][:, -1, :]
    logits = torch.matmul(seq, model.item_embeddings.weight.transpose(0, 1))
    out["logits"] = logits.detach().numpy().copy()
    loss = model.calculate_loss(tid, tans, None, None, None)
    out["loss"] = np.float64(loss.item())
    model.zero_grad()
    loss.backward()
    for n, p in model.named_parameters():
        out["g/" + n] = p.grad.detach().numpy().copy()
    if kw.get("adam_steps", 3):
        opt = torch.optim.Adam(model.parameters(), lr=args.lr, betas=(args.adam_beta1, args.adam_beta2),
                               weight_decay=args.weight_decay)
        losses = []
        for _ in range(kw.get("adam_steps", 3)):
            l = model.calculate_loss(tid, tans, None, None, None)
            opt.zero_grad()
            l.backward()
            opt.step()
            losses.append(l.item())
        out["adam_losses"] = np.asarray(losses, dtype=np.float64)
        for n, p in model.state_dict().items():
            out["a/" + n] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, f"e2e_{name}.npz"), **out)
    print("wrote", name, "loss", out["loss"])


def freq_ops():
    out = {}
    combos = [(50, 5), (50, 3), (20, 9), (50, 49), (50, 50), (50, 51), (51, 5), (7, 100)]
    out["combos"] = np.asarray(combos, dtype=np.int64)
    for i, (L, c) in enumerate(combos):
        torch.manual_seed(100 + i)
        args = mk_args(max_seq_length=L, hidden_size=8, c=c)
        fl = FrequencyLayer(args)
        with torch.no_grad():
            fl.LayerNorm.weight.add_(torch.randn(8) * 0.1)
            fl.LayerNorm.bias.normal_(0, 0.1)
        fl.train()
        x = torch.randn(3, L, 8, requires_grad=True)
        gy = torch.randn(3, L, 8)
        y = fl(x)
        y.backward(gy)
        out[f"{i}/x"] = x.detach().numpy()
        out[f"{i}/gy"] = gy.numpy()
        out[f"{i}/y"] = y.detach().numpy()
        out[f"{i}/dx"] = x.grad.numpy()
        out[f"{i}/sqrt_beta"] = fl.sqrt_beta.detach().numpy()
        out[f"{i}/dsqrt_beta"] = fl.sqrt_beta.grad.numpy()
        out[f"{i}/ln_w"] = fl.LayerNorm.weight.detach().numpy()
        out[f"{i}/ln_b"] = fl.LayerNorm.bias.detach().numpy()
        out[f"{i}/dln_w"] = fl.LayerNorm.weight.grad.numpy()
        out[f"{i}/dln_b"] = fl.LayerNorm.bias.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "freq_ops.npz"), **out)
    print("wrote freq_ops")


def mask_ops():
    rng = np.random.default_rng(7)
    args = mk_args(max_seq_length=12, hidden_size=8, num_attention_heads=1)
    model = BSARecModel(args)
    ids = mixed_ids(rng, 5, 12, 97)
    m = model.get_attention_mask(torch.from_numpy(ids)).numpy()
    np.savez_compressed(os.path.join(HERE, "mask_ops.npz"), ids=ids, mask=m)
    print("wrote mask_ops")


class _Log:
    def info(self, *a, **k):
        pass


def kat(data_name, heads, c, alpha):
    """Known-answer test: shipped checkpoint through the reference's own Trainer.test."""
    data_file = os.path.join(REF, "data", data_name + ".txt")
    user_seq, max_item, num_users = get_user_seqs(data_file)
    args = mk_args(item_size=max_item + 1, num_attention_heads=heads, c=c, alpha=alpha,
                   hidden_dropout_prob=0.5, attention_probs_dropout_prob=0.5, num_workers=0,
                   data_name=data_name, num_users=num_users + 1)
    seq_dic = {"user_seq": user_seq, "num_users": num_users}
    model = BSARecModel(args)
    sd = torch.load(os.path.join(REF, "output", f"BSARec_{data_name}_best.pt"), map_location="cpu",
                    weights_only=True)
    model.load_state_dict(sd)
    test_ds = RecDataset(args, user_seq, data_type="test")
    test_dl = torch.utils.data.DataLoader(test_ds, batch_size=256, shuffle=False, num_workers=0)
    args.valid_rating_matrix, args.test_rating_matrix = get_rating_matrix(data_name, seq_dic, max_item)
    tr = Trainer(model, None, None, test_dl, args, _Log())
    with torch.no_grad():
        scores, _ = tr.test(0)
    print(data_name, "metrics", scores)
    # first 64 users: logits + top-20 through the same code path, captured by hand
    model.eval()
    ids64 = torch.stack([test_ds[i][1] for i in range(64)])
    with torch.no_grad():
        h = model.predict(ids64, None)[:, -1, :]
        logits = tr.predict_full(h).numpy().copy()
    rp = logits.copy()
    rp[args.test_rating_matrix[np.arange(64)].toarray() > 0] = 0
    ind = np.argpartition(rp, -20)[:, -20:]
    arr = rp[np.arange(64)[:, None], ind]
    top = ind[np.arange(64)[:, None], np.argsort(arr)[np.arange(64), ::-1]]
    out = {"cfg": json.dumps({"item_size": max_item + 1, "hidden_size": 64, "max_seq_length": 50,
                              "num_hidden_layers": 2, "num_attention_heads": heads, "c": c, "alpha": alpha,
                              "hidden_dropout_prob": 0.5, "attention_probs_dropout_prob": 0.5,
                              "initializer_range": 0.02}),
           "metrics": np.asarray(scores, dtype=np.float64), "logits8": logits[:8].copy(), "top20_64": top.astype(np.int32),
           "seq_items": np.concatenate([np.asarray(s, dtype=np.int32) for s in user_seq]),
           "seq_offsets": np.cumsum([0] + [len(s) for s in user_seq]).astype(np.int64)}
    for n, p in sd.items():
        out["p/" + n] = p.numpy()
    np.savez_compressed(os.path.join(HERE, f"kat_{data_name}.npz"), **out)
    return user_seq, args


def data_facts(user_seq, args):
    facts = {}
    for split in ("train", "valid", "test"):
        ds = RecDataset(args, user_seq, data_type=split)
        n = len(ds)

        def samp(i):
            t = ds[i]
            return {"user": int(t[0]), "input_ids": t[1].tolist(), "answer": int(t[2])}
        facts[split] = {"n": n, "first": [samp(i) for i in range(3)], "last": [samp(n - 3 + i) for i in range(3)]}
    facts["train"]["batches_at_256"] = (facts["train"]["n"] + 255) // 256
    facts["valid_nnz"] = int(args.valid_rating_matrix.nnz)
    facts["test_nnz"] = int(args.test_rating_matrix.nnz)
    with open(os.path.join(HERE, "data_facts.json"), "w") as fh:
        json.dump(facts, fh)
    print("wrote data_facts", facts["train"]["n"], facts["train"]["batches_at_256"])


if __name__ == "__main__":
    torch.set_num_threads(8)
    e2e_case("A_d64_L50_h2", 1, 8, item_size=97, hidden_size=64, max_seq_length=50, num_hidden_layers=2,
             num_attention_heads=2, c=3, alpha=0.9)
    e2e_case("B_d16_L20_h1", 2, 6, item_size=53, hidden_size=16, max_seq_length=20, num_hidden_layers=2,
             num_attention_heads=1, c=5, alpha=0.7)
    e2e_case("C_d32_L12_h4", 3, 5, item_size=61, hidden_size=32, max_seq_length=12, num_hidden_layers=1,
             num_attention_heads=4, c=9, alpha=0.5)
    e2e_case("D_d128_L200_h4", 4, 3, item_size=101, hidden_size=128, max_seq_length=200, num_hidden_layers=1,
             num_attention_heads=4, c=9, alpha=0.7, adam_steps=0)
    freq_ops()
    mask_ops()
    us, a = kat("LastFM", 1, 3, 0.9)
    data_facts(us, a)
    kat("Beauty", 1, 5, 0.7)
