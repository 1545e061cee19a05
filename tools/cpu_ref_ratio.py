#!/usr/bin/env python3
"""How fast is the CPU oracle (oracle/bsarec_oracle.py, the `cpu_baseline.kind = "port"` of bench.py) next to the
reference's own CPU path?  Times whole C1-shape training steps (B=256, L=50, d=64, 2 layers, 2 heads, V=3417, dropout
0.5, Adam) of (a) the IMPORTED PyTorch reference (/root/reference/src, src/trainers.py:103-107) and (b) the numpy port,
on the same host, same thread budget.  Build container only: the reference does not exist on the GPU box.

    python tools/cpu_ref_ratio.py        # -> profiles/cpu_ref_ratio.json (read by bench.py: cpu_baseline.ref_ratio)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")

from model.bsarec import BSARecModel as RefModel  # noqa: E402
from oracle import bsarec_oracle as O  # noqa: E402

B, L, V = 256, 50, 3417
rng = np.random.default_rng(0)
ids = rng.integers(1, V, size=(B, L))
for b in range(B):
    ids[b, :rng.integers(0, L)] = 0
ans = rng.integers(1, V, size=B)

# (a) the reference
torch.set_num_threads(os.cpu_count())
args = argparse.Namespace(item_size=V, hidden_size=64, max_seq_length=L, batch_size=B, hidden_dropout_prob=0.5,
                          attention_probs_dropout_prob=0.5, num_hidden_layers=2, num_attention_heads=2, hidden_act="gelu",
                          initializer_range=0.02, c=3, alpha=0.9)
torch.manual_seed(0)
model = RefModel(args)
model.train()
optim = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.0)
tid, tans = torch.from_numpy(ids), torch.from_numpy(ans)


def ref_step():
    loss = model.calculate_loss(tid, tans, None, None, None)
    optim.zero_grad()
    loss.backward()
    optim.step()
    return loss.item()


for _ in range(3):
    ref_step()
n, t0 = 0, time.time()
while time.time() - t0 < 15.0:
    ref_step()
    n += 1
ref_rate = n * B / (time.time() - t0)

# (b) the port
cfg = O.Config(item_size=V, hidden_size=64, max_seq_length=L, num_hidden_layers=2, num_attention_heads=2, c=3, alpha=0.9)
P = O.init_params(cfg, 0)
st = O.AdamState()
_, _, G, _ = O.loss_and_grads(P, cfg, ids, ans, O.DropoutSpec(True, 1, 1))
O.adam_step(P, G, st)
m, t0 = 0, time.time()
while time.time() - t0 < 15.0:
    _, _, G, _ = O.loss_and_grads(P, cfg, ids, ans, O.DropoutSpec(True, 1, m + 2))
    O.adam_step(P, G, st)
    m += 1
port_rate = m * B / (time.time() - t0)

out = {"cores": os.cpu_count(), "shape": "C1: B=256 L=50 d=64 N=2 h=2 V=3417 dropout 0.5 Adam",
       "reference_seq_per_s": round(ref_rate, 1), "reference_steps": n, "port_seq_per_s": round(port_rate, 1), "port_steps": m,
       "port_over_reference_throughput": round(port_rate / ref_rate, 4),
       "torch": torch.__version__, "numpy": np.__version__,
       "note": "build container (no GPU); the imported reference is src/model/bsarec.py + torch.optim.Adam as src/trainers.py:103-107 drives them"}
with open(os.path.join(ROOT, "profiles", "cpu_ref_ratio.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out))
