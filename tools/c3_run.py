#!/usr/bin/env python3
"""Eager training steps at the C3 shape (generic tiled kernels) for rocprofv3 --kernel-trace --stats:
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/c3_run.py [steps] [f32|bf16]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bsarec_amd import BSARecModel, data as D
from bsarec_amd.trainer import Trainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
a3 = argparse.Namespace(item_size=3417, hidden=256, seq_len=200, batch=256, layers=4, heads=4, dtype=dtype)
m3 = bench.model_args(a3)
torch.manual_seed(42)
dev = torch.device("cuda", 0)
model = BSARecModel(m3).to(dev); model.set_seed(42, 0); model.train()
seqs = D.synth_ml1m_like(seed=42, n_items=3416)
u, x, y = D.train_table(seqs[:600], 200)
bt = D.DeviceBatches(u, x, y, 256, dev, shuffle=True, seed=42)
tr = Trainer(model, bt, None, None, m3, None, use_graph=False)
perm = bt.local_permutation(); pbuf = perm.clone(); cur = torch.zeros(1, dtype=torch.int64, device=dev)
import time
loss = tr.indexed_step(bt, pbuf, cur, None)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = tr.indexed_step(bt, pbuf, cur, None)
torch.cuda.synchronize()
print(dtype, "loss", float(loss.item()), "ms/step (eager)", round(1e3 * (time.perf_counter() - t0) / steps, 3))
