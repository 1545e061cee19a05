#!/usr/bin/env python3
"""Per-phase shader-clock shares of the fused kernels (workgroup 0), C1 shape."""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bsarec_amd import BSARecModel, _lib as Lb
import bench
a = argparse.Namespace(item_size=3417, hidden=64, seq_len=50, batch=256, layers=2, heads=2)
m = BSARecModel(bench.model_args(a)).cuda(); m.train(); m.configure_adam()
ids = torch.randint(1, 3417, (256, 50), device="cuda"); ids[:, :20] = 0
ans = torch.randint(1, 3417, (256,), device="cuda")
buf = torch.zeros(32 * 4, dtype=torch.int64, device="cuda")
lib = Lb.load()
for _ in range(3): m.train_step(ids, ans)
plan = m._plan(256)
lib.bsarec_debug_stamps(plan.handle, buf.data_ptr())
m.train_step(ids, ans); torch.cuda.synchronize()
lib.bsarec_debug_stamps(plan.handle, None)
s = buf.cpu().numpy().reshape(4, 32)
names = {0: ["load", "freq", "qkv", "attn", "dense+ln", "ffn1", "ffn2", "ln_ff"],
         1: ["ln_ff_bwd", "dU", "dH", "ln_a/f_bwd", "dC", "attn_bwd", "dXqkv", "freq_bwd"]}
for k in range(2):                      # the pruned top block (layer 1): raw stamp deltas per barrier step
    row = s[2 + k]
    n = 16
    vals = [int(v) for v in row[:n]]
    nz = sorted((i for i, v in enumerate(vals) if v), key=lambda i: vals[i])
    print(f"layer 1 {'top_bwd' if k else 'top_fwd'} stamps:", " ".join(f"{i}:{vals[i] - vals[nz[0]]}" for i in nz))
for l in range(1):
    for k in range(2):
        row = s[2 * l + k].copy()
        if k == 0:
            row[2] = row[1]              # the forward has no stamp between the FrequencyLayer || QKV phases
        d = np.diff(row[:9])
        print(f"layer {l} {'bwd' if k else 'fwd'} total {row[8]-row[0]} cyc:", " ".join(f"{n}={int(x)}" for n, x in zip(names[k], d)))
print("bwd launch: kernel entry -> head start", int(s[3][0] - s[1][0]), " head", int(s[3][15] - s[3][0]), " head end -> stage A1 end", int(s[1][1] - s[3][15]))
print("fwd launch: block end (stamp 8) -> tail start", int(s[2][1] - s[0][8]), " tail", int(s[2][15] - s[2][1]))
