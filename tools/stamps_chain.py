#!/usr/bin/env python3
"""Shader-clock stamps of workgroup 0 of the register-chain forward kernel (fused_chain.h) + its top-block tail, C1 shape."""
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
for rep in range(3):
    buf.zero_()
    lib.bsarec_debug_stamps(plan.handle, buf.data_ptr())
    m.train_step(ids, ans); torch.cuda.synchronize()
    lib.bsarec_debug_stamps(plan.handle, None)
    s = buf.cpu().numpy().reshape(4, 32)
    for name, row in (("layer0 fwd", s[0]), ("top fwd tail", s[2]), ("layer0 bwd", s[1]), ("top bwd head", s[3])):
        vals = [int(v) for v in row[:16]]
        nz = [i for i, v in enumerate(vals) if v]
        if nz:
            print(f"{name}:", " ".join(f"{i}:{vals[i] - vals[nz[0]]}" for i in nz))
