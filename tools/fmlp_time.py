#!/usr/bin/env python3
"""Step time of the sibling model FMLPRec at the C1 shape: fused per-sequence kernels vs generic tiled kernels."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bsarec_amd import MODEL_DICT, _lib as Lb
import bench
for fused in (1, 0):
    Lb.set_default_options(no_fused=1 - fused)
    a = bench.model_args(argparse.Namespace(item_size=3417, hidden=64, seq_len=50, batch=256, layers=2, heads=2))
    torch.manual_seed(0)
    m = MODEL_DICT["fmlprec"](args=a).cuda(); m.train(); m.configure_adam()
    g = torch.Generator(device="cuda").manual_seed(0)
    ids = torch.randint(1, 3417, (256, 50), device="cuda", generator=g); ids[:, :20] = 0
    pos = torch.randint(1, 3417, (256,), device="cuda", generator=g); neg = torch.randint(1, 3417, (256,), device="cuda", generator=g)
    for _ in range(10): m.train_step(ids, pos, neg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): loss = m.train_step(ids, pos, neg)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    print(f"FMLPRec C1 shape, {'fused' if fused else 'generic'} kernels (eager, 5 C calls per step): {dt * 1e3:.3f} ms/step = {256 / dt:,.0f} seq/s, loss {loss.item():.4f}")
