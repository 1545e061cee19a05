#!/usr/bin/env python3
"""A few C1 training steps (eager, the indexed step the bench replays) for rocprofv3 --pmc runs."""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bsarec_amd import BSARecModel
import bench
a = argparse.Namespace(item_size=3417, hidden=64, seq_len=50, batch=256, layers=2, heads=2, dtype=os.environ.get("BSAREC_PMC_DTYPE", "f32"))
m = BSARecModel(bench.model_args(a)).cuda(); m.train(); m.configure_adam()
g = torch.Generator(device="cuda"); g.manual_seed(0)
n = 4096
ids = torch.randint(1, 3417, (n, 50), device="cuda", generator=g)
lens = torch.randint(0, 51, (n,), device="cuda", generator=g)
ids[torch.arange(50, device="cuda")[None, :] < (50 - lens)[:, None]] = 0
ans = torch.randint(1, 3417, (n,), device="cuda", generator=g)
perm = torch.randperm(n, device="cuda", generator=g)
cursor = torch.zeros(1, dtype=torch.int64, device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    m.train_step_indexed(ids, ans, perm, cursor, 256)
torch.cuda.synchronize()
