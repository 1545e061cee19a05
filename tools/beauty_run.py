#!/usr/bin/env python3
"""The C2 workload at Beauty's own shape (V = 12,102, 1 head, c = 5, bf16 storage; bench.secondary_beauty_bf16) for
rocprofv3 --kernel-trace --stats:   rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/beauty_run.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bsarec_amd import BSARecModel, data as D
from bsarec_amd.trainer import Trainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 48
print(bench.secondary_beauty_bf16(torch.device("cuda", 0), D, BSARecModel, Trainer, steps=steps, warmup=8))
