#!/usr/bin/env python3
"""C4's per-rank step (Yelp shape V = 20,034, B = 1,024, fp32; bench.secondary_c4) for
rocprofv3 --kernel-trace --stats:   rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/c4_run.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bsarec_amd import BSARecModel, data as D
from bsarec_amd.trainer import Trainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 32
print(bench.secondary_c4(torch.device("cuda", 0), D, BSARecModel, Trainer, steps=steps, warmup=8))
