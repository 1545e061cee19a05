import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def load_e2e(name):
    """Load an end-to-end golden fixture -> (oracle Config, params, grads, after-Adam params, raw npz)."""
    from oracle import bsarec_oracle as O
    z = np.load(os.path.join(GOLDEN, f"e2e_{name}.npz"))
    c = json.loads(str(z["cfg"]))
    cfg = O.Config(**c)
    params = {k[2:]: z[k] for k in z.files if k.startswith("p/")}
    grads = {k[2:]: z[k] for k in z.files if k.startswith("g/")}
    after = {k[2:]: z[k] for k in z.files if k.startswith("a/")}
    return cfg, params, grads, after, z


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


E2E_CASES = ["A_d64_L50_h2", "B_d16_L20_h1", "C_d32_L12_h4", "D_d128_L200_h4"]
