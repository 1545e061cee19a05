#!/usr/bin/env python3
"""Train BSARec from scratch on a dataset held in the golden fixtures (LastFM / Beauty sequences) with the
reference's hyper-parameters and report the test metrics next to the reference's logged ones."""
import json, os, sys, logging
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bsarec_amd import main as M

name = sys.argv[1] if len(sys.argv) > 1 else "LastFM"
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", f"kat_{name}.npz"))
cfg = json.loads(str(z["cfg"]))
off, items = z["seq_offsets"], z["seq_items"]
seqs = [items[off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]
lr = {"LastFM": 0.001, "Beauty": 0.0005}[name]
args = M.parse_args(["--data_name", name, "--lr", str(lr), "--num_attention_heads", str(cfg["num_attention_heads"]),
                     "--c", str(cfg["c"]), "--alpha", str(cfg["alpha"])] + sys.argv[2:])
logging.basicConfig(level=logging.INFO if os.environ.get("VERBOSE") else logging.WARNING)
scores, info, epochs, secs = M.run(args, seqs)
print(json.dumps({"dataset": name, "epochs": epochs, "train_seconds": round(secs, 2),
                  "test": [round(s, 4) for s in scores], "reference_log": [round(float(x), 4) for x in z["metrics"]]}))
