#!/usr/bin/env python3
"""Time the C5 decode legs of bench.py alone (K = 7 lattice and loop grammar) at a given utterance count."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-recognition_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
torch.cuda.init()   # torch's bundled HIP runtime has to come up BEFORE the library's (system ROCm) one, not after
import bench
from sr.recognition import _hip
U = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ctx = _hip.Context(0)
print(json.dumps(bench._continuous_config(ctx, U, min(U, 5000), np.float64), indent=1))
