#!/usr/bin/env python3
"""Soft-EM iterations at the configs[3] shape (64 words x 16 states x 32 mixtures, 39-dim) alone: bench.py's C4_em leg.
Run under `rocprofv3 --kernel-trace --stats` for the kernel times.  usage: time_c4_em.py [utterances]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-recognition_amd")):
    sys.path.insert(0, p)
import bench
from sr.recognition import _hip
U = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ctx = _hip.Context(0)
group = bench.Group(argparse.Namespace(comm="native", backend="nccl"), 0, 1, 0)
print(json.dumps(bench._c4_em_config(ctx, group, U), indent=1))
