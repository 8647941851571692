# -*- coding: utf-8 -*-
"""Full likelihood matrix (configs[1] model) at growing batch sizes: ns per frame of gh_loglik -- does the kernel keep its
rate when the matrix grows from 0.8 GB to 16 GB?"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-recognition_amd"))
import bench
from sr.recognition import _hip

ctx = _hip.default_context()
wl = bench.synth_workload(1002, 10000)
W, n, M, D = wl["W"], wl["n"], wl["M"], wl["D"]
S = W * n
gmm = _hip.PackedGMM(ctx, wl["means"].reshape(S, M, D), wl["vars"].reshape(S, M, D), wl["w"].reshape(S, M))
base = _hip.Batch(ctx, feats=wl["X"], offsets=wl["off"])
for reps in (1, 2, 5, 10, 20):
    b = base.tile(reps) if reps > 1 else base
    for _ in range(3):
        b.loglik(gmm, fetch=False)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        b.loglik(gmm, fetch=False)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 5
    print("frames %9d  %.3f ms  %.3f ns per frame" % (b.N, dt * 1e3, dt * 1e9 / b.N), flush=True)
    if reps > 1:
        b.close()
