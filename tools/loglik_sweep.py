# -*- coding: utf-8 -*-
"""Likelihood-kernel rate over model shapes: python tools/loglik_sweep.py S:M:D[:dtype] ...  (1 M frames)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "speech-recognition_amd"))
from sr.recognition import _hip

N = int(os.environ.get("SWEEP_FRAMES", 1000000))
ctx = _hip.default_context(0)
rng = np.random.default_rng(0)
for spec in sys.argv[1:]:
    parts = spec.split(":")
    S, M, D = (int(v) for v in parts[:3])
    dt = np.float32 if len(parts) > 3 and parts[3] == "f32" else np.float64
    gmm = _hip.PackedGMM(ctx, rng.normal(size=(S, M, D)), rng.uniform(0.5, 1.5, size=(S, M, D)), rng.dirichlet(np.ones(M), size=S))
    X = rng.normal(size=(N, D))
    b = _hip.Batch(ctx, feats=X, offsets=np.array([0, N], dtype=np.int64), dtype=dt)
    for _ in range(3):
        b.loglik(gmm, fetch=False)
    ctx.sync()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        b.loglik(gmm, fetch=False)
    ctx.sync()
    ms = (time.perf_counter() - t0) / K * 1e3
    flops = 2.0 * 2 * D * S * M * N
    peak = 78.6 if dt == np.float64 else 157.3
    print("S=%d M=%d D=%d %s: %.3f ms  %.1f TF  %.1f %% of MFMA peak  | %.2f TB/s" % (
        S, M, D, dt.__name__, ms, flops / ms / 1e9, flops / ms / 1e9 / peak * 100, (D + S) * dt().itemsize * N / ms / 1e9), flush=True)
    b.close(); gmm.close()
