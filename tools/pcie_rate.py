# -*- coding: utf-8 -*-
"""PCIe-inclusive rate of the bench workload: every step uploads the batch's features
from host memory (gh_batch_create), runs likelihoods + Viterbi, and releases the batch."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "speech-recognition_amd"))
import bench
from sr.recognition import _hip

dt = np.float32 if "f32" in sys.argv else np.float64
wl = bench.synth_workload(1002, 10000)
W, n, M, D = wl["W"], wl["n"], wl["M"], wl["D"]
S = W * n
ctx = _hip.default_context(0)
gmm = _hip.PackedGMM(ctx, wl["means"].reshape(S, M, D), wl["vars"].reshape(S, M, D), wl["w"].reshape(S, M))
lat = _hip.Lattices(ctx, [bench.stacked_graph(W, n, wl["trans"])])
X = np.ascontiguousarray(wl["X"], dtype=dt)


def step():
    b = _hip.Batch(ctx, feats=X, offsets=wl["off"], dtype=dt)
    b.loglik(gmm, fetch=False)
    r = lat.viterbi(b, want_path=False)
    b.close()
    return r


for _ in range(5):
    step()
t0 = time.perf_counter()
K = 20
for _ in range(K):
    step()
ms = (time.perf_counter() - t0) / K * 1e3
print("PCIe-inclusive step (%s): %.2f ms -> %.3e frame-state/s (H2D of %.0f MB per step: %.1f GB/s if it were alone)" % (
    dt.__name__, ms, X.shape[0] * S / ms * 1e3, X.nbytes / 1e6, X.nbytes / ms / 1e6))
