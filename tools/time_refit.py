# -*- coding: utf-8 -*-
"""The device-resident refit's two per-iteration kernels on the continuous_train shape (50 states x ~28 000 frames, 39
dims): streaming matrix-core form (gh_refit_mfma.hip, the default) against the tile kernels (GMMHMM_REFIT=tiles) --
same assignments / centroids / mixtures, wall time per k-means and EM call; kernel times under rocprofv3.

    python tools/time_refit.py [k] [S] [frames per state] [D]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-recognition_amd"))
from sr.recognition import _hip

k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 50
n_avg = int(sys.argv[3]) if len(sys.argv) > 3 else 28000
D = int(sys.argv[4]) if len(sys.argv) > 4 else 39
ctx = _hip.default_context()
rng = np.random.default_rng(0)
lens = rng.integers(int(n_avg * 0.7), int(n_avg * 1.3) + 1, size=S)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
N = int(off[-1])
# every state a mixture of k clusters around its own point, scales differing per dimension
X = np.empty((N, D))
for s in range(S):
    c = rng.normal(size=(k, D)) * 2.0 + rng.normal(size=D) * 5.0
    which = rng.integers(0, k, size=lens[s])
    X[off[s]:off[s + 1]] = c[which] + rng.normal(size=(lens[s], D)) * rng.uniform(0.5, 1.5, size=D)
b = _hip.Batch(ctx, feats=X, offsets=[0, N])
part = rng.integers(0, k, size=N).astype(np.uint8)
seg_mean = np.array([X[off[s]:off[s + 1]].mean(axis=0) for s in range(S)])
c0 = np.stack([seg_mean * f for f in np.linspace(0.8, 1.2, k)], axis=1)
n_frames = np.diff(off).astype(np.float64)


def run(mode):
    if mode == "tiles":
        os.environ["GMMHMM_REFIT"] = "tiles"
    else:
        os.environ.pop("GMMHMM_REFIT", None)
    fit = _hip.FitSession(ctx, b, off, max(k, 2))
    out = {}
    for rep in range(3):
        t0 = time.perf_counter()
        cen, cov, cnt, its = fit.kmeans(k, c0, part, max_iteration=40)
        out["kmeans_ms"] = (time.perf_counter() - t0) * 1e3
    out.update(cen=cen, cov=cov, cnt=cnt, its=its, ids=fit.clusters())
    w = cnt / n_frames[:, None]
    for rep in range(3):
        mean, var, ww = cen.copy(), np.ascontiguousarray(np.broadcast_to(cov[:, :1, :], cen.shape)).copy(), w.copy()
        var = cov.copy()
        mu_old, sg_old, w_old = np.zeros_like(mean), np.ones_like(mean), np.zeros_like(ww)
        t0 = time.perf_counter()
        conv = fit.em(k, mean, var, ww, mu_old, sg_old, w_old, n_frames, max_iteration=30)
        out["em_ms"] = (time.perf_counter() - t0) * 1e3
    out.update(mean=mean, var=var, w=ww, conv=conv)
    fit.close()
    return out


new, old = run("mfma"), run("tiles")
print("%d states, %d frames, D = %d, k = %d" % (S, N, D, k))
print("k-means call: streaming %.2f ms (%d iterations max), tiles %.2f ms (%d)" % (new["kmeans_ms"], new["its"].max(), old["kmeans_ms"], old["its"].max()))
print("EM call:      streaming %.2f ms (converged at %s...), tiles %.2f ms (%s...)" % (new["em_ms"], new["conv"][:6], old["em_ms"], old["conv"][:6]))
print("assignments equal:", bool(np.array_equal(new["ids"], old["ids"])), " (differing frames: %d)" % int((new["ids"] != old["ids"]).sum()))
print("iterations equal:", bool(np.array_equal(new["its"], old["its"])))
print("centroids bitwise equal:", bool(np.array_equal(new["cen"], old["cen"], equal_nan=True)), " max |diff| %.3g" % np.nanmax(np.abs(new["cen"] - old["cen"])))
print("cluster sizes equal:", bool(np.array_equal(new["cnt"], old["cnt"])))
print("EM converged_at equal:", bool(np.array_equal(new["conv"], old["conv"])))
for name in ("mean", "var", "w"):
    d = np.abs(new[name] - old[name]) / (np.abs(old[name]) + 1e-300)
    print("EM %s: max rel diff %.3g" % (name, np.nanmax(d)))
