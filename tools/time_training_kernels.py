#!/usr/bin/env python3
"""The training-side kernels at realistic sizes (configs[2] model: 39-dim, 8 mixtures): template DP of skmeans
(gh_dtw, 5-row templates), one-state k-means assignment / EM statistics (gh_kmeans_assign, gh_em_accumulate), the
front-end (deltas + standardise from 13 cepstra; MFCC from PCM).  Wall time per call; run under
rocprofv3 --kernel-trace --stats for the kernel times.  usage: time_training_kernels.py [utterances]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-recognition_amd")):
    sys.path.insert(0, p)
import numpy as np
import bench
from sr.recognition import _hip

U = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
wl = bench.synth_workload(1003, U)
W, n, M, D = wl["W"], wl["n"], wl["M"], wl["D"]
ctx = _hip.Context(0)
b = _hip.Batch(ctx, feats=wl["X"], offsets=wl["off"])
N = b.N


def timed(name, fn, unit_count, unit, reps=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    dt = (time.perf_counter() - t0) / reps
    print("%-44s %8.3f ms per call   %.3g %s/s" % (name, dt * 1e3, unit_count / dt, unit), flush=True)


rng = np.random.default_rng(0)
y = wl["means"][0, :, 0, :]                               # a word's 5 state means as template rows
var = wl["vars"][0, :, 0, :]
timed("gh_dtw euclidean, 5 rows, paths", lambda: b.dtw(wl["trans"], y=y, want_costs=False), N, "frames")
timed("gh_dtw mahalanobis, 5 rows, costs + paths", lambda: b.dtw(wl["trans"], y=y, var=var), N, "frames")
cent = rng.normal(size=(8, D))
timed("gh_kmeans_assign k=8 (all frames, one state)", lambda: b.kmeans_assign(cent, var=var[0]), N, "frames")
mean, v, w = wl["means"][0, 0], wl["vars"][0, 0], wl["w"][0, 0]
timed("gh_em_accumulate k=8 (all frames, one state)", lambda: b.em_accumulate(mean, v, w), N, "frames")
S = W * n
seg = np.linspace(0, N, S + 1).astype(np.int64)
cents = rng.normal(size=(S, 8, D))
b.resident_clusters(reset=True, fetch=False)
timed("gh_kmeans_assign_multi 50 states k=8 + sums", lambda: b.kmeans_assign_multi(seg, cents, var=wl["vars"].reshape(S, M, D)[:, 0],
                                                                                  clusters=_hip.RESIDENT, want_sums=True), N, "frames")
timed("gh_em_accumulate_multi 50 states k=8", lambda: b.em_accumulate_multi(seg, wl["means"].reshape(S, M, D), wl["vars"].reshape(S, M, D),
                                                                            wl["w"].reshape(S, M)), N, "frames")
ceps = [rng.normal(size=(int(t), 13)) for t in np.diff(wl["off"])]


def frontend():
    _hip.Batch(ctx, cepstra=ceps).close()


timed("cepstra -> deltas -> standardise (resident)", frontend, N, "frames")
pcm = [(rng.normal(size=16000) * 3000).astype(np.int16) for _ in range(min(U, 2000))]


def mfcc_batch():
    _hip.Batch(ctx, pcm=pcm, sample_rate=16000).close()


timed("PCM (1 s each) -> MFCC -> deltas -> standardise", mfcc_batch, len(pcm) * 100, "frames")
b.close()
