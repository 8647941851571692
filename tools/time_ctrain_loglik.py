# -*- coding: utf-8 -*-
"""Where an outer iteration of continuous_train spends its time: every _hip call of the loop timed with a stream
synchronisation on either side (bench.py's C3_continuous_train shape)."""
import contextlib
import io
import os
import sys
import tempfile
import time
import warnings
from collections import defaultdict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-recognition_amd"))
import bench
from sr.recognition import _hip, continuous_speech as cs, lockstep
from sr.recognition.model_io import models_from_arrays

U, K, iters = int(os.environ.get("U", "2000")), 7, int(os.environ.get("ITERS", "4"))
wl = bench.synth_workload(1003, U * K)
W, n, M, D = wl["W"], wl["n"], wl["M"], wl["D"]
iso = [wl["X"][wl["off"][u]:wl["off"][u + 1]] for u in range(U * K)]
data = [np.concatenate(iso[i * K:(i + 1) * K]) for i in range(U)]
labels = [[int(w) for w in wl["words"][i * K:(i + 1) * K]] for i in range(U)]
means0 = wl["means"] + 0.3 * np.random.default_rng(0).normal(size=wl["means"].shape)
models = models_from_arrays(means0, wl["vars"], wl["w"], [wl["trans"]] * W, mu=means0[:, :, 0], sigma=wl["vars"][:, :, 0])
acc = defaultdict(float)
cnt = defaultdict(int)
ctx = _hip.default_context()


def timed(cls, name):
    f = getattr(cls, name)

    def g(*a, **k):
        ctx.sync()
        t0 = time.perf_counter()
        r = f(*a, **k)
        ctx.sync()
        acc[cls.__name__ + "." + name] += time.perf_counter() - t0
        cnt[cls.__name__ + "." + name] += 1
        return r
    setattr(cls, name, g)


which = os.environ.get("LEVEL", "outer")
tab = ((_hip.Batch, ["loglik"]), (_hip.Lattices, ["align_segments"]), (_hip.PackedGMM, ["update"]),
       (lockstep.LockstepFitter, ["split_and_fit", "segment_means", "__init__"]))
if which == "inner":      # the calls inside the fitter instead of the fitter's own methods
    tab = ((_hip.Batch, ["loglik", "gather", "gather_runs", "close"]), (_hip.Lattices, ["align_segments", "align_runs", "from_transcripts", "close"]),
           (_hip.FitSession, ["__init__", "kmeans", "em", "segment_means", "close", "clusters", "group_stats"]))
for cls, names in tab:
    for nm in names:
        if hasattr(cls, nm):
            timed(cls, nm)
np.random.seed(0)
t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings(), tempfile.TemporaryDirectory() as out:
    warnings.simplefilter("ignore")
    cs.continuous_train(data, models, labels, out, n_gaussians=M, n_segments=n, max_iteration=iters)
tot = time.perf_counter() - t0
print("total %.1f ms for %d outer iterations" % (tot * 1e3, iters))
for k in sorted(acc, key=lambda k: -acc[k]):
    print("%-40s %4d calls %8.2f ms per outer iteration" % (k, cnt[k], acc[k] * 1e3 / iters))
print("%-40s            %8.2f ms per outer iteration" % ("everything else (host)", (tot - sum(acc.values())) * 1e3 / iters))
