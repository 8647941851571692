# -*- coding: utf-8 -*-
"""One full pass of the lock-step E-step kernel (em_multi_kernel) and of the k-means assignment (kmeans_multi_kernel) on the
continuous_train shape: 50 states x ~28 000 frames, 39 dims, k = 8 -- wall time per call; kernel times under rocprofv3."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-recognition_amd"))
from sr.recognition import _hip

ctx = _hip.default_context()
rng = np.random.default_rng(0)
S, k, D = 50, 8, 39
lens = rng.integers(20000, 36000, size=S)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
N = int(off[-1])
X = rng.normal(size=(N, D))
b = _hip.Batch(ctx, feats=X, offsets=[0, N])
mean = rng.normal(size=(S, k, D))
var = rng.uniform(0.5, 2.0, size=(S, k, D))
w = rng.dirichlet(np.ones(k), size=S)
for name, fn in (("em_accumulate_multi", lambda: b.em_accumulate_multi(off, mean, var, w)),
                 ("kmeans_assign_multi", lambda: b.kmeans_assign_multi(off, mean, var=var[:, 0, :]))):
    for _ in range(2):
        fn()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    print("%s: %.3f ms per call (%d frames)" % (name, (time.perf_counter() - t0) / 5 * 1e3, N))
