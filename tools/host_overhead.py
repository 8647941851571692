#!/usr/bin/env python3
"""Where the wall time of Lattices.viterbi_labels goes at C5's per-GPU size: Python before / the C-ABI call / Python after."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-recognition_amd")):
    sys.path.insert(0, p)
import numpy as np
if "torch" in sys.argv:
    import torch
    torch.cuda.init()
import bench
from sr.recognition import _hip
from sr.recognition.continuous_speech import packed_lattice, packed_loop_lattice
U = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
K, W, n, M, D = 7, 10, 5, 8, 39
ctx = _hip.Context(0)
wl = bench.synth_workload(1005, 1, W=W, n=n, M=M, D=D)
rng = np.random.default_rng(7)
T = rng.integers(210, 421, size=U)
off = np.concatenate([[0], np.cumsum(T)]).astype(np.int64)
DT = np.float64 if "f64" in sys.argv else np.float32
X = rng.normal(size=(int(off[-1]), D)).astype(DT)
gmm = _hip.PackedGMM(ctx, wl["means"].reshape(W * n, M, D), wl["vars"].reshape(W * n, M, D), wl["w"].reshape(W * n, M))
b = _hip.Batch(ctx, feats=X, offsets=off, dtype=DT)
b.loglik(gmm, fetch=False)
lib = ctx.lib
orig = lib.gh_viterbi_labels_packed
for name, graph, ml in (("K7", packed_lattice([wl["trans"]] * W, n, [list(range(W))] * K)[0], K + 1),
                        ("loop", packed_loop_lattice([wl["trans"]] * W, n)[0], None)):
    lat = _hip.Lattices(ctx, [graph])
    row_word = np.where(graph["row_state"] >= 0, graph["row_state"] // n, -1).astype(np.int32)
    mlv = ml if ml is not None else b.lengths // (n - 1) + 2
    lat.viterbi_labels(b, row_word, max_labels=mlv, as_lists=False)
    spent = {"c": 0.0}
    def timed(*args):
        t0 = time.perf_counter()
        r = orig(*args)
        spent["c"] += time.perf_counter() - t0
        return r
    lib.gh_viterbi_labels_packed = timed
    t0 = time.perf_counter()
    for _ in range(3):
        lat.viterbi_labels(b, row_word, max_labels=mlv, as_lists=False)
    tot = (time.perf_counter() - t0) / 3
    lib.gh_viterbi_labels_packed = orig
    print("%s: total %.2f ms, inside the C-ABI call %.2f ms, Python around it %.2f ms" % (name, tot * 1e3, spent["c"] / 3 * 1e3, (tot - spent["c"] / 3) * 1e3))
