#!/usr/bin/env python3
"""Continuous decode over WIDE word models (12 / 16 states per word): the K-layer lattice and the word-loop grammar through the
layer-form / loop-form kernels (64-bit decision words where two register sets of N + 1 (+ N - 2) bits pass 32) against the
row-per-lane lean kernel those graphs took before round 5.  Same end costs, ends and paths; wall time per call.

    python tools/time_layers_wide.py [utterances] [states per word] [words] [layers]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-recognition_amd")):
    sys.path.insert(0, p)
import numpy as np
import bench
from sr.recognition import _hip
from sr.recognition.continuous_speech import packed_lattice, packed_loop_lattice

U = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
W = int(sys.argv[3]) if len(sys.argv) > 3 else 10
K = int(sys.argv[4]) if len(sys.argv) > 4 else 7
M, D = 8, 39
ctx = _hip.default_context()
wl = bench.synth_workload(1005, 1, W=W, n=n, M=M, D=D)
rng = np.random.default_rng(7)
T = rng.integers(K * n * 2, K * n * 4 + 1, size=U)
off = np.concatenate([[0], np.cumsum(T)]).astype(np.int64)
X = rng.normal(size=(int(off[-1]), D))
gmm = _hip.PackedGMM(ctx, wl["means"].reshape(W * n, M, D), wl["vars"].reshape(W * n, M, D), wl["w"].reshape(W * n, M))
b = _hip.Batch(ctx, feats=X, offsets=off)
b.loglik(gmm, fetch=False)
zeros = np.zeros(U, dtype=np.int32)
print("%d utterances, %d frames, %d words x %d states, %d mixtures, D = %d" % (U, off[-1], W, n, M, D))
for name, graph in (("K = %d lattice" % K, packed_lattice([wl["trans"]] * W, n, [list(range(W))] * K)[0]),
                    ("word loop", packed_loop_lattice([wl["trans"]] * W, n, 1.0)[0])):
    lat = _hip.Lattices(ctx, [graph])
    res = {}
    for form, kw in (("word templates", {}), ("row per lane", dict(utt_lattice=zeros))):
        r = lat.viterbi(b, want_path=True, **kw)
        t0 = time.perf_counter()
        for _ in range(3):
            r = lat.viterbi(b, want_path=True, **kw)
        res[form] = (r, (time.perf_counter() - t0) / 3 * 1e3)
    a, c = res["word templates"][0], res["row per lane"][0]
    same = (np.array_equal(a["end_cost_flat"], c["end_cost_flat"]) and np.array_equal(a["best_end"], c["best_end"])
            and all(np.array_equal(p, q) for p, q in zip(a["paths"], c["paths"])))
    print("%-14s forms %s: %.2f ms (word templates) vs %.2f ms (row per lane), identical results: %s"
          % (name, sorted(lat.forms()), res["word templates"][1], res["row per lane"][1], same))
    lat.close()
