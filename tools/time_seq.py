#!/usr/bin/env python3
"""Forced-alignment graphs (one word per layer, a graph per distinct transcript): Viterbi with paths and
forward-backward with occupancies through the sequence-form kernels against the row-per-lane kernels
(GMMHMM_VITERBI=lean / GMMHMM_FB=generic).  Wall time of the C-ABI calls.
usage: time_seq.py [utterances] [words per transcript]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-recognition_amd")):
    sys.path.insert(0, p)
import numpy as np
import bench
from sr.recognition import _hip

U = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
K = int(sys.argv[2]) if len(sys.argv) > 2 else 7
W, n, M, D = 10, 5, 8, 39
ctx = _hip.Context(0)
wl = bench.synth_workload(1005, 1, W=W, n=n, M=M, D=D)
rng = np.random.default_rng(7)
T = rng.integers(30 * K, 60 * K + 1, size=U)
off = np.concatenate([[0], np.cumsum(T)]).astype(np.int64)
X = rng.normal(size=(int(off[-1]), D))
gmm = _hip.PackedGMM(ctx, wl["means"].reshape(W * n, M, D), wl["vars"].reshape(W * n, M, D), wl["w"].reshape(W * n, M))
b = _hip.Batch(ctx, feats=X, offsets=off)
b.loglik(gmm, fetch=False)
labels = rng.integers(0, W, size=(U, K))
keys, ug = {}, np.empty(U, dtype=np.int32)
for u in range(U):
    ug[u] = keys.setdefault(tuple(int(v) for v in labels[u]), len(keys))
t0 = time.perf_counter()
lat = _hip.Lattices.from_transcripts(ctx, [wl["trans"]] * W, n, list(keys))   # (the row-per-lane arrays are expanded on demand)
print("%d utterances, %d frames, %d graphs (forms %s), Lattices.from_transcripts() %.1f ms" % (U, off[-1], len(keys), sorted(lat.forms()),
                                                                                             (time.perf_counter() - t0) * 1e3))


def timed(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


ref = {}
for name, env in (("sequence form", {}), ("row-per-lane", {"GMMHMM_VITERBI": "lean", "GMMHMM_FB": "generic"})):
    os.environ.update(env)
    v = timed(lambda: lat.viterbi(b, utt_lattice=ug, want_path=False), reps=2)
    f = timed(lambda: lat.forward_backward(b, utt_lattice=ug, want_occ=True, fetch_occ=False, want_self_xi=True), reps=2)
    r = lat.viterbi(b, utt_lattice=ug, want_path=False)["end_cost_flat"]
    lp = lat.forward_backward(b, utt_lattice=ug)["logp"]
    ref.setdefault("v", r); ref.setdefault("lp", lp)
    print("%-14s viterbi (costs) %.2f ms   forward-backward (occ, xi) %.2f ms   | costs identical: %s, max |dlogP|/|logP| %.1e" % (
        name, v, f, np.array_equal(r, ref["v"]), np.max(np.abs(lp - ref["lp"]) / np.abs(ref["lp"]))))
os.environ.pop("GMMHMM_VITERBI", None); os.environ.pop("GMMHMM_FB", None)
print("sequence-form forward-backward, parts: forward only %.2f ms | + backward with xi %.2f ms | + occupancies %.2f ms" % (
    timed(lambda: lat.forward_backward(b, utt_lattice=ug)),
    timed(lambda: lat.forward_backward(b, utt_lattice=ug, want_self_xi=True)),
    timed(lambda: lat.forward_backward(b, utt_lattice=ug, want_occ=True, fetch_occ=False, want_self_xi=True))))
