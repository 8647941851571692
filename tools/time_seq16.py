#!/usr/bin/env python3
"""Forced alignment of transcripts over 16-state word models (BASELINE configs[3]'s words: 64 HMMs x 16 states; 4 mixtures
here, the DP does not see them): Viterbi with paths through the sequence-form kernel (gh_seq.hip, N = 16: a word's 16 state
costs in one lane's registers, four utterances per wave) against the row-per-lane lean kernel it used to fall back to.
usage: time_seq16.py [utterances] [words per transcript] [states per word]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-recognition_amd")):
    sys.path.insert(0, p)
import numpy as np
import bench
from sr.recognition import _hip

U = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = int(sys.argv[3]) if len(sys.argv) > 3 else 16
W, M, D = 64, 4, 39
ctx = _hip.Context(0)
wl = bench.synth_workload(1007, 1, W=W, n=n, M=M, D=D)
rng = np.random.default_rng(7)
T = rng.integers(3 * n * K, 6 * n * K + 1, size=U)
off = np.concatenate([[0], np.cumsum(T)]).astype(np.int64)
X = rng.normal(size=(int(off[-1]), D))
gmm = _hip.PackedGMM(ctx, wl["means"].reshape(W * n, M, D), wl["vars"].reshape(W * n, M, D), wl["w"].reshape(W * n, M))
b = _hip.Batch(ctx, feats=X, offsets=off)
labels = rng.integers(0, W, size=(U, K))
keys, ug = {}, np.empty(U, dtype=np.int32)
for u in range(U):
    ug[u] = keys.setdefault(tuple(int(v) for v in labels[u]), len(keys))
lat = _hip.Lattices.from_transcripts(ctx, [wl["trans"]] * W, n, list(keys))
lo = np.repeat(np.arange(U + 1, dtype=np.int64) * K, 1)
b.loglik(gmm, fetch=False, state_sets=(lo, (labels.reshape(-1) * n).astype(np.int32), (labels.reshape(-1) * n + n).astype(np.int32)))
print("%d utterances of %d words x %d states, %d frames, %d graphs (forms %s)" % (U, K, n, off[-1], len(keys), sorted(lat.forms())))


def timed(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


res = {}
for name, env in (("sequence form", {}), ("row-per-lane (lean)", {"GMMHMM_VITERBI": "lean"})):
    os.environ.update(env)
    ms = timed(lambda: lat.viterbi(b, utt_lattice=ug, want_path=True))
    r = lat.viterbi(b, utt_lattice=ug, want_path=True)
    res[name] = r
    print("%-20s viterbi with paths %.2f ms" % (name, ms))
    os.environ.pop("GMMHMM_VITERBI", None)
a, c = res["sequence form"], res["row-per-lane (lean)"]
print("end costs identical:", bool(np.array_equal(a["end_cost_flat"], c["end_cost_flat"])),
      " paths identical:", all(np.array_equal(x, y) for x, y in zip(a["paths"], c["paths"])))
