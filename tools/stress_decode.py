# -*- coding: utf-8 -*-
"""Randomised sweep of the word-template Viterbi kernels -- layer form (K-layer lattice), loop form (word-loop grammar), sequence
form (forced alignment, a graph per transcript) -- against the row-per-lane lean kernel, which implements the same
decode_hmm_states semantics (decode.py:80-146) by another route: random word counts (1 .. 16, and 17 .. 64 for the wide layer kernel), states per word (2 .. 8, 12,
16), layers (1 .. 16), skip arcs, penalties, fp64 / fp32, utterances from too short to long.  End costs BITWISE, the chosen
end, paths and labels equal.

    python tools/stress_decode.py [trials] [seed]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-recognition_amd"))
from sr.recognition import _hip
from sr.recognition.continuous_speech import packed_lattice, packed_loop_lattice

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = _hip.default_context()


def word_trans(n, skip, last_self):
    t = np.full((n, n), np.inf)
    for i in range(n):
        t[i, i] = rng.uniform(0.05, 0.6) if i < n - 1 else last_self
        if i < n - 1:
            t[i + 1, i] = rng.uniform(0.8, 2.5)
        if skip and i < n - 2 and rng.random() < 0.6:
            t[i + 2, i] = rng.uniform(1.5, 4.0)
    return t


class lean_kernel:
    def __enter__(self):
        os.environ["GMMHMM_VITERBI"] = "lean"

    def __exit__(self, *exc):
        os.environ.pop("GMMHMM_VITERBI", None)


def compare(tag, lat, b, n, graph_row_state=None, utt_graph=None):
    """fast (the form's own kernel) against lean; returns a list of differences."""
    kw = {} if utt_graph is None else dict(utt_lattice=utt_graph)
    out = []

    def run(fn):
        try:
            return fn(), None
        except _hip.BackendError as e:
            return None, str(e)[:80]
    fast, ferr = run(lambda: lat.viterbi(b, want_path=True, **kw))
    with lean_kernel():
        lean, lerr = run(lambda: lat.viterbi(b, want_path=True, **(kw if utt_graph is not None else dict(utt_lattice=np.zeros(b.U, dtype=np.int32)))))
    if (fast is None) != (lean is None):
        return ["%s: one route fails (%s / %s)" % (tag, ferr, lerr)]
    if fast is None:
        return [] if ("without predecessor" in ferr) == ("without predecessor" in lerr) else ["%s: different errors (%s / %s)" % (tag, ferr, lerr)]
    if not np.array_equal(fast["end_cost_flat"], lean["end_cost_flat"], equal_nan=True):
        out.append("%s: end costs" % tag)
    if not np.array_equal(fast["best_end"], lean["best_end"]):
        out.append("%s: chosen ends" % tag)
    bad_paths = sum(not np.array_equal(p, q) for p, q in zip(fast["paths"], lean["paths"]))
    if bad_paths:
        out.append("%s: %d paths" % (tag, bad_paths))
    nopath = lat.viterbi(b, want_path=False, **kw)
    if not np.array_equal(nopath["end_cost_flat"], lean["end_cost_flat"], equal_nan=True):
        out.append("%s: end costs without paths" % tag)
    if graph_row_state is not None:
        row_word = np.where(graph_row_state >= 0, graph_row_state // n, -1).astype(np.int32)
        la = lat.viterbi_labels(b, row_word)
        lb = lat.viterbi_labels(b, row_word, utt_lattice=np.zeros(b.U, dtype=np.int32))
        bad = sum(not np.array_equal(x, y) for x, y in zip(la["labels"], lb["labels"]))
        if bad:
            out.append("%s: %d label strings" % (tag, bad))
    return out


bad = 0
t0 = time.time()
for trial in range(trials):
    n = int(rng.choice([2, 3, 4, 5, 6, 7, 8, 12, 16]))
    W = int(rng.integers(1, 17))
    K = int(rng.integers(1, 17 if n <= 8 else 9))
    wide = n <= 8 and rng.random() < 0.3          # more than 16 words per layer: the wide layer kernel (K <= 8, N <= 8)
    if wide:
        W, K = int(rng.integers(17, 65)), int(rng.integers(1, 9))
    skip = bool(rng.random() < 0.4) and n >= 3
    M, D = int(rng.choice([1, 2, 3])), int(rng.choice([2, 6, 13]))
    dtype = np.float64 if rng.random() < 0.7 else np.float32
    if (K * (W * n + 1) + 1) > 1500:          # (rows of the lattice: keep the lean kernel's run short)
        K = max(1, 1400 // (W * n + 1))
    means = rng.normal(size=(W, n, M, D)) * 2.0
    vars_ = rng.uniform(0.5, 1.5, size=(W, n, M, D))
    w = rng.dirichlet(np.ones(M), size=(W, n))
    wt = [word_trans(n, skip, rng.uniform(0.0, 0.3)) for _ in range(W)]
    xs, labels = [], []
    U = int(rng.integers(3, 70))
    for u in range(U):
        kk = int(rng.integers(1, K + 1))
        words = [int(v) for v in rng.integers(0, W, size=kk)]
        labels.append(words)
        r = rng.random()
        if r < 0.15:
            xs.append(rng.normal(size=(int(rng.integers(2, max(3, kk * (n - 1) + 2))), D)) * 2.0)      # too short for its words
            continue
        segs = []
        for wd in words:
            Tw = int(rng.integers(n, 3 * n + 4)) if r < 0.9 else int(rng.integers(8 * n, 20 * n))
            st = np.minimum(np.arange(Tw) * n // Tw, n - 1)
            comp = rng.integers(0, M, size=Tw)
            segs.append(means[wd, st, comp] + np.sqrt(vars_[wd, st, comp]) * rng.normal(size=(Tw, D)))
        xs.append(np.concatenate(segs))
    gmm = _hip.PackedGMM(ctx, means.reshape(W * n, M, D), vars_.reshape(W * n, M, D), w.reshape(W * n, M))
    b = _hip.Batch(ctx, xs, dtype=dtype)
    b.loglik(gmm, fetch=False)
    problems = []
    try:
        # layer form
        graph = packed_lattice(wt, n, [list(range(W))] * K)[0]
        lat = _hip.Lattices(ctx, [graph])
        if "layers" not in lat.forms():
            problems.append("K = %d lattice not taken as layer form (%s)" % (K, sorted(lat.forms())))
        problems += compare("layers", lat, b, n, graph["row_state"])
        lat.close()
        # loop form (up to 16 words: four utterances per wave; more: one per wave, up to 8 states per word)
        if W <= 16 or n <= 8:
            graph = packed_loop_lattice(wt, n, float(rng.choice([0.0, 0.7, 2.5])))[0]
            lat = _hip.Lattices(ctx, [graph])
            if "loop" not in lat.forms():
                problems.append("loop grammar not taken as loop form (%s)" % sorted(lat.forms()))
            problems += compare("loop", lat, b, n, graph["row_state"])
            lat.close()
        # sequence form: every utterance against its own transcript
        keys, graphs, utt_graph = {}, [], np.empty(U, dtype=np.int32)
        for u, l in enumerate(labels):
            if tuple(l) not in keys:
                keys[tuple(l)] = len(graphs)
                graphs.append(packed_lattice(wt, n, [[x] for x in l])[0])
            utt_graph[u] = keys[tuple(l)]
        lat = _hip.Lattices(ctx, graphs)
        if "sequence" not in lat.forms():
            problems.append("transcripts not taken as sequence form (%s)" % sorted(lat.forms()))
        problems += compare("sequence", lat, b, n, None, utt_graph)
        lat.close()
    finally:
        b.close()
        gmm.close()
    if problems:
        bad += 1
        print("trial %d: W=%d n=%d K=%d skip=%s M=%d D=%d %s U=%d: %s" % (trial, W, n, K, skip, M, D, np.dtype(dtype).name, U, "; ".join(problems)), flush=True)
    elif trial % 10 == 0:
        print("trial %d ok (W=%d n=%d K=%d skip=%s %s), %.0f s" % (trial, W, n, K, skip, np.dtype(dtype).name, time.time() - t0), flush=True)
print("%d trials, %d with differences" % (trials, bad))
sys.exit(1 if bad else 0)
