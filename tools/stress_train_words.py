# -*- coding: utf-8 -*-
"""Randomised sweep of `batch.train_words` (all words of a vocabulary trained in one pass: sr/core.py:47-60 does it word after
word) against the word-after-word loop `HMM(n).fit(templates, n_gaussians)` under the same numpy seed: random vocabularies
(1 .. 7 words, 2 .. 12 templates of 8 .. 90 frames), 2 .. 8 states, 1 .. 8 mixtures, D = 2 .. 39, with and without mixtures /
EM.  Segmental k-means results and the templates' segments exactly, mixtures to 1e-9; an error (LinAlgError for a collapsed
mixture, NameError for a template that is too short) must be the same error.

    python tools/stress_train_words.py [trials] [seed]
"""
import contextlib
import io
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-recognition_amd"))
import sr.recognition as R
from sr.recognition import _pack
from sr.recognition.batch import train_words

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        yield


def guarded(fn):
    try:
        with quiet():
            return fn(), None
    # (IndexError: hmm.py:116-118 with an empty cluster; AssertionError: hmm.py:105, n_gaussians < 3 with mixtures)
    except (np.linalg.LinAlgError, NameError, ValueError, ZeroDivisionError, IndexError, AssertionError) as e:
        return None, type(e).__name__


bad = 0
t0 = time.time()
for trial in range(trials):
    W, n, D = int(rng.integers(1, 8)), int(rng.integers(2, 9)), int(rng.choice([2, 5, 13, 39]))
    ng = int(rng.choice([1, 2, 3, 4, 8]))
    use_gmm = bool(rng.random() < 0.85)
    use_em = use_gmm and bool(rng.random() < 0.8)
    words = []
    for w in range(W):
        mu = rng.normal(size=(n, D)) * 3.0
        ys = []
        for r in range(int(rng.integers(2, 13))):
            T = int(rng.integers(max(8, 2 * n), 91))
            st = np.minimum(np.arange(T) * n // T, n - 1)
            ys.append(mu[st] + rng.normal(size=(T, D)) * rng.uniform(0.5, 1.5, size=D))
        words.append(ys)
    seed = int(rng.integers(0, 1 << 30))
    np.random.seed(seed)
    seq, seq_err = guarded(lambda: [R.HMM(n).fit([y.copy() for y in ys], ng, use_gmm=use_gmm, use_em=use_em) for ys in words])
    np.random.seed(seed)
    par, par_err = guarded(lambda: train_words(words, n, n_gaussians=ng, use_gmm=use_gmm, use_em=use_em))
    problems = []
    if seq_err != par_err:
        problems.append("errors differ: loop %s, train_words %s" % (seq_err, par_err))
    elif seq is not None:
        for wi, (a, b) in enumerate(zip(par, seq)):
            if not (np.array_equal(a.mu, b.mu) and np.array_equal(a.sigma, b.sigma) and np.array_equal(a.transitions, b.transitions)):
                problems.append("word %d: segmental k-means" % wi)
            if [len(s) for s in a.segments] != [len(s) for s in b.segments] or not all(np.array_equal(x, y) for x, y in zip(a.segments, b.segments)):
                problems.append("word %d: segments" % wi)
            if use_gmm:
                (ma, va, wa), (mb, vb, wb) = _pack.stack_gmms(a.gmm_states), _pack.stack_gmms(b.gmm_states)
                with np.errstate(all="ignore"):
                    if not (np.allclose(ma, mb, rtol=1e-9, atol=1e-12, equal_nan=True) and np.allclose(va, vb, rtol=1e-9, equal_nan=True)
                            and np.allclose(wa, wb, rtol=1e-9, equal_nan=True)):
                        problems.append("word %d: mixtures (max rel %.3g)" % (wi, float(np.nanmax(np.abs(va - vb) / np.abs(vb)))))
    if problems:
        bad += 1
        print("trial %d: W=%d n=%d D=%d ng=%d use_gmm=%s use_em=%s seed=%d: %s" % (trial, W, n, D, ng, use_gmm, use_em, seed, "; ".join(problems[:4])), flush=True)
    elif trial % 10 == 0:
        print("trial %d ok (W=%d n=%d D=%d ng=%d gmm=%s em=%s%s), %.0f s" % (trial, W, n, D, ng, use_gmm, use_em, ", both raise " + seq_err if seq_err else "", time.time() - t0), flush=True)
print("%d trials, %d with differences" % (trials, bad))
sys.exit(1 if bad else 0)
