# -*- coding: utf-8 -*-
"""End to end through the mirror API, the way the reference's scripts chain it (cli.py isolated_train ->
aurora_continuous_train -> main.py): isolated-word training (HMM.fit), embedded Viterbi training on word strings
(continuous_train, pickles written per iteration), then recognition with the models that were written -- isolated words
(core.test's report) and word strings through the K-layer lattice and the loop grammar (main.py's report)."""
import contextlib
import io
import os
import pickle
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _utt(rng, means, vars_, words, n, M, tmin, tmax):
    segs = []
    for wd in words:
        T = int(rng.integers(tmin, tmax))
        st = np.minimum(np.arange(T) * n // T, n - 1)
        comp = rng.integers(0, M, size=T)
        segs.append(means[wd, st, comp] + np.sqrt(vars_[wd, st, comp]) * rng.normal(size=(T, means.shape[-1])))
    return np.concatenate(segs)


def test_train_then_recognise(tmp_path):
    import sr.recognition as R
    from sr.recognition.batch import ContinuousDecoder, IsolatedWordRecognizer
    rng = np.random.default_rng(2024)
    W, n, M, D, K = 5, 4, 2, 10, 3
    means = rng.normal(size=(W, n, M, D)) * 2.5
    vars_ = rng.uniform(0.5, 1.2, size=(W, n, M, D))
    quiet = contextlib.redirect_stdout(io.StringIO())
    np.random.seed(1)
    with quiet, warnings.catch_warnings():
        warnings.simplefilter("ignore")
        models = [R.HMM(n).fit([_utt(rng, means, vars_, [w], n, M, 24, 40) for _ in range(12)], 4) for w in range(W)]
        for h in models:
            for s in h.gmm_states:
                s.parent = h
        strings = [[int(v) for v in rng.integers(0, W, size=K)] for _ in range(40)]
        data = [_utt(rng, means, vars_, s, n, M, 24, 40) for s in strings]
        R.continuous_train(data, models, strings, str(tmp_path), n_gaussians=4, n_segments=n, max_iteration=2)
    trained = [pickle.load(open(os.path.join(str(tmp_path), "%d.pkl" % w), "rb")) for w in range(W)]
    for h in trained:
        assert np.all(np.isfinite(h.transitions[np.arange(n), np.arange(n)][:-1]))
        assert all(np.all(np.asarray(d.cov) > 0) for s in h.gmm_states for d in s.dists)
    # isolated words (core.py:63-94)
    test_words = rng.integers(0, W, size=60)
    acc, got = IsolatedWordRecognizer(trained).accuracy([_utt(rng, means, vars_, [w], n, M, 24, 40) for w in test_words], test_words)
    assert acc >= 0.95, acc
    # word strings (main.py:35, 59-84): exactly-K lattice and the loop grammar
    test_strings = [[int(v) for v in rng.integers(0, W, size=K)] for _ in range(40)]
    xs = [_utt(rng, means, vars_, s, n, M, 24, 40) for s in test_strings]
    rep = ContinuousDecoder(trained, n_layers=K).accuracy(xs, test_strings)
    assert rep["sequence_accuracy"] >= 0.85 and rep["digit_accuracy"] >= 0.95, rep
    rep_loop = ContinuousDecoder(trained, grammar="loop", word_penalty=5.0).accuracy(xs, test_strings)
    assert rep_loop["digit_accuracy"] >= 0.85, rep_loop
