# -*- coding: utf-8 -*-
"""Isolated-word training of MANY word models in one pass (`batch.train_words`, `kmeans.skmeans_multi`) against the
word-after-word loop of the reference's driver (sr/core.py:47-60: one `HMM(n).fit(templates_of_the_word)` per digit):
the words only share launches, so under the same numpy seed every model must come out the same."""
import numpy as np
import pytest

from conftest import load_golden
from test_gpu_api import quiet, pack_hmm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    import sr.recognition as R
    return R


def _km():
    import importlib
    return importlib.import_module("sr.recognition.kmeans")     # (sr.recognition.kmeans the attribute is the function)


def _words(W=4, n=5, D=13, seed=3):
    """W synthetic words: n states with their own means, templates of 30-70 frames, 6-9 templates per word."""
    rng = np.random.default_rng(seed)
    out = []
    for w in range(W):
        mu = rng.normal(size=(n, D)) * 3.0
        ys = []
        for r in range(int(rng.integers(6, 10))):
            T = int(rng.integers(30, 71))
            st = np.minimum(np.arange(T) * n // T, n - 1)
            ys.append(mu[st] + rng.normal(size=(T, D)) * rng.uniform(0.5, 1.5, size=D))
        out.append(ys)
    return out


def test_skmeans_multi_equals_skmeans_word_by_word(R):
    km = _km()
    words = _words()
    multi = km.skmeans_multi(words, 5)
    for ys, (res, var, trans, starts) in zip(words, multi):
        r1, v1, t1, seg1 = R.skmeans(ys, 5, return_segmented_data=True)
        np.testing.assert_array_equal(res, r1)
        np.testing.assert_array_equal(var, v1)
        np.testing.assert_array_equal(trans, t1)
        for a, b in zip(km.segment_data(ys, len(ys), 5, starts), seg1):
            np.testing.assert_array_equal(a, b)
    # ... and the device path against the host bookkeeping of the same algorithm (np.mean / np.cov per segment):
    # same alignments, means bit for bit, variances to the rounding of np.cov's BLAS product
    for ys, (res, var, trans, starts) in zip(words, multi):
        r2, v2, t2, s2 = km._skmeans_host(ys, 5, km.euclidean, 1000)
        np.testing.assert_array_equal(starts, s2)
        np.testing.assert_array_equal(res, r2)
        np.testing.assert_allclose(var, v2, rtol=1e-12)
        np.testing.assert_array_equal(trans, t2)


def test_skmeans_golden_through_the_device_path(R):
    """G9 (the reference's own skmeans): means / variances / transition costs / segments."""
    g = load_golden("G9_hmm_fit_single")
    ys = [g["y%d" % i] for i in range(int(g["n_templates"]))] if "n_templates" in g else None
    if ys is None:
        from test_gpu_api import _ys
        ys = _ys(g)
    km = _km()
    (res, var, trans, starts), = km.skmeans_multi([ys], 5)
    np.testing.assert_allclose(res, g["mu"], rtol=1e-12)
    np.testing.assert_allclose(var, g["sigma"], rtol=1e-12)
    np.testing.assert_allclose(trans, g["transitions"], rtol=1e-12)
    assert [len(s) for s in km.segment_data(ys, len(ys), 5, starts)] == list(g["seg_lens"])


@pytest.mark.parametrize("use_gmm,use_em,ng", [(True, True, 4), (True, False, 8), (False, False, 1)])
def test_train_words_equals_the_word_after_word_loop(R, use_gmm, use_em, ng):
    from sr.recognition.batch import train_words
    words = _words(W=3, seed=8)
    np.random.seed(41)
    with quiet():
        seq = [R.HMM(5).fit([y.copy() for y in ys], ng, use_gmm=use_gmm, use_em=use_em) for ys in words]
    np.random.seed(41)
    with quiet():
        par = train_words(words, 5, n_gaussians=ng, use_gmm=use_gmm, use_em=use_em)
    assert len(par) == len(seq)
    for a, b in zip(par, seq):
        np.testing.assert_array_equal(a.mu, b.mu)
        np.testing.assert_array_equal(a.sigma, b.sigma)
        np.testing.assert_array_equal(a.transitions, b.transitions)
        assert a.use_gmm == b.use_gmm and a.use_em == b.use_em and a.n_segments == b.n_segments
        assert [len(s) for s in a.segments] == [len(s) for s in b.segments]
        for sa, sb in zip(a.segments, b.segments):
            np.testing.assert_array_equal(sa, sb)
        if use_gmm:
            (ma, va, wa), (mb, vb, wb) = pack_hmm(a), pack_hmm(b)
            np.testing.assert_allclose(ma, mb, rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(va, vb, rtol=1e-9)
            np.testing.assert_allclose(wa, wb, rtol=1e-9)
            x = words[0][0]
            np.testing.assert_allclose(a.evaluate(x), b.evaluate(x), rtol=1e-9)


def test_train_words_rejects_short_templates(R):
    from sr.recognition.batch import train_words
    words = _words(W=2, seed=1)
    words[1][0] = words[1][0][:4]
    with pytest.raises(NameError):
        with quiet():
            train_words(words, 2)
