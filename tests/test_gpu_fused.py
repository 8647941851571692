# -*- coding: utf-8 -*-
"""GPU parity tests of the FUSED single-Gaussian decode (gh_viterbi_fused: distance + dynamic program in one sweep, no
[N, S] likelihood matrix) -- the reference's config-1 path, `HMM.evaluate` of a model with one Gaussian per state
(sr/recognition/hmm.py:131-135: `dtw(x, self.mu, mahalanobis, self.transitions, self.sigma)`, or `decode_hmm_states` over
one-component mixtures) -- against the reference's own golden vectors (G3 c1, G5, G9) and the CPU oracle.

Tolerances: costs fp64 1e-10 relative, fp32 1e-3; paths, end choices and recognised words bit-exact.
"""
import contextlib
import io
import warnings

import numpy as np
import pytest

from conftest import load_golden
from oracle import ref_numpy as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from sr.recognition import _hip
    return _hip


@pytest.fixture(scope="module")
def ctx(hip):
    return hip.default_context()


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        yield


def graph(row_state, trans, start_rows, end_rows):
    to, frm = np.nonzero(~np.isinf(trans))
    return dict(row_state=row_state, arc_to=to, arc_from=frm, arc_cost=trans[to, frm], start_rows=start_rows, end_rows=end_rows)


def assert_costs(got, ref, rtol):
    fin = ~np.isinf(ref)
    np.testing.assert_array_equal(np.isinf(got), ~fin)
    np.testing.assert_allclose(got[fin], ref[fin], rtol=rtol)


def stacked(W, n, chains):
    big = np.full((W * n, W * n), np.inf)
    for i in range(W):
        big[i * n:(i + 1) * n, i * n:(i + 1) * n] = chains[i]
    return big


# ------------------------------------------------------------------ the reference's own outputs
def test_fused_decode_golden_g3_c1(hip, ctx):
    """G3 c1 (10 word models x 5 states, ONE Gaussian, 13 dims -- BASELINE configs[0]): full cost matrices and paths of
    `decode_hmm_states` per word, `HMM.evaluate` of all words and the recognised word, through the fused kernel."""
    g = load_golden("G3_isolated_decode_c1")
    means, vars_, w, trans = g["means"], g["vars"], g["w"], g["trans"]
    W, n, M, D = means.shape
    assert M == 1
    U = len(g["words"])
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, M, D), vars_.reshape(W * n, M, D), w.reshape(W * n, M))
    xs = [g["x%d" % u] for u in range(U)]
    b = hip.Batch(ctx, xs)
    for i in range(W):
        lat = hip.Lattices(ctx, [graph(np.arange(n) + i * n, trans, [0], [n - 1])])
        r = lat.viterbi(b, want_costs=True, fused_gmm=gmm)
        assert ctx.last_fused
        for u in range(U):
            assert_costs(r["costs"][u], g["costs_%d_%d" % (u, i)], 1e-10)
            np.testing.assert_array_equal(r["paths"][u], g["path_%d_%d" % (u, i)])
    st = hip.Lattices(ctx, [graph(np.arange(W * n), stacked(W, n, [trans] * W), [i * n for i in range(W)],
                                  [i * n + n - 1 for i in range(W)])])
    for want_path in (False, True):
        r = st.viterbi(b, want_path=want_path, fused_gmm=gmm)
        assert ctx.last_fused
        for u in range(U):
            np.testing.assert_allclose(r["end_cost"][u], g["evaluate_%d" % u], rtol=1e-10)
            assert int(np.argmin(r["end_cost"][u])) == int(g["words"][u])
            if want_path:      # the winner's path == the reference's path of that word model, rows shifted into the stack
                k = int(r["best_end"][u])
                np.testing.assert_array_equal(r["paths"][u] - [k * n, 0], g["path_%d_%d" % (u, k)])
    assert b.h and not hip.load_library().gh_loglik_dev_ptr(b.h), "the fused decode must not materialise the [N, S] matrix"


def test_fused_decode_golden_g5_dtw_mahalanobis(hip, ctx):
    """G5: the reference's `dtw(x, y, mahalanobis, trans, var)` -- the single-Gaussian branch of HMM.evaluate
    (hmm.py:133-134) -- cost matrix and path."""
    g = load_golden("G5_dtw")
    x, y, var = g["x"], g["y"], g["var"]
    n = len(y)
    gmm = hip.PackedGMM(ctx, y[:, None, :], var[:, None, :], np.ones((n, 1)))
    b = hip.Batch(ctx, [x])
    # (G5's skip-transition template carries NaN / -inf costs from an empty segment: dtw-only semantics, gh_dtw's business)
    for trans, costs, path in ((g["trans"], g["costs_mahal"], g["path_mahal"]),):
        lat = hip.Lattices(ctx, [graph(np.arange(n), trans, [0], [n - 1])])
        r = lat.viterbi(b, want_costs=True, fused_gmm=gmm, log_domain=True)
        assert ctx.last_fused
        assert_costs(r["costs"][0], costs, 1e-10)
        np.testing.assert_array_equal(r["paths"][0], path)


def test_recognizer_takes_use_gmm_false_models_golden_g9(hip, ctx):
    """`HMM.fit(use_gmm=False)` leaves gmm_states = None (hmm.py:57-76); core.test's loop (`[m.evaluate(x) for m in
    models]`, sr/core.py:82-87) scores such models with dtw + mahalanobis.  IsolatedWordRecognizer takes them: the G9
    model (trained by the reference) next to two perturbed copies; costs == HMM.evaluate of the mirror == the oracle's
    restatement of the reference's dtw."""
    import sr.recognition as R
    from sr.recognition.batch import IsolatedWordRecognizer
    g = load_golden("G9_hmm_fit_single")
    n = int(g["n"])
    rng = np.random.default_rng(9)
    models = []
    for k in range(3):
        h = R.HMM(n)
        h.use_gmm = False
        h.mu = g["mu"] + 0.3 * k * rng.normal(size=g["mu"].shape)
        h.sigma = g["sigma"] * (1.0 + 0.2 * k)
        h.transitions = g["transitions"].copy()
        assert h.gmm_states is None
        models.append(h)
    xs = [g["y%d" % i] for i in range(6)]
    rec = IsolatedWordRecognizer(models)
    words, costs = rec.recognize(xs)
    assert rec.ctx.last_fused
    for u, x in enumerate(xs):
        for k, h in enumerate(models):
            ref = O.hmm_evaluate(x, dict(n_segments=n, mu=h.mu, sigma=h.sigma, transitions=h.transitions), use_gmm=False)
            np.testing.assert_allclose(costs[u, k], ref, rtol=1e-10)
            with quiet():
                np.testing.assert_allclose(costs[u, k], h.evaluate(x), rtol=1e-10)
    np.testing.assert_array_equal(words, np.argmin(costs, axis=1))
    acc, got = rec.accuracy(xs, np.zeros(len(xs), dtype=int))
    assert acc == 1.0, got      # the training templates of the G9 model are closest to the G9 model


def test_recognizer_on_models_fit_without_mixtures(hip, ctx):
    """End to end: three words trained by `HMM.fit(ys, 1, use_gmm=False)` (segmental k-means only) and recognised by the
    batch driver: the cheapest model per utterance is the one HMM.evaluate picks."""
    import sr.recognition as R
    from sr.recognition.batch import IsolatedWordRecognizer
    rng = np.random.default_rng(21)
    W, n, D = 3, 4, 13
    protos = rng.normal(size=(W, n, D)) * 2.0
    def sample(wd, T):
        seg = np.minimum(np.arange(T) * n // T, n - 1)
        return protos[wd, seg] + 0.4 * rng.normal(size=(T, D))
    models = []
    with quiet():
        for wd in range(W):
            models.append(R.HMM(n).fit([sample(wd, int(T)) for T in rng.integers(20, 40, size=6)], 1, use_gmm=False))
    xs, truth = [], []
    for wd in range(W):
        for T in (17, 32, 33, 64, 65, 90):
            xs.append(sample(wd, T))
            truth.append(wd)
    rec = IsolatedWordRecognizer(models)
    words, costs = rec.recognize(xs)
    assert rec.ctx.last_fused
    with quiet():
        ref = np.array([[m.evaluate(x) for m in models] for x in xs])
    np.testing.assert_allclose(costs, ref, rtol=1e-10)
    np.testing.assert_array_equal(words, np.argmin(ref, axis=1))
    np.testing.assert_array_equal(words, truth)


# ------------------------------------------------------------------ against the oracle, shapes and edges
@pytest.mark.parametrize("D", [3, 8, 13, 16, 20, 26, 39, 40])
@pytest.mark.parametrize("dtype,rtol", [(np.float64, 1e-10), (np.float32, 1e-3)])
def test_fused_vs_oracle_shapes(hip, ctx, D, dtype, rtol):
    """Every instantiation (D = 13 / 39 exact, padded otherwise), chains with skip transitions, 15 chains of 6 rows =
    90 rows (two lane groups), utterance lengths around the 32-frame tile, weights != 1, both feature dtypes:
    costs / paths / end choices against the oracle (fp32: costs only)."""
    rng = np.random.default_rng(100 + D)
    W, n = 15, 6
    S = W * n
    means, vars_ = rng.normal(size=(S, 1, D)), rng.uniform(0.5, 1.5, size=(S, 1, D))
    w = rng.uniform(0.2, 1.0, size=(S, 1))
    gmm = hip.PackedGMM(ctx, means, vars_, w)
    Ts = (2, 31, 32, 33, 63, 64, 65, 97, 5)
    xs = [means[rng.integers(0, S, size=T), 0] + rng.normal(size=(T, D)) for T in Ts]
    b = hip.Batch(ctx, xs, dtype=dtype)
    X = np.concatenate(xs)
    if dtype == np.float32:
        X = X.astype(np.float32).astype(np.float64)
    nll = O.gmm_neg_loglik_batch(X, means, vars_, w)

    def chain(skip):
        t = np.full((n, n), np.inf)
        for i in range(n):
            t[i, i] = rng.uniform(0.05, 1.0)
            if i + 1 < n:
                t[i + 1, i] = rng.uniform(0.5, 3.0)
            if skip and i + 2 < n and rng.random() < 0.5:
                t[i + 2, i] = rng.uniform(0.5, 3.0)
        return t

    for skip in (False, True):
        chains = [chain(skip and k % 2 == 0) for k in range(W)]
        rows = rng.permutation(S)
        st = hip.Lattices(ctx, [graph(rows, stacked(W, n, chains), [k * n for k in range(W)], [k * n + n - 1 for k in range(W)])])
        r = st.viterbi(b, want_path=True, want_costs=True, fused_gmm=gmm)
        assert ctx.last_fused
        r2 = st.viterbi(b, want_path=False, fused_gmm=gmm)              # the variant without back-pointers
        np.testing.assert_array_equal(r2["end_cost_flat"], r["end_cost_flat"])
        np.testing.assert_array_equal(r2["best_end"], r["best_end"])
        for u in range(len(xs)):
            ref = []
            for k in range(W):
                E = nll[b.offsets[u]:b.offsets[u + 1]][:, rows[k * n:(k + 1) * n]].T
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    ref.append(O.decode_states(E, np.zeros(n, dtype=bool), chains[k]))
            full = np.concatenate([c for c, _ in ref])
            assert_costs(r["costs"][u], full, rtol)
            ends = np.array([c[-1, -1] for c, _ in ref])
            assert_costs(r["end_cost"][u], ends, rtol)
            if dtype == np.float64:
                best, bk = np.inf, -1
                for k, c in enumerate(ends):
                    if best >= c:          # decode.py:129-134: the LAST of equal minima
                        best, bk = c, k
                assert r["best_end"][u] == bk
                if np.isfinite(best):
                    np.testing.assert_array_equal(r["paths"][u] - [bk * n, 0], ref[bk][1].reshape(-1, 2))


def test_fused_equals_two_kernel_form_and_fallbacks(hip, ctx, monkeypatch):
    """gh_viterbi_fused always answers: shapes the fused kernel does not take (a single-frame utterance: the reference's
    column wrap; GMMHMM_FUSED=0; several components) run gh_loglik + gh_viterbi inside the call; on shapes it takes, both
    forms agree to rounding with identical paths."""
    rng = np.random.default_rng(77)
    W, n, D = 10, 5, 13
    S = W * n
    means, vars_ = rng.normal(size=(S, 1, D)), rng.uniform(0.5, 1.5, size=(S, 1, D))
    gmm = hip.PackedGMM(ctx, means, vars_, np.ones((S, 1)))
    t = np.full((n, n), np.inf)
    for i in range(n):
        t[i, i] = -np.log(0.9) if i < n - 1 else 0.0
        if i < n - 1:
            t[i + 1, i] = -np.log(0.1)
    st = hip.Lattices(ctx, [graph(np.arange(S), stacked(W, n, [t] * W), [k * n for k in range(W)], [k * n + n - 1 for k in range(W)])])
    xs = [means[(u % W) * n + np.minimum(np.arange(T) * n // T, n - 1), 0] + rng.normal(size=(T, D))
          for u, T in enumerate(rng.integers(20, 130, size=300))]
    b = hip.Batch(ctx, xs)
    f = st.viterbi(b, want_path=True, fused_gmm=gmm)
    assert ctx.last_fused
    monkeypatch.setenv("GMMHMM_FUSED", "0")
    two = st.viterbi(b, want_path=True, fused_gmm=gmm)
    assert not ctx.last_fused
    monkeypatch.delenv("GMMHMM_FUSED")
    np.testing.assert_allclose(f["end_cost_flat"], two["end_cost_flat"], rtol=1e-11)
    np.testing.assert_array_equal(f["best_end"], two["best_end"])
    for u in range(len(xs)):
        np.testing.assert_array_equal(f["paths"][u], two["paths"][u])
    np.testing.assert_array_equal(f["best_end"], np.arange(len(xs)) % W)
    # a single-frame utterance in the batch: the other kernels implement the reference's T == 1 semantics
    b1 = hip.Batch(ctx, xs[:3] + [xs[3][:1]])
    r1 = st.viterbi(b1, want_path=True, fused_gmm=gmm)
    assert not ctx.last_fused
    b1.loglik(gmm, fetch=False)
    r1b = st.viterbi(b1, want_path=True)
    np.testing.assert_array_equal(r1["end_cost_flat"], r1b["end_cost_flat"])
    # a mixture: two kernels
    gm2 = hip.PackedGMM(ctx, np.repeat(means, 2, axis=1), np.repeat(vars_, 2, axis=1), np.full((S, 2), 0.5))
    r2 = st.viterbi(b, want_path=False, fused_gmm=gm2)
    assert not ctx.last_fused
    np.testing.assert_allclose(r2["end_cost_flat"], f["end_cost_flat"], rtol=1e-11)


def test_fused_linear_domain_underflow_rule(hip, ctx):
    """GMM.evaluate works in the linear domain (hmm_state.py:114-120): a frame whose exp(-q/2) -- or whose weighted
    density -- rounds to 0 costs +inf, also with a normaliser > 1 (tight variances); mahalanobis()
    (log_domain=True, the use_gmm=False models) never underflows."""
    rng = np.random.default_rng(3)
    n, D = 4, 13
    var = np.full((n, 1, D), 0.01)                       # log normaliser = +18: the bare exponent underflows first
    means = rng.normal(size=(n, 1, D)) * 0.005
    gmm = hip.PackedGMM(ctx, means, var, np.ones((n, 1)))
    t = np.full((n, n), np.inf)
    for i in range(n):
        t[i, i] = 0.1
        if i + 1 < n:
            t[i + 1, i] = 1.0
    lat = hip.Lattices(ctx, [graph(np.arange(n), t, [0], [n - 1])])
    T = 12
    x = np.tile(means[0, 0], (T, 1)) + 0.05 * rng.normal(size=(T, D))
    # frame 5: -q/2 = -754 +- 3 for every state: log(norm * exp(-q/2)) = -736 is representable, but np.exp(-754) == 0
    d = rng.normal(size=D)
    x[5] = means[0, 0] + d / np.linalg.norm(d) * np.sqrt(2 * 754 * 0.01)
    b = hip.Batch(ctx, [x])
    with np.errstate(divide="ignore"):
        ref_lin = np.array([[O.gmm_evaluate(x[j], means[i], var[i], np.ones(1)) for j in range(T)] for i in range(n)])
    assert np.all(np.isinf(ref_lin[:, 5])) and np.all(np.isfinite(ref_lin[:, 4]))
    q5 = 0.5 * np.sum((x[5] - means[:, 0]) ** 2 / var[:, 0], axis=1)
    assert np.all(q5 > 746) and np.all(q5 - 17.9 < 745)     # the band the total-logarithm test alone would miss
    r = lat.viterbi(b, want_costs=True, fused_gmm=gmm)
    assert ctx.last_fused
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        costs, _ = O.decode_states(ref_lin, np.zeros(n, dtype=bool), t)
    assert np.all(np.isinf(costs[:, 5:]))                # unreachable from frame 5 on, as in the reference
    assert_costs(r["costs"][0], costs, 1e-10)
    rl = lat.viterbi(b, want_costs=True, fused_gmm=gmm, log_domain=True)
    E = O.distance_matrix(x, means[:, 0], "mahalanobis", var[:, 0])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        costs_log, _ = O.dtw(E, t)
    assert np.isfinite(costs_log[-1, -1])
    assert_costs(rl["costs"][0], costs_log, 1e-10)


def test_fused_at_scale_equals_two_kernels_on_every_utterance(hip, ctx, monkeypatch):
    """Size-independent property at 20 000 utterances of the configs[0] model (2 M frames): the fused sweep and the
    two-kernel form recognise the same word on every utterance, with end costs equal to rounding; the serpentine
    work distribution covers every utterance exactly once (no cost left at its initial value)."""
    rng = np.random.default_rng(1000)
    W, n, D = 10, 5, 13
    S = W * n
    means, vars_ = rng.normal(size=(S, 1, D)), rng.uniform(0.5, 1.5, size=(S, 1, D))
    gmm = hip.PackedGMM(ctx, means, vars_, np.ones((S, 1)))
    t = np.full((n, n), np.inf)
    for i in range(n):
        t[i, i] = -np.log(0.9) if i < n - 1 else 0.0
        if i < n - 1:
            t[i + 1, i] = -np.log(0.1)
    st = hip.Lattices(ctx, [graph(np.arange(S), stacked(W, n, [t] * W), [k * n for k in range(W)], [k * n + n - 1 for k in range(W)])])
    U = 20000
    lens = rng.integers(50, 151, size=U)
    words = rng.integers(0, W, size=U)
    off = np.concatenate([[0], np.cumsum(lens)])
    seg = np.concatenate([np.minimum(np.arange(T) * n // T, n - 1) for T in lens])
    feats = means[np.repeat(words, lens) * n + seg, 0] + np.sqrt(vars_[np.repeat(words, lens) * n + seg, 0]) * rng.normal(size=(off[-1], D))
    b = hip.Batch(ctx, feats=feats, offsets=off)
    f = st.viterbi(b, want_path=False, fused_gmm=gmm)
    assert ctx.last_fused
    monkeypatch.setenv("GMMHMM_FUSED", "0")
    two = st.viterbi(b, want_path=False, fused_gmm=gmm)
    monkeypatch.delenv("GMMHMM_FUSED")
    assert np.all(np.isfinite(f["end_cost_flat"]))
    np.testing.assert_allclose(f["end_cost_flat"], two["end_cost_flat"], rtol=1e-11)
    np.testing.assert_array_equal(f["best_end"], two["best_end"])
    got = np.argmin(f["end_cost_flat"].reshape(U, W), axis=1)
    assert np.mean(got == words) > 0.999
