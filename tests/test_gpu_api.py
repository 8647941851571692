# -*- coding: utf-8 -*-
"""GPU parity tests of the `sr.recognition` mirror API (objects in, reference-shaped
results out) against the golden vectors captured from the reference itself.
These read like tests the reference could have had for its own package."""
import contextlib
import io
import os
import pickle
import warnings

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    import sr.recognition as R
    return R


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        yield


def make_gmm(R, means, vars_, w):
    g = R.GMM(means[0].copy(), vars_[0].copy(), len(w))
    g.update_models(means.copy(), vars_.copy(), w.copy())
    return g


def make_hmm(R, means, vars_, w, trans):
    h = R.HMM(means.shape[0])
    h.gmm_states = [make_gmm(R, means[s], vars_[s], w[s]) for s in range(means.shape[0])]
    h.transitions = trans.copy()
    h.mu, h.sigma = means[:, 0].copy(), vars_[:, 0].copy()
    return h


def pack_hmm(h):
    return (np.array([[d.mean for d in g.dists] for g in h.gmm_states]),
            np.array([[d.cov for d in g.dists] for g in h.gmm_states]),
            np.array([g.w for g in h.gmm_states]))


def assert_costs(got, ref, rtol=1e-10):
    fin = ~np.isinf(ref)
    np.testing.assert_array_equal(np.isinf(got), ~fin)
    np.testing.assert_allclose(got[fin], ref[fin], rtol=rtol)


# ------------------------------------------------------------------ A1-A4, A8
def test_gmm_evaluate_and_pdf(R):
    g = load_golden("G1_gmm_evaluate_m8d39")
    for s in (0, 17, 49):
        st = make_gmm(R, g["means"][s], g["vars"][s], g["w"][s])
        for i in (0, 5, 63):
            np.testing.assert_allclose(st.evaluate(g["X"][i]), g["nll"][i, s], rtol=1e-10)
            np.testing.assert_allclose(st.evaluate(g["X"][i], False), g["comp"][i, s], rtol=1e-9)
        np.testing.assert_allclose(st.evaluate_batch(g["X"]), g["nll"][:, s], rtol=1e-10)
        d = st.dists[3]
        np.testing.assert_allclose(d.pdf(g["X"][2]) * st.w[3], g["comp"][2, s, 3], rtol=1e-9)
    with pytest.raises(NameError):
        st.dists[0].pdf(np.zeros(5))
    with pytest.raises(np.linalg.LinAlgError):
        R.MultivariateNormal(np.zeros(39), np.ones(39)).cov = np.zeros(39)
    assert R.NES().evaluate(g["X"][0]) == 0
    a, b = R.NES(), R.NES()
    assert a == a and not (a == b) and len({a, b}) == 2
    assert st == make_gmm(R, g["means"][49], g["vars"][49], g["w"][49]) and len(st) == 8


def test_mahalanobis(R):
    g = load_golden("G2_mahalanobis")
    out = [R.mahalanobis(a, b, c) for a, b, c in zip(g["v1"][:8], g["v2"][:8], g["var"][:8])]
    np.testing.assert_allclose(out, g["out"][:8], rtol=1e-13)


# ------------------------------------------------------------------------- A6
@pytest.mark.parametrize("tag", ["c1", "c2"])
def test_decode_hmm_states_isolated(R, tag):
    g = load_golden("G3_isolated_decode_" + tag)
    W = g["means"].shape[0]
    hmms = [make_hmm(R, g["means"][i], g["vars"][i], g["w"][i], g["trans"]) for i in range(W)]
    for u in range(len(g["words"])):
        x = g["x%d" % u]
        for i in (0, W - 1):
            costs, path = R.decode_hmm_states(x, hmms[i].gmm_states, hmms[i].transitions)
            assert_costs(costs, g["costs_%d_%d" % (u, i)])
            np.testing.assert_array_equal(path, g["path_%d_%d" % (u, i)])
            assert path.dtype == np.int64 and costs.dtype == np.float64
        np.testing.assert_allclose([h.evaluate(x) for h in hmms], g["evaluate_%d" % u], rtol=1e-10)
    from sr.recognition.batch import IsolatedWordRecognizer
    xs = [g["x%d" % u] for u in range(len(g["words"]))]
    words, costs = IsolatedWordRecognizer(hmms).recognize(xs)
    np.testing.assert_array_equal(words, g["words"])
    np.testing.assert_allclose(costs, [g["evaluate_%d" % u] for u in range(len(xs))], rtol=1e-10)
    # core.py:63-94 as a call: the labels are the reference's own decisions here, so everything passes
    acc, got = IsolatedWordRecognizer(hmms).accuracy(xs, g["words"])
    assert acc == 1.0 and list(got) == list(g["words"])
    wrong = (np.asarray(g["words"]) + 1) % W
    assert IsolatedWordRecognizer(hmms).accuracy(xs, wrong)[0] == 0.0


def test_build_state_sequences_and_lattice_decode(R):
    g = load_golden("G4_lattice_decode")
    W, n = g["means"].shape[:2]
    hmms = [make_hmm(R, g["means"][i], g["vars"][i], g["w"][i], g["word_trans"]) for i in range(W)]
    from sr.recognition.batch import ContinuousDecoder, path_to_words
    for K in (1, 2, 3, 7):
        p = "K%d_" % K
        seq, trans, ends = R.build_state_sequences(hmms, [list(range(W))] * K)
        assert len(seq) == int(g[p + "R"]) and list(ends) == list(g[p + "ends"])
        ref_t = np.full_like(trans, np.inf)
        ref_t[g[p + "arc_to"], g[p + "arc_from"]] = g[p + "arc_cost"]
        np.testing.assert_array_equal(trans, ref_t)
        assert all((type(s) is R.NES) == (w < 0) for s, w in zip(seq, g[p + "row_word"]))
        assert seq[1] is seq[1 + (W * n + 1)] if K > 1 else True  # state objects shared between layers
        with quiet():
            costs, path = R.decode_hmm_states(g[p + "x"], seq, trans, end_points=[[e, -1] for e in ends])
        assert_costs(costs, g[p + "costs"])
        np.testing.assert_array_equal(path, g[p + "path"])
        dec = ContinuousDecoder(hmms, n_layers=K)
        assert dec.decode([g[p + "x"]])[0] == list(g[p + "digits"])
        assert path_to_words(path, dec.row_state, n) == list(g[p + "digits"])
        rep = dec.accuracy([g[p + "x"]] * 2, [list(g[p + "digits"]), [(d + 1) % W for d in g[p + "digits"]]])   # main.py:69-84
        assert rep["sequence_accuracy"] == 0.5 and rep["n_digit_errors"] == K and rep["digit_accuracy"] == 0.5
    labels = list(g["forced_labels"])
    seq, trans, ends = R.build_state_sequences(hmms, [[l] for l in labels])
    costs, path = R.decode_hmm_states(g["forced_x"], seq, trans, end_points=[[e, -1] for e in ends])
    assert_costs(costs, g["forced_costs"])
    np.testing.assert_array_equal(path, g["forced_path"])


@pytest.mark.parametrize("pen", [0, 1])
def test_loop_grammar_decode(R, pen):
    """N4: `build_loop_grammar` lays the rows out so that the reference's decode_hmm_states decodes a word loop;
    golden G14 holds what the reference computes on that graph."""
    g = load_golden("G14_loop_grammar")
    W, n = g["means"].shape[:2]
    hmms = [make_hmm(R, g["means"][i], g["vars"][i], g["w"][i], g["word_trans"]) for i in range(W)]
    from sr.recognition.batch import ContinuousDecoder, path_to_words
    pp = "p%d_" % pen
    penalty = float(g[pp + "penalty"])
    seq, trans, ends = R.build_loop_grammar(hmms, word_penalty=penalty)
    ref_t = np.full_like(trans, np.inf)
    ref_t[g[pp + "arc_to"], g[pp + "arc_from"]] = g[pp + "arc_cost"]
    np.testing.assert_array_equal(trans, ref_t)
    assert list(ends) == list(g[pp + "ends"])
    assert all((type(s) is R.NES) == (wd < 0) for s, wd in zip(seq, g[pp + "row_word"]))
    dec = ContinuousDecoder(hmms, grammar="loop", word_penalty=penalty)
    xs = [g[pp + "x%d" % u] for u in range(int(g["n_utts"]))]
    for u, x in enumerate(xs):
        with quiet():
            costs, path = R.decode_hmm_states(x, seq, trans, end_points=[[e, -1] for e in ends])
        assert_costs(costs, g[pp + "costs%d" % u])
        np.testing.assert_array_equal(path, g[pp + "path%d" % u])
        assert path_to_words(path, dec.row_state, n) == list(g[pp + "digits%d" % u])
        if pen == 0:
            np.testing.assert_allclose(min(costs[e, -1] for e in ends), np.min(g["layer_costs%d" % u]), rtol=1e-12)
    assert dec.decode(xs) == [list(g[pp + "digits%d" % u]) for u in range(len(xs))]
    with pytest.raises(ValueError):
        ContinuousDecoder(hmms, grammar="ring")


def test_decode_edges(R):
    g = load_golden("G6_decode_edges")
    h = make_hmm(R, g["means"], g["vars"], g["w"], g["trans"])
    with quiet():
        c, p = R.decode_hmm_states(g["t1_x"], h.gmm_states, h.transitions)
    np.testing.assert_allclose(c, g["t1_costs"], rtol=1e-10)
    assert tuple(p.shape) == tuple(g["t1_path_shape"])
    with pytest.warns(UserWarning):
        c, p = R.decode_hmm_states(g["t2_x"], h.gmm_states, h.transitions)
    assert_costs(c, g["t2_costs"])
    np.testing.assert_array_equal(p, g["t2_path"])
    st = h.gmm_states[:4] + [h.gmm_states[3]]
    c, p = R.decode_hmm_states(g["tie_x"], st, g["tie_trans"], end_points=[[3, -1], [4, -1]])
    assert_costs(c, g["tie_costs"])
    np.testing.assert_array_equal(p, g["tie_path"])
    c, p = R.decode_hmm_states(g["tie_x"], st, g["tie_trans"], end_points=[[4, -1], [3, -1]])
    np.testing.assert_array_equal(p, g["tie_path_rev"])
    st3 = [h.gmm_states[0], h.gmm_states[1], h.gmm_states[1]]
    c, p = R.decode_hmm_states(g["tie_x"], st3, g["ptie_trans"])
    assert_costs(c, g["ptie_costs"])
    np.testing.assert_array_equal(p, g["ptie_path"])
    # an end point in an inner column == decoding the prefix
    c_in, p_in = R.decode_hmm_states(g["tie_x"], h.gmm_states, h.transitions, end_points=[[3, 6]])
    c_pre, p_pre = R.decode_hmm_states(g["tie_x"][:7], h.gmm_states, h.transitions, end_points=[[3, -1]])
    np.testing.assert_array_equal(p_in, p_pre)
    np.testing.assert_allclose(c_in[:, :7][~np.isinf(c_pre)], c_pre[~np.isinf(c_pre)], rtol=1e-12)


# ------------------------------------------------------------------------- A5
def test_dtw(R):
    g = load_golden("G5_dtw")
    from sr.recognition.hmm_state import euclidean
    x, y, var, trans = g["x"], g["y"], g["var"], g["trans"]
    cases = [("euclid", euclidean, trans, None, np.inf), ("mahal", R.mahalanobis, trans, var, np.inf),
             ("beam3", R.mahalanobis, trans, var, 3), ("beam2", euclidean, trans, None, 2),
             ("skip", R.mahalanobis, g["trans_skip"], var, np.inf),
             ("euclid", lambda *a: np.linalg.norm(a[0] - a[1]), trans, None, np.inf)]  # arbitrary callable
    for tag, fn, tr, v, beam in cases:
        c, p = R.dtw(x, y, fn, tr, v, beam=beam)
        assert_costs(c, g["costs_" + tag])
        np.testing.assert_array_equal(p, g["path_" + tag])
    with pytest.raises(AssertionError):
        R.dtw(x[:1], y, euclidean, trans)


# ------------------------------------------------------------------------- A7
@pytest.mark.parametrize("k", [2, 3])
@pytest.mark.parametrize("tag,iters", [("it1", 1), ("conv", 10000)])
def test_gmm_em(R, k, tag, iters):
    g = load_golden("G7_gmm_em")
    st = R.GMM(g["mu0"].copy(), g["var0"].copy(), len(g["init_w"]))
    st.update_models(g["init_means"].copy(), g["init_vars"].copy(), g["init_w"].copy())
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        st.em(g["data"], k, max_iteration=iters)
    txt = buf.getvalue()
    n_it = int(txt.rsplit("EM converged at iteration:", 1)[1].split()[0]) + 1 if "converged" in txt else iters
    p = "k%d_%s_" % (k, tag)
    assert n_it == int(g[p + "iters"])
    np.testing.assert_allclose(np.array([d.mean for d in st.dists]), g[p + "means"], rtol=1e-8)
    np.testing.assert_allclose(np.array([d.cov for d in st.dists]), g[p + "vars"], rtol=1e-8)
    np.testing.assert_allclose(st.w, g[p + "w"], rtol=1e-8)
    np.testing.assert_allclose(st.mu_old, g[p + "mu_old"], rtol=1e-8)


@pytest.mark.parametrize("k", [2, 3])
def test_gmm_em_linear_domain_underflow(R, k):
    """the E-step keeps the reference's linear-domain semantics -- a frame whose every weighted density
    rounds to 0 moves nothing, an underflowed component gets no share (golden G16, captured from the reference)."""
    g = load_golden("G16_gmm_em_underflow")
    for tag, iters in (("it1", 1), ("it3", 3)):
        st = R.GMM(g["mu0"].copy(), g["var0"].copy(), len(g["init_w"]))
        st.update_models(g["init_means"].copy(), g["init_vars"].copy(), g["init_w"].copy())
        with contextlib.redirect_stdout(io.StringIO()):
            st.em(g["data"], k, max_iteration=iters)
        np.testing.assert_allclose(np.array([d.mean for d in st.dists]), g["k%d_%s_means" % (k, tag)], rtol=1e-8)
        np.testing.assert_allclose(np.array([d.cov for d in st.dists]), g["k%d_%s_vars" % (k, tag)], rtol=1e-8)
        np.testing.assert_allclose(st.w, g["k%d_%s_w" % (k, tag)], rtol=1e-8)


# ------------------------------------------------------------------------ A14
@pytest.mark.parametrize("tag,k,dist", [("k2m", 2, "m"), ("k4m", 4, "m"), ("k4e", 4, "e")])
def test_kmeans(R, tag, k, dist):
    g = load_golden("G8_kmeans")
    np.random.seed(0)
    if dist == "e":
        cl, ce, cov = R.kmeans(g["data"], k, g[tag + "_c0"].copy())
    else:
        cl, ce, cov = R.kmeans(g["data"], k, g[tag + "_c0"].copy(), dist_fun=R.mahalanobis)
    np.testing.assert_array_equal(cl, g[tag + "_clusters"])
    np.testing.assert_allclose(ce, g[tag + "_centroids"], rtol=0)
    np.testing.assert_allclose(cov, g[tag + "_cov"], rtol=0)


# ------------------------------------------------------------------- A15 / A9
def _ys(g):
    return [g["y%d" % i] for i in range(int(g["n"]))]


def test_hmm_fit_single_gaussian(R):
    g = load_golden("G9_hmm_fit_single")
    with quiet():
        h = R.HMM(5).fit(_ys(g), 1, use_gmm=False)
    np.testing.assert_allclose(h.mu, g["mu"], rtol=1e-12)
    np.testing.assert_allclose(h.sigma, g["sigma"], rtol=1e-12)
    np.testing.assert_allclose(h.transitions, g["transitions"], rtol=1e-12)
    assert [len(s) for s in h.segments] == list(g["seg_lens"])
    for i, s in enumerate(h.segments):
        np.testing.assert_array_equal(s, g["seg%d" % i])
    np.testing.assert_array_equal(R.get_segments_from_path(g["gsp_path"], 5), g["gsp_out"])
    np.testing.assert_allclose(R.calc_transition_costs(2, g["ctc_lens"]), g["ctc_out"], rtol=0)
    np.testing.assert_allclose(h.evaluate(g["y0"]),
                               R.dtw(g["y0"], h.mu, R.mahalanobis, h.transitions, h.sigma)[0][-1, -1])
    with pytest.raises(NameError):
        R.skmeans([g["y0"][:4]], 2)  # 2 frames per segment: fits, but < 5 frames (kmeans.py:142)


@pytest.mark.parametrize("ng,em", [(4, True), (8, True), (4, False)])
def test_hmm_fit_gmm(R, ng, em):
    g = load_golden("G10_hmm_fit_gmm")
    tag = "g%d_%s" % (ng, "em" if em else "km")
    ys = _ys(g)
    np.random.seed(5)
    with quiet():
        h = R.HMM(5).fit([y.copy() for y in ys], ng, use_gmm=True, use_em=em)
    m, v, w = pack_hmm(h)
    np.testing.assert_allclose(h.mu, g[tag + "_mu"], rtol=1e-12)
    np.testing.assert_allclose(h.transitions, g[tag + "_transitions"], rtol=1e-12)
    np.testing.assert_allclose(m, g[tag + "_means"], rtol=1e-7)
    np.testing.assert_allclose(v, g[tag + "_vars"], rtol=1e-7)
    np.testing.assert_allclose(w, g[tag + "_w"], rtol=1e-7)
    assert [len(s) for s in h.segments] == list(g[tag + "_seg_lens"])
    np.testing.assert_allclose([h.evaluate(y) for y in ys], g[tag + "_evaluate"], rtol=1e-7)
    assert h[0] is h.gmm_states[0] and h == h
    with pytest.raises(TypeError):
        h["a"]
    with pytest.raises(AssertionError):
        R.HMM(5)._fit_GMM(ys[0], 2, 0)  # int(ln 2) == 0 splits (hmm.py:104-105)


# ------------------------------------------------------------------------ A11
@pytest.mark.parametrize("iters", [1, 3])
def test_continuous_train(R, iters, tmp_path):
    g = load_golden("G11_continuous_train")
    W, U = int(g["n_words"]), int(g["n_utts"])
    data = [g["x%d" % i] for i in range(U)]
    labels = [list(g["labels%d" % i]) for i in range(U)]
    models = []
    for wi in range(W):
        h = make_hmm(R, g["init%d_means" % wi], g["init%d_vars" % wi], g["init%d_w" % wi],
                     g["init%d_transitions" % wi])
        for s, st in enumerate(h.gmm_states):
            st.mu_old[:] = g["init%d_mu_old" % wi][s]
            st.sigma_old[:] = g["init%d_sigma_old" % wi][s]
            st.w_old[:] = g["init%d_w_old" % wi][s]
            st.parent = h
        models.append(h)
    np.random.seed(9)
    with quiet():
        R.continuous_train(data, models, labels, str(tmp_path), n_gaussians=4, n_segments=5, max_iteration=iters)
    for wi in range(W):
        with open(os.path.join(str(tmp_path), "%d.pkl" % wi), "rb") as f:
            h = pickle.load(f)
        assert sorted(h.__dict__) == ["gmm_states", "mu", "n_segments", "segments", "sigma", "transitions",
                                      "use_em", "use_gmm"]
        assert sorted(h.gmm_states[0].__dict__) == ["dists", "id", "mu_old", "n_gaussians", "parent", "sigma_old",
                                                    "w", "w_old"]
        assert sorted(h.gmm_states[0].dists[0].__dict__) == ["_cov", "inv_cov", "mean"]
        m, v, w = pack_hmm(h)
        p = "it%d_%d_" % (iters, wi)
        np.testing.assert_allclose(m, g[p + "means"], rtol=1e-6)
        np.testing.assert_allclose(v, g[p + "vars"], rtol=1e-6)
        np.testing.assert_allclose(w, g[p + "w"], rtol=1e-6)
        np.testing.assert_allclose(h.transitions, g[p + "transitions"], rtol=1e-9)


@pytest.mark.parametrize("iters", [1, 2])
def test_continuous_train_8mix(R, iters, tmp_path):
    """G17: the reference's continuous_train at 13 dims / 8 mixtures (three binary splits) / 4 words of 3 states, strings
    of 2-4 words with repeats -- the whole device pipeline (own-state likelihoods, forced alignment + regrouping,
    lock-step split k-means with device centroid sums, EM) against the models the reference wrote."""
    g = load_golden("G17_continuous_train_8mix")
    W, U, ng, nseg = int(g["n_words"]), int(g["n_utts"]), int(g["n_gaussians"]), int(g["n_segments"])
    data = [g["x%d" % i] for i in range(U)]
    labels = [list(g["labels%d" % i]) for i in range(U)]
    models = []
    for wi in range(W):
        h = make_hmm(R, g["init%d_means" % wi], g["init%d_vars" % wi], g["init%d_w" % wi],
                     g["init%d_transitions" % wi])
        for s, st in enumerate(h.gmm_states):
            st.mu_old[:] = g["init%d_mu_old" % wi][s]
            st.sigma_old[:] = g["init%d_sigma_old" % wi][s]
            st.w_old[:] = g["init%d_w_old" % wi][s]
            st.parent = h
        models.append(h)
    np.random.seed(19)
    with quiet():
        R.continuous_train(data, models, labels, str(tmp_path), n_gaussians=ng, n_segments=nseg, max_iteration=iters)
    for wi in range(W):
        with open(os.path.join(str(tmp_path), "%d.pkl" % wi), "rb") as f:
            h = pickle.load(f)
        m, v, w = pack_hmm(h)
        p = "it%d_%d_" % (iters, wi)
        np.testing.assert_allclose(m, g[p + "means"], rtol=1e-6)
        np.testing.assert_allclose(v, g[p + "vars"], rtol=1e-6)
        np.testing.assert_allclose(w, g[p + "w"], rtol=1e-6)
        np.testing.assert_allclose(h.transitions, g[p + "transitions"], rtol=1e-9)


# ------------------------------------------------------- A13 + 8(e): soft EM loop
def test_baum_welch_trainer_increases_likelihood(R):
    """Forward-backward E-step + statistics + M-step: the total log-likelihood is monotone
    (EM guarantee; transitions fixed), and the first iteration equals a numpy M-step computed
    from the same occupancies."""
    from sr.recognition.train import BaumWelchTrainer
    g = load_golden("G11_continuous_train")
    W, U = int(g["n_words"]), int(g["n_utts"])
    data = [g["x%d" % i] for i in range(U)]
    labels = [list(g["labels%d" % i]) for i in range(U)]
    means = np.array([g["init%d_means" % wi] for wi in range(W)])
    vars_ = np.array([g["init%d_vars" % wi] for wi in range(W)])
    w = np.array([g["init%d_w" % wi] for wi in range(W)])
    trans = [g["init%d_transitions" % wi] for wi in range(W)]
    tr = BaumWelchTrainer(means, vars_, w, trans, data, labels, var_floor=1e-3, update_transitions=False)
    # reference M-step for iteration 1 from the trainer's own E-step pieces
    stats, _xi, ll0 = tr.e_step()
    counts = stats[:, :, 0].sum(axis=1)
    from sr.recognition.parallel import m_step
    seen = counts > 0
    mu, sigma, wn = m_step(stats[seen], counts[seen], tr.means[seen])
    hist = tr.fit(6)
    assert abs(hist[0] - ll0) <= 1e-9 * abs(ll0)
    assert all(b >= a - 1e-7 * abs(a) for a, b in zip(hist, hist[1:])), hist
    assert hist[-1] > hist[0]
    assert np.all(tr.vars > 0) and np.all(np.isfinite(tr.means))
    np.testing.assert_allclose(tr.weights[seen].sum(axis=1), 1.0, rtol=1e-9)
    tr2 = BaumWelchTrainer(means, vars_, w, trans, data, labels, var_floor=1e-3, update_transitions=False)
    tr2.iteration()
    ok = stats[seen][:, :, 0] > 0
    np.testing.assert_allclose(tr2.means[seen][ok], mu[ok], rtol=1e-12)
    np.testing.assert_allclose(tr2.vars[seen][ok], np.maximum(sigma, 1e-3)[ok], rtol=1e-12)
    tr.close()
    tr2.close()


def test_reference_pickle_scores_identically(R):
    """N2: a model pickled by the reference evaluates to the reference's own cost on the GPU."""
    g = load_golden("G12_reference_pickle")
    hmms = pickle.loads(g["pickle"].tobytes())
    np.testing.assert_allclose([h.evaluate(g["x"]) for h in hmms], g["evaluate"], rtol=1e-10)
    # the packed .npz wire format carries the same models
    import tempfile
    from sr.recognition.model_io import save_models_npz, load_models_npz
    with tempfile.TemporaryDirectory() as td:
        save_models_npz(os.path.join(td, "v.npz"), hmms)
        back = load_models_npz(os.path.join(td, "v.npz"))
    np.testing.assert_allclose([h.evaluate(g["x"]) for h in back], g["evaluate"], rtol=1e-10)


# ------------------------------------------------------------------ N3: front-end
def test_feature_stack_matches_reference(R):
    import sr
    from sr.core import delta_feature, stack_features, stack_features_batch
    from sr.feature import standardize
    g = load_golden("G13_feature_stack")
    n = int(g["n"])
    for i in range(n):
        ceps = g["ceps%d" % i]
        np.testing.assert_allclose(delta_feature(ceps), g["delta%d" % i], rtol=1e-14, atol=1e-14)
        np.testing.assert_allclose(stack_features(ceps), g["feats%d" % i], rtol=1e-11, atol=1e-12)
    assert sr.delta_feature is delta_feature
    # standardize subtracts the mean from the caller's array in place, like the reference
    raw = np.concatenate([g["ceps2"], g["delta2"], g["ddelta2"]], axis=1)
    arg = raw.copy()
    out = standardize(arg)
    np.testing.assert_allclose(out, g["feats2"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(arg, raw - raw.mean(axis=0), rtol=1e-13, atol=1e-13)
    with pytest.raises(IndexError):
        delta_feature(g["ceps0"][:1])
    # batched, resident: front-end -> likelihood without a host round trip (fp64 and fp32 storage)
    for dt, tol in ((np.float64, 1e-11), (np.float32, 2e-6)):
        b = stack_features_batch([g["ceps%d" % i] for i in range(1, n)], dtype=dt)
        for i, f in zip(range(1, n), b.features()):
            np.testing.assert_allclose(f, g["feats%d" % i], rtol=tol, atol=tol)
        b.close()


def test_mfcc_features_matches_reference(R, tmp_path):
    """N3: sr.feature.mfcc_features (wav path in, like the reference) and the batched / resident entry points."""
    from scipy.io import wavfile
    from sr.feature import mfcc_features, mfcc_from_signals, features_from_signals
    g = load_golden("G15_mfcc")
    n = int(g["n"])
    for i in (4, 5, 7, 8):
        path = str(tmp_path / ("c%d.wav" % i))
        wavfile.write(path, int(g["rate%d" % i]), g["signal%d" % i])
        fb, mf = mfcc_features(path)
        np.testing.assert_allclose(fb, g["fbank%d" % i], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(mf, g["mfcc%d" % i], rtol=1e-9, atol=1e-9)
    idx = [i for i in range(n) if int(g["rate%d" % i]) == 16000]
    fbs, mfs = mfcc_from_signals([g["signal%d" % i] for i in idx], 16000)
    for i, fb, mf in zip(idx, fbs, mfs):
        np.testing.assert_allclose(fb, g["fbank%d" % i], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(mf, g["mfcc%d" % i], rtol=1e-9, atol=1e-9)
    # audio -> resident 39-dim batch == reference MFCC -> delta -> delta-delta -> standardize
    from sr.core import stack_features
    long = [i for i in idx if len(g["mfcc%d" % i]) >= 3 and np.all(np.std(g["mfcc%d" % i], axis=0) > 0)]
    b = features_from_signals([g["signal%d" % i] for i in long], 16000)
    for i, f in zip(long, b.features()):
        np.testing.assert_allclose(f, stack_features(g["mfcc%d" % i]), rtol=1e-7, atol=1e-7)
    b.close()
    with pytest.raises(IndexError):
        mfcc_from_signals([np.zeros(0, dtype=np.int16)])


def test_in_flight_lanes_give_the_same_results(R):
    """batch.InFlight: two contexts / host threads on one GPU decode a stream of batches; results equal the single-lane
    ones, in input order."""
    from sr.recognition.batch import IsolatedWordRecognizer, InFlight
    g = load_golden("G3_isolated_decode_c2")
    W = g["means"].shape[0]
    hmms = [make_hmm(R, g["means"][i], g["vars"][i], g["w"][i], g["trans"]) for i in range(W)]
    xs = [g["x%d" % u] for u in range(len(g["words"]))]
    batches = [xs, xs[::-1], xs[:1], xs[1:], xs]
    single = IsolatedWordRecognizer(hmms)
    ref = [single.recognize(b) for b in batches]
    pool = InFlight(lambda ctx: IsolatedWordRecognizer(hmms, ctx=ctx), n_lanes=2)
    got = pool.map(lambda rec, b: rec.recognize(b), batches)
    for (w0, c0), (w1, c1) in zip(ref, got):
        np.testing.assert_array_equal(w0, w1)
        np.testing.assert_array_equal(c0, c1)
    with pytest.raises(ZeroDivisionError):
        pool.map(lambda rec, b: 1 // 0, batches)
    pool.close()
