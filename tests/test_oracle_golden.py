# -*- coding: utf-8 -*-
"""Pin the CPU oracle (oracle/ref_numpy.py) against outputs captured from the
reference itself (tests/golden/*.npz, produced by tools/make_goldens.py).
fp64 values: rtol 1e-12 (summation order differs from the reference's BLAS
dots); paths, cluster ids, iteration counts and decodes: exact."""
import io
import contextlib
import warnings

import numpy as np
import pytest

from oracle import ref_numpy as O
from conftest import load_golden

RT = 1e-12


def close(a, b, rtol=RT, atol=0.0):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


# ------------------------------------------------------------------ A1 / A3 / A2
@pytest.mark.parametrize("tag", ["m1d13", "m8d39"])
def test_G1_gmm_evaluate(tag):
    g = load_golden("G1_gmm_evaluate_" + tag)
    means, vars_, w, X = g["means"], g["vars"], g["w"], g["X"]
    S = means.shape[0]
    nll = np.array([[O.gmm_evaluate(x, means[s], vars_[s], w[s]) for s in range(S)] for x in X[:16]])
    close(nll, g["nll"][:16])
    comp = np.array([[O.gmm_evaluate(x, means[s], vars_[s], w[s], neg_log=False) for s in range(S)] for x in X[:8]])
    close(comp, g["comp"][:8], rtol=1e-11)
    # reference-shaped dense inverse (cost structure of hmm_state.py:17,42)
    dinv = np.array([[np.linalg.inv(np.diag(v)) for v in vars_[s]] for s in range(3)])
    nll_d = np.array([[O.gmm_evaluate(x, means[s], vars_[s], w[s], dense_inv=dinv[s]) for s in range(3)] for x in X[:8]])
    close(nll_d, g["nll"][:8, :3])
    # log-domain batch algebra (what the HIP kernels compute)
    close(O.gmm_neg_loglik_batch(X, means, vars_, w), g["nll"], rtol=1e-12)


def test_G2_mahalanobis():
    g = load_golden("G2_mahalanobis")
    out = np.array([O.mahalanobis(a, b, c) for a, b, c in zip(g["v1"], g["v2"], g["var"])])
    close(out, g["out"], rtol=1e-14)


# ---------------------------------------------------------------------------- A6
def _word_states(means, vars_, w):
    return [(means[s], vars_[s], w[s]) for s in range(means.shape[0])]


@pytest.mark.parametrize("tag", ["c1", "c2"])
def test_G3_isolated_decode(tag):
    g = load_golden("G3_isolated_decode_" + tag)
    means, vars_, w, trans = g["means"], g["vars"], g["w"], g["trans"]
    W = means.shape[0]
    U = len(g["words"])
    nes = np.zeros(means.shape[1], dtype=bool)
    for u in range(U):
        x = g["x%d" % u]
        ev = []
        for i in range(W):
            E = O.emission_matrix(x, _word_states(means[i], vars_[i], w[i]))
            costs, path = O.decode_states(E, nes, trans)
            close(costs, g["costs_%d_%d" % (u, i)])
            np.testing.assert_array_equal(path, g["path_%d_%d" % (u, i)])
            ev.append(costs[-1, -1])
        close(ev, g["evaluate_%d" % u])
        assert int(np.argmin(ev)) == int(g["words"][u])


def _dense(R, to, frm, cost):
    t = np.full((R, R), np.inf)
    t[to, frm] = cost
    return t


@pytest.mark.parametrize("K", [1, 2, 3, 7])
def test_G4_lattice_decode(K):
    g = load_golden("G4_lattice_decode")
    means, vars_, w, wt = g["means"], g["vars"], g["w"], g["word_trans"]
    W, n = means.shape[:2]
    rw, rs, nes, trans, ends = O.build_state_sequences(n, [wt] * W, [list(range(W))] * K)
    p = "K%d_" % K
    assert len(rw) == int(g[p + "R"])
    np.testing.assert_array_equal(rw, g[p + "row_word"])
    np.testing.assert_array_equal(rs, g[p + "row_state"])
    np.testing.assert_array_equal(ends, g[p + "ends"])
    ref_trans = _dense(len(rw), g[p + "arc_to"], g[p + "arc_from"], g[p + "arc_cost"])
    np.testing.assert_array_equal(trans, ref_trans)
    shared = {}
    states = [None if nes[r] else shared.setdefault((rw[r], rs[r]), (means[rw[r], rs[r]], vars_[rw[r], rs[r]], w[rw[r], rs[r]]))
              for r in range(len(rw))]
    x = g[p + "x"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        costs, path = O.decode_states(O.emission_matrix(x, states), nes, trans,
                                      end_points=[[e, -1] for e in ends])
    close(costs, g[p + "costs"])
    np.testing.assert_array_equal(path, g[p + "path"])
    assert O.path_to_words(path, nes, rw) == list(g[p + "digits"])


def test_G4_forced_alignment_lattice():
    g = load_golden("G4_lattice_decode")
    means, vars_, w, wt = g["means"], g["vars"], g["w"], g["word_trans"]
    W, n = means.shape[:2]
    labels = list(g["forced_labels"])
    rw, rs, nes, trans, ends = O.build_state_sequences(n, [wt] * W, [[l] for l in labels])
    np.testing.assert_array_equal(
        trans, _dense(len(rw), g["forced_arc_to"], g["forced_arc_from"], g["forced_arc_cost"]))
    np.testing.assert_array_equal(ends, g["forced_ends"])
    states = [None if nes[r] else (means[rw[r], rs[r]], vars_[rw[r], rs[r]], w[rw[r], rs[r]]) for r in range(len(rw))]
    costs, path = O.decode_states(O.emission_matrix(g["forced_x"], states), nes, trans,
                                  end_points=[[e, -1] for e in ends])
    close(costs, g["forced_costs"])
    np.testing.assert_array_equal(path, g["forced_path"])
    assert O.path_to_words(path, nes, rw) == list(g["forced_digits"]) == labels


@pytest.mark.parametrize("pen", [0, 1])
def test_G14_loop_grammar(pen):
    """N4: the word-loop graph (expressed by row order) decoded by the REFERENCE: the oracle reproduces its
    costs / path / digits, and the loop cost equals the minimum of the reference's K-layer costs."""
    g = load_golden("G14_loop_grammar")
    means, vars_, w, wt = g["means"], g["vars"], g["w"], g["word_trans"]
    W, n = means.shape[:2]
    pp = "p%d_" % pen
    nes, rw, rs, trans, ends = O.loop_grammar([wt] * W, n, float(g[pp + "penalty"]))
    np.testing.assert_array_equal(rw, g[pp + "row_word"])
    np.testing.assert_array_equal(rs, g[pp + "row_state"])
    np.testing.assert_array_equal(ends, g[pp + "ends"])
    np.testing.assert_array_equal(trans, _dense(len(rw), g[pp + "arc_to"], g[pp + "arc_from"], g[pp + "arc_cost"]))
    states = [None if nes[r] else (means[rw[r], rs[r]], vars_[rw[r], rs[r]], w[rw[r], rs[r]]) for r in range(len(rw))]
    for u in range(int(g["n_utts"])):
        x = g[pp + "x%d" % u]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            costs, path = O.decode_states(O.emission_matrix(x, states), nes, trans, end_points=[[e, -1] for e in ends])
        close(costs, g[pp + "costs%d" % u])
        np.testing.assert_array_equal(path, g[pp + "path%d" % u])
        assert O.path_to_words(path, nes, rw) == list(g[pp + "digits%d" % u])
        if pen == 0:
            ref = g[pp + "costs%d" % u]
            assert min(ref[e, -1] for e in ends) == np.min(g["layer_costs%d" % u])          # reference vs reference
            # the oracle's own K-layer lattices agree too
            kc = []
            for K in range(1, int(g["Kmax"]) + 1):
                rwk, rsk, nk, tk, ek = O.build_state_sequences(n, [wt] * W, [list(range(W))] * K)
                stk = [None if nk[r] else (means[rwk[r], rsk[r]], vars_[rwk[r], rsk[r]], w[rwk[r], rsk[r]])
                       for r in range(len(rwk))]
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    ck, _ = O.decode_states(O.emission_matrix(x, stk), nk, tk, end_points=[[e, -1] for e in ek])
                kc.append(min(ck[e, -1] for e in ek))
            close(kc, g["layer_costs%d" % u])
            assert min(costs[e, -1] for e in ends) == min(kc)


def test_G6_decode_edges():
    g = load_golden("G6_decode_edges")
    means, vars_, w, trans = g["means"], g["vars"], g["w"], g["trans"]
    st = _word_states(means, vars_, w)
    nes = np.zeros(5, dtype=bool)
    # T = 1: c-1 wraps onto column 0 itself, path is empty
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        c1, p1 = O.decode_states(O.emission_matrix(g["t1_x"], st), nes, trans)
    close(c1, g["t1_costs"])
    assert tuple(p1.shape) == tuple(g["t1_path_shape"])
    # T = 2: end row unreachable -> warning, inf, back-pointers of all-inf cells still followed
    with pytest.warns(UserWarning):
        c2, p2 = O.decode_states(O.emission_matrix(g["t2_x"], st), nes, trans)
    np.testing.assert_array_equal(np.isinf(c2), np.isinf(g["t2_costs"]))
    close(c2[~np.isinf(c2)], g["t2_costs"][~np.isinf(c2)])
    np.testing.assert_array_equal(p2, g["t2_path"])
    # equal end costs: the LAST listed end point wins
    st5 = st[:4] + [st[3]]
    E = O.emission_matrix(g["tie_x"], st5)
    c, p = O.decode_states(E, nes, g["tie_trans"], end_points=[[3, -1], [4, -1]])
    close(c, g["tie_costs"])
    np.testing.assert_array_equal(p, g["tie_path"])
    c, p = O.decode_states(E, nes, g["tie_trans"], end_points=[[4, -1], [3, -1]])
    np.testing.assert_array_equal(p, g["tie_path_rev"])
    assert not np.array_equal(g["tie_path"], g["tie_path_rev"])
    # equal-cost predecessors: lowest origin wins
    st3 = [st[0], st[1], st[1]]
    c, p = O.decode_states(O.emission_matrix(g["tie_x"], st3), np.zeros(3, dtype=bool), g["ptie_trans"])
    close(c, g["ptie_costs"])
    np.testing.assert_array_equal(p, g["ptie_path"])


# ---------------------------------------------------------------------------- A5
def test_G5_dtw():
    g = load_golden("G5_dtw")
    x, y, var, trans = g["x"], g["y"], g["var"], g["trans"]
    Ee = O.distance_matrix(x, y, "euclid")
    Em = O.distance_matrix(x, y, "mahalanobis", var)
    for tag, E, tr, beam in (("euclid", Ee, trans, np.inf), ("mahal", Em, trans, np.inf),
                             ("beam3", Em, trans, 3), ("beam2", Ee, trans, 2),
                             ("skip", Em, g["trans_skip"], np.inf)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            c, p = O.dtw(E, tr, beam=beam)
        ref = g["costs_" + tag]
        np.testing.assert_array_equal(np.isinf(c), np.isinf(ref))
        fin = ~np.isinf(ref)
        close(c[fin], ref[fin])
        np.testing.assert_array_equal(p, g["path_" + tag])
    close(O.calc_transition_costs(2, g["seg_lens_skip"]), g["trans_skip"], rtol=0)


def test_lattice_beam_reproduces_the_reference_dtw_beam_on_chains():
    """N4: decode_hmm_states has no pruning in the reference; the oracle's lattice beam is dtw's rank beam
    (decode.py:62-68) carried over.  On a left-to-right chain, where dtw and decode_hmm_states are the same DP, it
    reproduces the reference's own beam results (G5): the same cells survive, same costs, same path."""
    g = load_golden("G5_dtw")
    x, y, var, trans = g["x"], g["y"], g["var"], g["trans"]
    n = len(y)
    for tag, E, beam in (("beam3", O.distance_matrix(x, y, "mahalanobis", var), 3), ("beam2", O.distance_matrix(x, y, "euclid"), 2)):
        ref = g["costs_" + tag]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            c, p = O.decode_states(E, np.zeros(n, dtype=bool), trans, beam=beam)
            c_inf, p_inf = O.decode_states(E, np.zeros(n, dtype=bool), trans, beam=np.inf)
            c_big, p_big = O.decode_states(E, np.zeros(n, dtype=bool), trans, beam=n)
        alive = np.isfinite(ref) & (ref != -1)                    # (dtw leaves -1 marks in its last column only)
        np.testing.assert_array_equal(np.isfinite(c[:, :-1]), alive[:, :-1])
        close(c[:, :-1][alive[:, :-1]], ref[:, :-1][alive[:, :-1]])
        np.testing.assert_array_equal(p, g["path_" + tag])
        np.testing.assert_array_equal(c_big, c_inf)              # a beam as wide as the column prunes nothing
        np.testing.assert_array_equal(p_big, p_inf)
        assert c[-1, -1] >= c_inf[-1, -1]                        # pruning can only cost


# ---------------------------------------------------------------------------- A7
@pytest.mark.parametrize("k", [2, 3])
@pytest.mark.parametrize("tag,iters", [("it1", 1), ("conv", 10000)])
def test_G7_gmm_em(k, tag, iters):
    g = load_golden("G7_gmm_em")
    M = len(g["init_w"])
    st = O.new_gmm_state(g["mu0"], g["var0"], M)
    st["means"][:] = g["init_means"]
    st["vars"][:] = g["init_vars"]
    st["w"][:] = g["init_w"]
    n = O.gmm_em(g["data"], st["means"], st["vars"], st["w"], k, max_iteration=iters,
                 old=(st["mu_old"], st["sigma_old"], st["w_old"]))
    p = "k%d_%s_" % (k, tag)
    assert n == int(g[p + "iters"])
    close(st["means"], g[p + "means"], rtol=1e-9)
    close(st["vars"], g[p + "vars"], rtol=1e-9)
    close(st["w"], g[p + "w"], rtol=1e-9)
    close(st["mu_old"], g[p + "mu_old"], rtol=1e-9)
    close(st["w_old"], g[p + "w_old"], rtol=1e-9)


@pytest.mark.parametrize("k", [2, 3])
def test_G16_gmm_em_linear_domain_underflow(k):
    """Frames whose weighted densities round to 0 in the reference's linear domain get an all-zero responsibility
    row (hmm_state.py:130-133) / no share of an underflowed component."""
    g = load_golden("G16_gmm_em_underflow")
    p0 = np.array([O.gmm_evaluate(x, g["init_means"], g["init_vars"], g["init_w"], neg_log=False)[:k] for x in g["data"]])
    np.testing.assert_array_equal(p0 == 0, g["k%d_p0" % k] == 0)
    assert (p0.sum(axis=1) == 0).sum() == 3
    for tag, iters in (("it1", 1), ("it3", 3)):
        M = len(g["init_w"])
        st = O.new_gmm_state(g["mu0"], g["var0"], M)
        st["means"][:], st["vars"][:], st["w"][:] = g["init_means"], g["init_vars"], g["init_w"]
        O.gmm_em(g["data"], st["means"], st["vars"], st["w"], k, max_iteration=iters,
                 old=(st["mu_old"], st["sigma_old"], st["w_old"]))
        close(st["means"], g["k%d_%s_means" % (k, tag)], rtol=1e-9)
        close(st["vars"], g["k%d_%s_vars" % (k, tag)], rtol=1e-9)
        close(st["w"], g["k%d_%s_w" % (k, tag)], rtol=1e-9)


# --------------------------------------------------------------------------- A14
@pytest.mark.parametrize("tag,k,dist", [("k2m", 2, "mahalanobis"), ("k4m", 4, "mahalanobis"), ("k4e", 4, "euclid")])
def test_G8_kmeans(tag, k, dist):
    g = load_golden("G8_kmeans")
    np.random.seed(0)
    cl, ce, cov = O.kmeans(g["data"], k, g[tag + "_c0"].copy(), dist=dist)
    np.testing.assert_array_equal(cl, g[tag + "_clusters"])
    close(ce, g[tag + "_centroids"], rtol=0)
    close(cov, g[tag + "_cov"], rtol=0)


# ---------------------------------------------------------------------- A15 / A9
def _ys(g):
    return [g["y%d" % i] for i in range(int(g["n"]))]


def test_G9_hmm_fit_single_gaussian():
    g = load_golden("G9_hmm_fit_single")
    m = O.hmm_fit(_ys(g), 5, 1, use_gmm=False)
    close(m["mu"], g["mu"])
    close(m["sigma"], g["sigma"])
    close(m["transitions"], g["transitions"])
    assert [len(s) for s in m["segments"]] == list(g["seg_lens"])
    for i, s in enumerate(m["segments"]):
        close(s, g["seg%d" % i], rtol=0)
    np.testing.assert_array_equal(O.get_segments_from_path(g["gsp_path"], 5), g["gsp_out"])
    close(O.calc_transition_costs(2, g["ctc_lens"]), g["ctc_out"], rtol=0)


@pytest.mark.parametrize("ng,em", [(4, True), (8, True), (4, False)])
def test_G10_hmm_fit_gmm(ng, em):
    g = load_golden("G10_hmm_fit_gmm")
    tag = "g%d_%s" % (ng, "em" if em else "km")
    ys = _ys(g)
    np.random.seed(5)
    m = O.hmm_fit(ys, 5, ng, use_gmm=True, use_em=em)
    close(m["mu"], g[tag + "_mu"])
    close(m["transitions"], g[tag + "_transitions"])
    close(np.array([s["means"] for s in m["gmm"]]), g[tag + "_means"], rtol=1e-8)
    close(np.array([s["vars"] for s in m["gmm"]]), g[tag + "_vars"], rtol=1e-8)
    close(np.array([s["w"] for s in m["gmm"]]), g[tag + "_w"], rtol=1e-8)
    assert [len(s) for s in m["segments"]] == list(g[tag + "_seg_lens"])
    close([O.hmm_evaluate(y, m) for y in ys[:2]], g[tag + "_evaluate"][:2], rtol=1e-8)
    # int(ln(n)) splits: a "4-Gaussian" model trains 2 components, the rest stay at (mu, sigma, 1/4)
    if ng == 4:
        for s in range(5):
            close(m["gmm"][s]["means"][2:], np.tile(m["mu"][s], (2, 1)), rtol=0)
            close(m["gmm"][s]["w"][2:], [0.25, 0.25], rtol=0)


# --------------------------------------------------------------------------- A11
@pytest.mark.parametrize("iters", [1, 3])
def test_G11_continuous_train(iters):
    g = load_golden("G11_continuous_train")
    W = int(g["n_words"])
    U = int(g["n_utts"])
    data = [g["x%d" % i] for i in range(U)]
    labels = [list(g["labels%d" % i]) for i in range(U)]
    models = []
    for wi in range(W):
        n = g["init%d_means" % wi].shape[0]
        models.append(dict(transitions=g["init%d_transitions" % wi].copy(), gmm=[
            dict(means=g["init%d_means" % wi][s].copy(), vars=g["init%d_vars" % wi][s].copy(),
                 w=g["init%d_w" % wi][s].copy(), mu_old=g["init%d_mu_old" % wi][s].copy(),
                 sigma_old=g["init%d_sigma_old" % wi][s].copy(), w_old=g["init%d_w_old" % wi][s].copy())
            for s in range(n)]))
    np.random.seed(9)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out, n_it, _ = O.continuous_train(data, models, labels, n_gaussians=4, n_segments=5, max_iteration=iters)
    for wi in range(W):
        p = "it%d_%d_" % (iters, wi)
        close(np.array([s["means"] for s in out[wi]["gmm"]]), g[p + "means"], rtol=1e-7)
        close(np.array([s["vars"] for s in out[wi]["gmm"]]), g[p + "vars"], rtol=1e-7)
        close(np.array([s["w"] for s in out[wi]["gmm"]]), g[p + "w"], rtol=1e-7)
        close(out[wi]["transitions"], g[p + "transitions"], rtol=1e-9)


@pytest.mark.parametrize("iters", [1, 2])
def test_G17_continuous_train_8mix(iters):
    """A second capture of the reference's continuous_train: 13-dim, 8 mixtures (three binary splits), 4 words of 3
    states, label strings of 2-4 words with repeats."""
    g = load_golden("G17_continuous_train_8mix")
    W, U, ng, nseg = int(g["n_words"]), int(g["n_utts"]), int(g["n_gaussians"]), int(g["n_segments"])
    data = [g["x%d" % i] for i in range(U)]
    labels = [list(g["labels%d" % i]) for i in range(U)]
    models = []
    for wi in range(W):
        n = g["init%d_means" % wi].shape[0]
        models.append(dict(transitions=g["init%d_transitions" % wi].copy(), gmm=[
            dict(means=g["init%d_means" % wi][s].copy(), vars=g["init%d_vars" % wi][s].copy(),
                 w=g["init%d_w" % wi][s].copy(), mu_old=g["init%d_mu_old" % wi][s].copy(),
                 sigma_old=g["init%d_sigma_old" % wi][s].copy(), w_old=g["init%d_w_old" % wi][s].copy())
            for s in range(n)]))
    np.random.seed(19)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out, n_it, _ = O.continuous_train(data, models, labels, n_gaussians=ng, n_segments=nseg, max_iteration=iters)
    for wi in range(W):
        p = "it%d_%d_" % (iters, wi)
        close(np.array([s["means"] for s in out[wi]["gmm"]]), g[p + "means"], rtol=1e-7)
        close(np.array([s["vars"] for s in out[wi]["gmm"]]), g[p + "vars"], rtol=1e-7)
        close(np.array([s["w"] for s in out[wi]["gmm"]]), g[p + "w"], rtol=1e-7)
        close(out[wi]["transitions"], g[p + "transitions"], rtol=1e-9)


# --------------------------------------------------------------------------- A13
def _brute_force(E, is_nes, trans, end_rows):
    """Enumerate every path of the A6 lattice semantics explicitly."""
    R, T = E.shape
    total = {}  # (r,c) -> sum of exp(-cost) of complete paths through the cell
    all_w = []

    def extend(path, cost):
        r, c = path[-1]
        if c == T - 1 and r in end_rows:
            all_w.append((list(path), cost))
        for s in range(R):
            if np.isinf(trans[s, r]):
                continue
            same = is_nes[s] or is_nes[r]
            if same:
                if s <= r:
                    continue
                extend(path + [(s, c)], cost + trans[s, r] + E[s, c])
            elif c + 1 < T:
                extend(path + [(s, c + 1)], cost + trans[s, r] + E[s, c + 1])

    extend([(0, 0)], E[0, 0])
    Z = sum(np.exp(-c) for _, c in all_w)
    for p, c in all_w:
        for cell in set(p):
            total[cell] = total.get(cell, 0.0) + np.exp(-c)
    gamma = np.zeros((R, T))
    for (r, c), v in total.items():
        gamma[r, c] = v / Z
    return np.log(Z), gamma, min(c for _, c in all_w)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_A13_forward_backward_bruteforce(seed):
    rng = np.random.default_rng(seed)
    n, K, T = 2, 2, 5
    wt = np.full((n, n), np.inf)
    wt[0, 0], wt[1, 0], wt[1, 1] = rng.uniform(0.1, 2, 3)
    rw, rs, nes, trans, ends = O.build_state_sequences(n, [wt, wt * 1.3], [[0, 1]] * K)
    E = rng.uniform(0.5, 4.0, size=(len(rw), T))
    E[nes] = 0.0
    la, lb, gamma, logp = O.forward_backward(E, nes, trans, ends)
    bl, bg, bmin = _brute_force(E, nes, trans, set(ends))
    close(logp, bl, rtol=1e-12)
    close(gamma, bg, rtol=1e-10, atol=1e-14)
    # log P from beta at the start cell equals log P from alpha
    close(lb[0, 0] - E[0, 0], logp, rtol=1e-12)
    # Viterbi cost >= -log P, and equals the brute-force minimum
    costs, _ = O.decode_states(E, nes, trans, end_points=[[e, -1] for e in ends])
    vit = min(costs[e, -1] for e in ends)
    close(vit, bmin, rtol=1e-12)
    assert vit >= -logp - 1e-12


# ---------------------------------------------------------------------------- N3
def test_G13_feature_stack():
    g = load_golden("G13_feature_stack")
    for i in range(int(g["n"])):
        ceps = g["ceps%d" % i]
        close(O.delta_feature(ceps), g["delta%d" % i], rtol=0)
        close(O.delta_feature(g["delta%d" % i]), g["ddelta%d" % i], rtol=0)
        close(O.stack_features(ceps), g["feats%d" % i], rtol=1e-12, atol=1e-13)
    with pytest.raises(IndexError):
        O.delta_feature(g["ceps0"][:1])


def test_G15_mfcc():
    """N3: mfcc_features of the reference on int16 wav files (edge lengths, 8 kHz, silence): bit-exact restatement."""
    g = load_golden("G15_mfcc")
    for i in range(int(g["n"])):
        fb, mf = O.mfcc_features_signal(g["signal%d" % i], int(g["rate%d" % i]))
        np.testing.assert_array_equal(fb, g["fbank%d" % i])
        np.testing.assert_array_equal(mf, g["mfcc%d" % i])
