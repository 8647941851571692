# -*- coding: utf-8 -*-
"""GPU parity tests of the HIP kernels, called through the C ABI (ctypes),
against the CPU oracle and the golden vectors captured from the reference.

Tolerances: likelihoods fp64 1e-10 rel (north star: 1e-5), fp32 1e-3 rel (north
star: 1e-3); DP costs fp64 1e-10 rel; paths / end choices / decodes bit-exact.
"""
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


def arcs_of(trans):
    to, frm = np.nonzero(~np.isinf(trans))
    return to, frm, trans[to, frm]


def graph(row_state, trans, start_rows, end_rows):
    to, frm, cost = arcs_of(trans)
    return dict(row_state=row_state, arc_to=to, arc_from=frm, arc_cost=cost, start_rows=start_rows,
                end_rows=end_rows)


# ---------------------------------------------------------------------------- A3
@pytest.mark.parametrize("tag", ["m1d13", "m8d39"])
@pytest.mark.parametrize("dtype,rtol", [(np.float64, 1e-10), (np.float32, 1e-3)])
def test_loglik_golden(hip, ctx, tag, dtype, rtol):
    g = load_golden("G1_gmm_evaluate_" + tag)
    gmm = hip.PackedGMM(ctx, g["means"], g["vars"], g["w"])
    b = hip.Batch(ctx, feats=g["X"], offsets=[0, len(g["X"])], dtype=dtype)
    nll = b.loglik(gmm)
    np.testing.assert_allclose(nll, g["nll"], rtol=rtol)
    if dtype == np.float64:
        comp = gmm.component_loglik(7, g["X"])
        np.testing.assert_allclose(np.exp(comp), g["comp"][:, 7], rtol=1e-9)


@pytest.mark.parametrize("S,M,D,N", [(3, 1, 6, 130), (50, 8, 39, 1000), (7, 5, 13, 257), (4, 3, 50, 64), (2, 9, 24, 63),
                                     (5, 32, 39, 100), (3, 20, 13, 70), (70, 2, 13, 300), (130, 4, 39, 65), (9, 48, 7, 33),
                                     # round 4: every D <= 64 on the matrix cores (operands zero-padded to the next
                                     # instantiated length: 10 -> 16, 18 -> 24, 30 -> 40, 45 -> 48, 64), D = 70 on the vector kernel
                                     (6, 8, 10, 200), (5, 4, 18, 97), (11, 8, 30, 130), (4, 8, 45, 96), (3, 16, 64, 70), (3, 2, 70, 40)])
def test_loglik_vs_oracle_shapes(hip, ctx, S, M, D, N):
    rng = np.random.default_rng(S * 100 + M)
    means = rng.normal(size=(S, M, D))
    vars_ = rng.uniform(0.5, 1.5, size=(S, M, D))
    w = rng.dirichlet(np.ones(M), size=S)
    X = rng.normal(size=(N, D)) * 1.3
    ref = O.gmm_neg_loglik_batch(X, means, vars_, w)
    gmm = hip.PackedGMM(ctx, means, vars_, w)
    for dtype, rtol in ((np.float64, 1e-11), (np.float32, 1e-3)):
        b = hip.Batch(ctx, feats=X, offsets=[0, 10, 10, N], dtype=dtype)
        np.testing.assert_allclose(b.loglik(gmm), ref, rtol=rtol)


def test_loglik_zero_weight_and_singular(hip, ctx):
    rng = np.random.default_rng(5)
    means, vars_ = rng.normal(size=(2, 3, 4)), rng.uniform(0.5, 1.5, size=(2, 3, 4))
    w = np.array([[0.5, 0.0, 0.5], [1.0, 0.0, 0.0]])
    X = rng.normal(size=(9, 4))
    gmm = hip.PackedGMM(ctx, means, vars_, w)
    b = hip.Batch(ctx, feats=X, offsets=[0, 9])
    with np.errstate(divide="ignore"):
        ref = O.gmm_neg_loglik_batch(X, means, vars_, w)
    np.testing.assert_allclose(b.loglik(gmm), ref, rtol=1e-11)
    vars_[1, 2, 3] = 0.0
    with pytest.raises(np.linalg.LinAlgError):
        hip.PackedGMM(ctx, means, vars_, w)


# ---------------------------------------------------------------------------- A6
@pytest.mark.parametrize("tag", ["c1", "c2"])
def test_viterbi_isolated_golden(hip, ctx, tag):
    """G3: each word model on its own (one graph per word, reference semantics:
    single start row 0, end row R-1) AND all words stacked into one graph."""
    g = load_golden("G3_isolated_decode_" + tag)
    means, vars_, w, trans = g["means"], g["vars"], g["w"], g["trans"]
    W, n, M, D = means.shape
    U = len(g["words"])
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, M, D), vars_.reshape(W * n, M, D), w.reshape(W * n, M))
    xs = [g["x%d" % u] for u in range(U)]
    b = hip.Batch(ctx, xs)
    b.loglik(gmm, fetch=False)
    graphs = [graph(np.arange(n) + i * n, trans, [0], [n - 1]) for i in range(W)]
    lat = hip.Lattices(ctx, graphs)
    for i in range(W):
        r = lat.viterbi(b, utt_lattice=np.full(U, i), want_costs=True)
        for u in range(U):
            ref = g["costs_%d_%d" % (u, i)]
            fin = ~np.isinf(ref)
            np.testing.assert_array_equal(np.isinf(r["costs"][u]), ~fin)
            np.testing.assert_allclose(r["costs"][u][fin], ref[fin], rtol=1e-10)
            np.testing.assert_array_equal(r["paths"][u], g["path_%d_%d" % (u, i)])
    # stacked: W independent chains in one graph, one start and one end row per word
    big = np.full((W * n, W * n), np.inf)
    for i in range(W):
        big[i * n:(i + 1) * n, i * n:(i + 1) * n] = trans
    st = hip.Lattices(ctx, [graph(np.arange(W * n), big, [i * n for i in range(W)], [i * n + n - 1 for i in range(W)])])
    r = st.viterbi(b, want_path=False)
    for u in range(U):
        np.testing.assert_allclose(r["end_cost"][u], g["evaluate_%d" % u], rtol=1e-10)
        assert int(np.argmin(r["end_cost"][u])) == int(g["words"][u])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_viterbi_lattice_golden(hip, ctx, dtype):
    """G4: K-layer word lattices (same-column hops through non-emitting rows)."""
    g = load_golden("G4_lattice_decode")
    means, vars_, w, wt = g["means"], g["vars"], g["w"], g["word_trans"]
    W, n, M, D = means.shape
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, M, D), vars_.reshape(W * n, M, D), w.reshape(W * n, M))
    graphs, xs, Ks = [], [], (1, 2, 3, 7)
    for K in Ks:
        p = "K%d_" % K
        rw, rs = g[p + "row_word"], g[p + "row_state"]
        graphs.append(dict(row_state=np.where(rw < 0, -1, rw * n + rs), arc_to=g[p + "arc_to"],
                           arc_from=g[p + "arc_from"], arc_cost=g[p + "arc_cost"], start_rows=[0],
                           end_rows=g[p + "ends"]))
        xs.append(g[p + "x"])
    lat = hip.Lattices(ctx, graphs)
    b = hip.Batch(ctx, xs, dtype=dtype)
    b.loglik(gmm, fetch=False)
    r = lat.viterbi(b, utt_lattice=np.arange(len(Ks)), want_costs=True)
    for u, K in enumerate(Ks):
        p = "K%d_" % K
        ref = g[p + "costs"]
        fin = ~np.isinf(ref)
        np.testing.assert_array_equal(np.isinf(r["costs"][u]), ~fin)
        np.testing.assert_allclose(r["costs"][u][fin], ref[fin], rtol=1e-10 if dtype == np.float64 else 1e-5)
        np.testing.assert_array_equal(r["paths"][u], g[p + "path"])
        rw = g[p + "row_word"]
        assert O.path_to_words(r["paths"][u], rw < 0, rw) == list(g[p + "digits"])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_viterbi_loop_grammar_golden(hip, ctx, dtype):
    """G14 (N4): word-loop graph decoded by the reference; costs, bit-exact paths, digits."""
    from sr.recognition.continuous_speech import packed_loop_lattice
    g = load_golden("G14_loop_grammar")
    means, vars_, w, wt = g["means"], g["vars"], g["w"], g["word_trans"]
    W, n, M, D = means.shape
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, M, D), vars_.reshape(W * n, M, D), w.reshape(W * n, M))
    U = int(g["n_utts"])
    graphs = [packed_loop_lattice([wt] * W, n, float(g["p%d_penalty" % pen]))[0] for pen in (0, 1)]
    for pen in (0, 1):
        rw, rs = g["p%d_row_word" % pen], g["p%d_row_state" % pen]
        np.testing.assert_array_equal(graphs[pen]["row_state"], np.where(rw < 0, -1, rw * n + rs))
    lat = hip.Lattices(ctx, graphs)
    b = hip.Batch(ctx, [g["p%d_x%d" % (pen, u)] for pen in (0, 1) for u in range(U)], dtype=dtype)
    b.loglik(gmm, fetch=False)
    r = lat.viterbi(b, utt_lattice=np.repeat([0, 1], U), want_costs=True)
    for pen in (0, 1):
        rw = g["p%d_row_word" % pen]
        for u in range(U):
            k = pen * U + u
            ref = g["p%d_costs%d" % (pen, u)]
            fin = ~np.isinf(ref)
            np.testing.assert_array_equal(np.isinf(r["costs"][k]), ~fin)
            np.testing.assert_allclose(r["costs"][k][fin], ref[fin], rtol=1e-10 if dtype == np.float64 else 1e-5)
            np.testing.assert_array_equal(r["paths"][k], g["p%d_path%d" % (pen, u)])
            assert O.path_to_words(r["paths"][k], rw < 0, rw) == list(g["p%d_digits%d" % (pen, u)])
            if pen == 0 and dtype == np.float64:
                np.testing.assert_allclose(np.min(r["end_cost"][k]), np.min(g["layer_costs%d" % u]), rtol=1e-12)


def test_viterbi_loop_equals_min_over_layers_at_scale(hip, ctx):
    """Size-independent property of N4: on every utterance the word-loop cost equals the minimum over K of the
    exactly-K-words lattice costs (same kernels, same likelihoods), and the decoded word string equals that of
    the best K."""
    from sr.recognition.continuous_speech import packed_loop_lattice, packed_lattice
    from sr.recognition.batch import path_to_words
    rng = np.random.default_rng(77)
    W, n, M, D, U, KMAX = 10, 5, 2, 13, 240, 6
    means = rng.normal(size=(W, n, M, D)) * 2.0
    vars_ = rng.uniform(0.5, 1.5, size=(W, n, M, D))
    w = rng.dirichlet(np.ones(M), size=(W, n))
    trans = np.full((n, n), np.inf)
    for i in range(n):
        trans[i, i] = -np.log(0.8) if i < n - 1 else 0.0
        if i < n - 1:
            trans[i + 1, i] = -np.log(0.2)
    xs, truth = [], []
    for u in range(U):
        words = rng.integers(0, W, size=rng.integers(1, 6))
        segs = []
        for wd in words:
            T = int(rng.integers(8, 14))
            st = np.minimum(np.arange(T) * n // T, n - 1)
            comp = rng.integers(0, M, size=T)
            segs.append(means[wd, st, comp] + np.sqrt(vars_[wd, st, comp]) * rng.normal(size=(T, D)))
        xs.append(np.concatenate(segs))
        truth.append([int(v) for v in words])
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, M, D), vars_.reshape(W * n, M, D), w.reshape(W * n, M))
    b = hip.Batch(ctx, xs)
    b.loglik(gmm, fetch=False)
    wt = [trans] * W
    loop_graph = packed_loop_lattice(wt, n)[0]
    graphs = [loop_graph] + [packed_lattice(wt, n, [list(range(W))] * K)[0] for K in range(1, KMAX + 1)]
    lat = hip.Lattices(ctx, graphs)
    res = [lat.viterbi(b, utt_lattice=np.full(U, k, dtype=np.int32)) for k in range(len(graphs))]
    loop_cost = np.array([np.min(e) for e in res[0]["end_cost"]])
    layer_cost = np.array([[np.min(e) for e in res[k]["end_cost"]] for k in range(1, KMAX + 1)])   # [K, U]
    np.testing.assert_array_equal(loop_cost, layer_cost.min(axis=0))
    best_k = layer_cost.argmin(axis=0)
    # A12 on the device: the label sequences of gh_viterbi_labels equal path_to_words of the returned paths
    for k in (0, 3, KMAX):
        rw = np.where(graphs[k]["row_state"] >= 0, graphs[k]["row_state"] // n, -1)
        full = [np.zeros(0, dtype=np.int32)] * len(graphs)
        full = [np.where(g_["row_state"] >= 0, g_["row_state"] // n, -1) for g_ in graphs]
        rl = lat.viterbi_labels(b, full, utt_lattice=np.full(U, k, dtype=np.int32))
        np.testing.assert_array_equal(rl["best_end"], res[k]["best_end"])
        np.testing.assert_array_equal(rl["end_cost_flat"], res[k]["end_cost_flat"])
        for u in range(U):
            assert [int(v) for v in rl["labels"][u]] == path_to_words(res[k]["paths"][u], graphs[k]["row_state"], n)
    n_right = 0
    for u in range(U):
        got = path_to_words(res[0]["paths"][u], loop_graph["row_state"], n)
        ref = path_to_words(res[1 + best_k[u]]["paths"][u], graphs[1 + best_k[u]]["row_state"], n)
        assert got == ref and len(got) == best_k[u] + 1
        n_right += got == truth[u]
    assert n_right >= 0.9 * U   # well-separated synthetic words: the loop decode recovers the spoken string


def test_viterbi_edges_golden(hip, ctx):
    g = load_golden("G6_decode_edges")
    means, vars_, w, trans = g["means"], g["vars"], g["w"], g["trans"]
    gmm = hip.PackedGMM(ctx, means, vars_, w)
    b = hip.Batch(ctx, [g["t1_x"], g["t2_x"], g["tie_x"], np.zeros((0, means.shape[2]))])
    b.loglik(gmm, fetch=False)
    lat = hip.Lattices(ctx, [
        graph(np.arange(5), trans, [0], [4]),
        graph([0, 1, 2, 3, 3], g["tie_trans"], [0], [3, 4]),
        graph([0, 1, 2, 3, 3], g["tie_trans"], [0], [4, 3]),
        graph([0, 1, 1], g["ptie_trans"], [0], [2]),
    ])
    r = lat.viterbi(b, utt_lattice=[0, 0, 1, 0], want_costs=True)
    # T = 1: every arc reads column 0 itself; empty path
    np.testing.assert_allclose(r["costs"][0], g["t1_costs"], rtol=1e-10)
    assert r["paths"][0].shape[0] == 0
    # T = 2: unreachable end, back-pointers of all-inf cells still followed
    ref = g["t2_costs"]
    np.testing.assert_array_equal(np.isinf(r["costs"][1]), np.isinf(ref))
    np.testing.assert_array_equal(r["paths"][1], g["t2_path"])
    assert np.isinf(r["end_cost"][1][0])
    # equal end costs: the last listed end wins
    np.testing.assert_array_equal(r["paths"][2], g["tie_path"])
    assert r["best_end"][2] == 1 and r["end_cost"][2][0] == r["end_cost"][2][1]
    assert r["paths"][3].shape[0] == 0 and r["best_end"][3] == -1  # empty utterance
    r2 = lat.viterbi(b, utt_lattice=[0, 0, 2, 3])
    np.testing.assert_array_equal(r2["paths"][2], g["tie_path_rev"])
    r3 = lat.viterbi(b, utt_lattice=[3, 3, 3, 3], want_costs=True)
    np.testing.assert_allclose(r3["costs"][2], g["ptie_costs"], rtol=1e-10)
    np.testing.assert_array_equal(r3["paths"][2], g["ptie_path"])


def test_viterbi_random_graphs_vs_oracle(hip, ctx):
    """Random sparse graphs with non-emitting rows at random positions (multi-level
    same-column chains, dead same-column back arcs) against the oracle."""
    rng = np.random.default_rng(77)
    S, M, D = 6, 2, 5
    means, vars_ = rng.normal(size=(S, M, D)), rng.uniform(0.5, 1.5, size=(S, M, D))
    w = rng.dirichlet(np.ones(M), size=S)
    gmm = hip.PackedGMM(ctx, means, vars_, w)
    graphs, xs, refs = [], [], []
    while len(graphs) < 24:
        R = int(rng.integers(3, 40))
        row_state = rng.integers(0, S, size=R)
        row_state[rng.random(R) < 0.3] = -1
        trans = np.full((R, R), np.inf)
        for r in range(R):
            for o in rng.integers(0, R, size=rng.integers(1, 4)):
                trans[r, o] = rng.uniform(0.1, 3.0)
        ends = list(rng.integers(0, R, size=rng.integers(1, 4)))
        T = int(rng.integers(2, 30))
        x = rng.normal(size=(T, D))
        nes = row_state < 0
        states = [None if nes[r] else (means[row_state[r]], vars_[row_state[r]], w[row_state[r]]) for r in range(R)]
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ref = O.decode_states(O.emission_matrix(x, states), nes, trans, end_points=[[e, -1] for e in ends])
        except (IndexError, NameError, RuntimeError):
            continue  # graphs on which the reference itself raises / never returns
        graphs.append(graph(row_state, trans, [0], ends))
        xs.append(x)
        refs.append(ref)
    lat = hip.Lattices(ctx, graphs)
    b = hip.Batch(ctx, xs)
    b.loglik(gmm, fetch=False)
    r = lat.viterbi(b, utt_lattice=np.arange(len(xs)), want_costs=True)
    for u, (costs, path) in enumerate(refs):
        fin = ~np.isinf(costs)
        np.testing.assert_array_equal(np.isinf(r["costs"][u]), ~fin)
        np.testing.assert_allclose(r["costs"][u][fin], costs[fin], rtol=1e-10)
        np.testing.assert_array_equal(r["paths"][u], path.reshape(-1, 2))


def test_viterbi_self_pointer_is_nameerror(hip, ctx):
    gmm = hip.PackedGMM(ctx, np.zeros((1, 1, 2)), np.ones((1, 1, 2)), np.ones((1, 1)))
    b = hip.Batch(ctx, [np.zeros((3, 2))])
    b.loglik(gmm, fetch=False)
    trans = np.full((2, 2), np.inf)
    trans[1, 1] = 0.0  # non-emitting self loop: first (only) candidate is the cell itself
    lat = hip.Lattices(ctx, [graph([0, -1], trans, [0], [1])])
    with pytest.raises(NameError):
        lat.viterbi(b)


# ---------------------------------------------------------------------------- A2
def test_distance_matrix(hip, ctx):
    g = load_golden("G2_mahalanobis")
    v1, v2, var = g["v1"], g["v2"], g["var"]
    out = hip.distance_matrix(ctx, v1, v2[:1], var[:1])
    ref = np.array([O.mahalanobis(a, v2[0], var[0]) for a in v1])
    np.testing.assert_allclose(out[0], ref, rtol=1e-13)
    out = hip.distance_matrix(ctx, v1, v2, var)
    np.testing.assert_allclose(np.diag(out), g["out"], rtol=1e-13)
    out = hip.distance_matrix(ctx, v1, v2)
    np.testing.assert_allclose(out[3, 5], np.linalg.norm(v1[5] - v2[3]), rtol=1e-14)


# --------------------------------------------------------------------------- A13
def test_forward_backward_vs_oracle(hip, ctx):
    """Sum-product twin of A6 -- not in the reference; the oracle it is compared with is pinned
    by brute-force path enumeration (tests/test_oracle_golden.py)."""
    g = load_golden("G4_lattice_decode")
    means, vars_, w, wt = g["means"], g["vars"], g["w"], g["word_trans"]
    W, n, M, D = means.shape
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, M, D), vars_.reshape(W * n, M, D), w.reshape(W * n, M))
    graphs, xs, refs = [], [], []
    for K in (1, 2, 3):
        p = "K%d_" % K
        rw, rs = g[p + "row_word"], g[p + "row_state"]
        row_state = np.where(rw < 0, -1, rw * n + rs)
        graphs.append(dict(row_state=row_state, arc_to=g[p + "arc_to"], arc_from=g[p + "arc_from"],
                           arc_cost=g[p + "arc_cost"], start_rows=[0], end_rows=g[p + "ends"]))
        x = g[p + "x"]
        xs.append(x)
        R = len(rw)
        trans = np.full((R, R), np.inf)
        trans[g[p + "arc_to"], g[p + "arc_from"]] = g[p + "arc_cost"]
        nll = O.gmm_neg_loglik_batch(x, means.reshape(W * n, M, D), vars_.reshape(W * n, M, D), w.reshape(W * n, M))
        E = np.where(row_state[:, None] >= 0, nll[:, np.maximum(row_state, 0)].T, 0.0)
        refs.append((O.forward_backward(E, rw < 0, trans, list(g[p + "ends"])), row_state, g[p + "costs"], g[p + "ends"]))
    lat = hip.Lattices(ctx, graphs)
    b = hip.Batch(ctx, xs)
    b.loglik(gmm, fetch=False)
    r = lat.forward_backward(b, utt_lattice=np.arange(len(xs)), want_matrices=True, want_occ=True)
    for u, ((la, lb, gamma, logp), row_state, vit_costs, ends) in enumerate(refs):
        np.testing.assert_allclose(r["logp"][u], logp, rtol=1e-10)
        for got, ref in ((r["alpha"][u], la), (r["beta"][u], lb)):
            fin = ~np.isinf(ref)
            np.testing.assert_array_equal(np.isinf(got), ~fin)
            np.testing.assert_allclose(got[fin], ref[fin], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(r["gamma"][u], gamma, rtol=1e-8, atol=1e-12)
        # per-frame state occupancies = gamma summed over the rows of each state
        occ = np.zeros((gamma.shape[1], W * n))
        for row, s in enumerate(row_state):
            if s >= 0:
                occ[:, s] += gamma[row]
        np.testing.assert_allclose(r["occ"][b.offsets[u]:b.offsets[u + 1]], occ, rtol=1e-8, atol=1e-12)
        # Viterbi cost is an upper bound of -log P
        assert min(vit_costs[e, -1] for e in ends) >= -logp - 1e-9


def test_forward_backward_isolated_properties(hip, ctx):
    """Left-to-right chains without non-emitting rows: sum_r gamma_t(r) = 1 for every frame,
    log P from alpha equals log P from beta, fp32 likelihood input agrees to 1e-5."""
    g = load_golden("G3_isolated_decode_c2")
    means, vars_, w, trans = g["means"], g["vars"], g["w"], g["trans"]
    W, n, M, D = means.shape
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, M, D), vars_.reshape(W * n, M, D), w.reshape(W * n, M))
    xs = [g["x0"], g["x1"]]
    lat = hip.Lattices(ctx, [graph(np.arange(n) + i * n, trans, [0], [n - 1]) for i in range(W)])
    out = {}
    for dtype in (np.float64, np.float32):
        b = hip.Batch(ctx, xs, dtype=dtype)
        nll = b.loglik(gmm)
        out[dtype] = (lat.forward_backward(b, utt_lattice=[0, 1], want_matrices=True), nll)
    r, nll = out[np.float64]
    for u in range(2):
        np.testing.assert_allclose(r["gamma"][u].sum(axis=0), 1.0, rtol=1e-9)
        e00 = nll[b.offsets[u], u * n]
        np.testing.assert_allclose(r["beta"][u][0, 0] - e00, r["logp"][u], rtol=1e-10)
        costs = g["costs_%d_%d" % (u, u)]
        assert costs[-1, -1] >= -r["logp"][u]
    np.testing.assert_allclose(out[np.float32][0]["logp"], r["logp"], rtol=1e-5)


def test_baum_welch_statistics_vs_numpy(hip, ctx):
    """gh_bw_accumulate: occupancy-weighted centred statistics of all states in one pass."""
    g = load_golden("G3_isolated_decode_c2")
    means, vars_, w, trans = g["means"], g["vars"], g["w"], g["trans"]
    W, n, M, D = means.shape
    S = W * n
    fm, fv, fw = means.reshape(S, M, D), vars_.reshape(S, M, D), w.reshape(S, M)
    gmm = hip.PackedGMM(ctx, fm, fv, fw)
    xs = [g["x0"], g["x1"]]
    words = [int(g["words"][0]), int(g["words"][1])]
    lat = hip.Lattices(ctx, [graph(np.arange(n) + i * n, trans, [0], [n - 1]) for i in range(W)])
    b = hip.Batch(ctx, xs)
    b.loglik(gmm, fetch=False)
    r = lat.forward_backward(b, utt_lattice=words, want_occ=True)
    stats = b.bw_accumulate(gmm)
    occ = r["occ"]
    X = np.concatenate(xs)
    np.testing.assert_allclose(occ.sum(axis=1), 1.0, rtol=1e-9)
    ref = np.zeros((S, M, 1 + 2 * D))
    for s in range(S):
        if occ[:, s].max() == 0:
            continue
        p = np.array([O.gmm_evaluate(x, fm[s], fv[s], fw[s], neg_log=False) for x in X])
        rr = occ[:, [s]] * p / p.sum(axis=1, keepdims=True)
        for m in range(M):
            d = X - fm[s, m]
            ref[s, m, 0] = rr[:, m].sum()
            ref[s, m, 1:1 + D] = (rr[:, [m]] * d).sum(axis=0)
            ref[s, m, 1 + D:] = (rr[:, [m]] * d * d).sum(axis=0)
    np.testing.assert_allclose(stats, ref, rtol=1e-8, atol=1e-12)
    assert np.all(stats[[s for s in range(S) if s // n not in words]] == 0)  # unvisited words untouched
    # M-step from the statistics: occupancy-weighted means move toward the data, variances stay positive
    from sr.recognition.parallel import m_step
    counts = occ.sum(axis=0)
    vis = [s for s in range(S) if s // n in words]
    mu, sigma, wn = m_step(stats[vis], counts[vis], fm[vis])
    assert np.all(sigma[stats[vis][:, :, 0] > 1e-3] > 0)
    np.testing.assert_allclose(wn.sum(axis=1), 1.0, rtol=1e-9)


def test_viterbi_chain_kernel_skip_arcs_and_stacking(hip, ctx):
    """Left-to-right graphs take the one-wave-per-chain-group kernel (DPP neighbour exchange, 1-byte
    back-pointers): chains with skip transitions (calc_transition_costs' jump over an empty segment),
    several chains stacked in one graph, > 64 rows in total (two lane groups), costs + paths + ends."""
    rng = np.random.default_rng(123)
    S, M, D = 12, 2, 4
    means, vars_ = rng.normal(size=(S, M, D)), rng.uniform(0.5, 1.5, size=(S, M, D))
    w = rng.dirichlet(np.ones(M), size=S)
    gmm = hip.PackedGMM(ctx, means, vars_, w)
    xs = [rng.normal(size=(T, D)) for T in (9, 30, 2, 17)]
    b = hip.Batch(ctx, xs)
    nll = b.loglik(gmm)

    def chain(n, skip):
        t = np.full((n, n), np.inf)
        for i in range(n):
            t[i, i] = rng.uniform(0.05, 1.0)
            if i + 1 < n:
                t[i + 1, i] = rng.uniform(0.5, 3.0)
            if skip and i + 2 < n and rng.random() < 0.5:
                t[i + 2, i] = rng.uniform(0.5, 3.0)
        return t

    # (a) one chain with skips: costs, path, end against the oracle
    n = 7
    t1 = chain(n, True)
    rows = rng.integers(0, S, size=n)
    lat = hip.Lattices(ctx, [graph(rows, t1, [0], [n - 1])])
    r = lat.viterbi(b, want_costs=True)
    for u, x in enumerate(xs):
        E = nll[b.offsets[u]:b.offsets[u + 1]][:, rows].T
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            costs, path = O.decode_states(E, np.zeros(n, dtype=bool), t1)
        fin = ~np.isinf(costs)
        np.testing.assert_array_equal(np.isinf(r["costs"][u]), ~fin)
        np.testing.assert_allclose(r["costs"][u][fin], costs[fin], rtol=1e-12)
        np.testing.assert_array_equal(r["paths"][u], path.reshape(-1, 2))
    # (b) 15 chains of 6 rows stacked (90 rows -> two lane groups), every chain with its own start / end
    W, n = 15, 6
    big = np.full((W * n, W * n), np.inf)
    chains = [chain(n, k % 2 == 0) for k in range(W)]
    rows = rng.integers(0, S, size=W * n)
    for k in range(W):
        big[k * n:(k + 1) * n, k * n:(k + 1) * n] = chains[k]
    st = hip.Lattices(ctx, [graph(rows, big, [k * n for k in range(W)], [k * n + n - 1 for k in range(W)])])
    r = st.viterbi(b, want_path=True)
    for u, x in enumerate(xs):
        ref_end = []
        for k in range(W):
            E = nll[b.offsets[u]:b.offsets[u + 1]][:, rows[k * n:(k + 1) * n]].T
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                costs, path = O.decode_states(E, np.zeros(n, dtype=bool), chains[k])
            ref_end.append((costs[-1, -1], path))
        got = r["end_cost"][u]
        ref = np.array([c for c, _ in ref_end])
        np.testing.assert_array_equal(np.isinf(got), np.isinf(ref))
        np.testing.assert_allclose(got[~np.isinf(ref)], ref[~np.isinf(ref)], rtol=1e-12)
        # best end: the LAST of equal minima (all-inf for the 2-frame utterance -> last chain)
        best = np.inf
        for k, c in enumerate(ref):
            if best >= c:
                best, bk = c, k
        assert r["best_end"][u] == bk
        if np.isfinite(best):
            np.testing.assert_array_equal(r["paths"][u] - [bk * n, 0], ref_end[bk][1].reshape(-1, 2))


def test_mfcc_golden(hip, ctx):
    """G15 (N3): PCM -> log mel filterbank -> cepstra, against the reference's mfcc_features; all sample formats."""
    g = load_golden("G15_mfcc")
    n = int(g["n"])
    idx16 = [i for i in range(n) if int(g["rate%d" % i]) == 16000]
    for dt, tol in ((np.int16, 1e-9), (np.float64, 1e-9), (np.float32, 1e-9)):   # int16 values are exact in all three
        fbs, mfs = hip.mfcc(ctx, [g["signal%d" % i].astype(dt) for i in idx16], 16000)
        for i, fb, mf in zip(idx16, fbs, mfs):
            assert fb.shape == g["fbank%d" % i].shape and mf.shape == g["mfcc%d" % i].shape
            np.testing.assert_allclose(fb, g["fbank%d" % i], rtol=tol, atol=tol)
            np.testing.assert_allclose(mf, g["mfcc%d" % i], rtol=tol, atol=tol)
    i8 = next(i for i in range(n) if int(g["rate%d" % i]) == 8000)
    fbs, mfs = hip.mfcc(ctx, [g["signal%d" % i8]], 8000)
    np.testing.assert_allclose(fbs[0], g["fbank%d" % i8], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(mfs[0], g["mfcc%d" % i8], rtol=1e-9, atol=1e-9)
    with pytest.raises(hip.BackendError):           # 48 kHz: 1200-sample frames exceed the reference's NFFT = 512
        hip.mfcc(ctx, [np.zeros(4800, dtype=np.int16)], 48000)


def test_mfcc_at_scale_matches_oracle(hip, ctx):
    """10 s of audio in ragged utterances: every frame against the numpy restatement; then the resident chain
    PCM -> 39-dim standardised features against oracle MFCC -> delta -> delta-delta -> standardize."""
    rng = np.random.default_rng(15)
    sigs = []
    for n in rng.integers(2000, 24000, size=12):
        t = np.arange(n) / 16000.0
        f0 = rng.uniform(100, 3000)
        sigs.append(np.round(5000 * np.sin(2 * np.pi * f0 * t * (1 + 0.3 * t)) + 1500 * rng.normal(size=n)).astype(np.int16))
    fbs, mfs = hip.mfcc(ctx, sigs, 16000)
    for x, fb, mf in zip(sigs, fbs, mfs):
        rfb, rmf = O.mfcc_features_signal(x, 16000)
        np.testing.assert_allclose(fb, rfb, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(mf, rmf, rtol=1e-9, atol=1e-9)
    for dt, tol in ((np.float64, 1e-7), (np.float32, 2e-5)):
        b = hip.Batch(ctx, pcm=sigs, sample_rate=16000, dtype=dt)
        assert b.D == 39 and b.U == len(sigs)
        for x, f in zip(sigs, b.features()):
            c = O.mfcc_features_signal(x, 16000)[1]
            d = O.delta_feature(c)
            ref = O.standardize(np.concatenate([c, d, O.delta_feature(d)], axis=1))
            np.testing.assert_allclose(f, ref, rtol=tol, atol=tol)
        b.close()


def test_forward_backward_chain_kernel_vs_generic_and_oracle(hip, ctx, monkeypatch):
    """One-word graphs (plain chains and the NES-wrapped forced-alignment lattices of packed_lattice) take the
    one-lane-per-utterance kernel when no matrices are requested: log P and the frame x state occupancies must equal
    the generic kernel's (GMMHMM_FB=generic) and the oracle's."""
    from sr.recognition.continuous_speech import packed_lattice
    g = load_golden("G3_isolated_decode_c2")
    means, vars_, w, trans = g["means"], g["vars"], g["w"], g["trans"]
    W, n, M, D = means.shape
    S = W * n
    gmm = hip.PackedGMM(ctx, means.reshape(S, M, D), vars_.reshape(S, M, D), w.reshape(S, M))
    rng = np.random.default_rng(5)
    xs, words = [], []
    for u in range(70):                       # more than one wave of utterances, ragged lengths incl. T = 1 and 2
        wd = int(rng.integers(0, W))
        T = [1, 2, 3][u] if u < 3 else int(rng.integers(6, 40))
        st = np.minimum(np.arange(T) * n // T, n - 1)
        xs.append(means[wd, st, 0] + np.sqrt(vars_[wd, st, 0]) * rng.normal(size=(T, D)))
        words.append(wd)
    plain = [graph(np.arange(n) + i * n, trans, [0], [n - 1]) for i in range(W)]
    wrapped = [packed_lattice([trans] * W, n, [[i]])[0] for i in range(W)]
    for graphs in (plain, wrapped):
        lat = hip.Lattices(ctx, graphs)
        b = hip.Batch(ctx, xs)
        nll = b.loglik(gmm)
        monkeypatch.delenv("GMMHMM_FB", raising=False)
        r_chain = lat.forward_backward(b, utt_lattice=words, want_occ=True)
        monkeypatch.setenv("GMMHMM_FB", "generic")
        r_gen = lat.forward_backward(b, utt_lattice=words, want_occ=True)
        monkeypatch.delenv("GMMHMM_FB", raising=False)
        np.testing.assert_allclose(r_chain["logp"], r_gen["logp"], rtol=1e-11)
        np.testing.assert_allclose(r_chain["occ"], r_gen["occ"], rtol=1e-9, atol=1e-12)
        long_enough = np.repeat(b.lengths >= n, b.lengths)     # a chain of n states needs n frames to reach its end
        np.testing.assert_allclose(r_chain["occ"][long_enough].sum(axis=1), 1.0, rtol=1e-9)
        assert np.all(r_chain["occ"][~long_enough] == 0) and np.all(np.isinf(r_chain["logp"][:3]))
        for u in (0, 1, 2, 5, 69):            # oracle on a few utterances
            gr = graphs[words[u]]
            rs = np.asarray(gr["row_state"])
            R = len(rs)
            dense = np.full((R, R), np.inf)
            dense[np.asarray(gr["arc_to"]), np.asarray(gr["arc_from"])] = gr["arc_cost"]
            x_nll = nll[b.offsets[u]:b.offsets[u + 1]]
            E = np.where(rs[:, None] >= 0, x_nll[:, np.maximum(rs, 0)].T, 0.0)
            al, be, ga, logp = O.forward_backward(E, rs < 0, dense, gr["end_rows"])
            if np.isinf(logp):
                assert np.isinf(r_chain["logp"][u])
                continue
            np.testing.assert_allclose(r_chain["logp"][u], logp, rtol=1e-10)
            occ = np.zeros((len(x_nll), S))
            for r in range(R):
                if rs[r] >= 0:
                    occ[:, rs[r]] += ga[r]
            np.testing.assert_allclose(r_chain["occ"][b.offsets[u]:b.offsets[u + 1]], occ, rtol=1e-8, atol=1e-12)
        # fp32 likelihood matrix through both kernels
        b32 = hip.Batch(ctx, xs, dtype=np.float32)
        b32.loglik(gmm, fetch=False)
        c32 = lat.forward_backward(b32, utt_lattice=words, want_occ=True)
        monkeypatch.setenv("GMMHMM_FB", "generic")
        g32 = lat.forward_backward(b32, utt_lattice=words, want_occ=True)
        monkeypatch.delenv("GMMHMM_FB", raising=False)
        np.testing.assert_allclose(c32["logp"], g32["logp"], rtol=1e-11)
        np.testing.assert_allclose(c32["occ"], g32["occ"], rtol=1e-9, atol=1e-12)
        fin = np.isfinite(r_chain["logp"])
        np.testing.assert_allclose(c32["logp"][fin], r_chain["logp"][fin], rtol=1e-5)
        b32.close()
        b.close()
        lat.close()


def test_viterbi_configs3_shape_1024_rows(hip, ctx):
    """configs[3] shape (64 word models x 16 states = 1024 rows / 1024 states): the stacked chain graph through
    the chain kernel, and the same graph per utterance (lean / generic kernel selection with its LDS budget)."""
    rng = np.random.default_rng(64)
    W, n, M, D = 64, 16, 1, 4
    S = W * n
    means = rng.normal(size=(S, M, D)) * 3
    vars_ = rng.uniform(0.5, 1.5, size=(S, M, D))
    w = np.ones((S, M))
    trans = np.full((n, n), np.inf)
    for i in range(n):
        trans[i, i] = -np.log(0.7) if i < n - 1 else 0.0
        if i < n - 1:
            trans[i + 1, i] = -np.log(0.3)
    big = np.full((S, S), np.inf)
    for i in range(W):
        big[i * n:(i + 1) * n, i * n:(i + 1) * n] = trans
    g = graph(np.arange(S), big, [i * n for i in range(W)], [i * n + n - 1 for i in range(W)])
    xs, words = [], []
    for u in range(6):
        wd = int(rng.integers(0, W))
        T = int(rng.integers(20, 40))
        st = np.minimum(np.arange(T) * n // T, n - 1)
        xs.append(means[wd * n + st, 0] + np.sqrt(vars_[wd * n + st, 0]) * rng.normal(size=(T, D)))
        words.append(wd)
    gmm = hip.PackedGMM(ctx, means, vars_, w)
    b = hip.Batch(ctx, xs)
    nll = b.loglik(gmm)
    lat = hip.Lattices(ctx, [g])
    r1 = lat.viterbi(b, want_path=False)
    r2 = lat.viterbi(b, utt_lattice=np.zeros(len(xs), dtype=np.int32), want_path=True)
    np.testing.assert_array_equal(r1["end_cost_flat"], r2["end_cost_flat"])
    ec = r1["end_cost_flat"].reshape(len(xs), W)
    assert list(np.argmin(ec, axis=1)) == words
    for u, wd in enumerate(words):
        E = nll[b.offsets[u]:b.offsets[u + 1], wd * n:(wd + 1) * n].T
        costs, path = O.decode_states(E, np.zeros(n, dtype=bool), trans)
        np.testing.assert_allclose(ec[u, wd], costs[-1, -1], rtol=1e-12)
        got = r2["paths"][u]
        assert r2["best_end"][u] == wd
        np.testing.assert_array_equal(got[:, 0] - wd * n, path[:, 0])


def test_empty_and_degenerate_batches(hip, ctx):
    """No utterances at all, utterances without frames, one-frame utterances: every entry point returns empty / the
    reference's degenerate results instead of faulting."""
    rng = np.random.default_rng(9)
    S, M, D, n = 6, 2, 5, 3
    gmm = hip.PackedGMM(ctx, rng.normal(size=(S, M, D)), rng.uniform(0.5, 1.5, size=(S, M, D)), rng.dirichlet(np.ones(M), size=S))
    trans = np.full((n, n), np.inf)
    for i in range(n):
        trans[i, i] = 0.2
        if i + 1 < n:
            trans[i + 1, i] = 1.0
    lat = hip.Lattices(ctx, [graph(np.arange(n), trans, [0], [n - 1])])
    # (a) an empty batch
    b0 = hip.Batch(ctx, feats=np.zeros((0, D)), offsets=np.array([0], dtype=np.int64))
    assert b0.U == 0 and b0.N == 0
    assert b0.loglik(gmm).shape == (0, S)
    r = lat.viterbi(b0)
    assert len(r["paths"]) == 0 and len(r["best_end"]) == 0
    assert len(lat.forward_backward(b0)["logp"]) == 0
    b0.close()
    # (b) empty and one-frame utterances between ordinary ones
    xs = [rng.normal(size=(7, D)), np.zeros((0, D)), rng.normal(size=(1, D)), np.zeros((0, D)), rng.normal(size=(4, D))]
    b = hip.Batch(ctx, xs)
    nll = b.loglik(gmm)
    assert nll.shape == (12, S)
    r = lat.viterbi(b, want_path=True)
    assert r["best_end"][1] == -1 and r["best_end"][3] == -1 and len(r["paths"][1]) == 0 and len(r["paths"][2]) == 0
    rl = lat.viterbi_labels(b, np.zeros(n, dtype=np.int32))
    assert [len(l) for l in rl["labels"]] == [1, 0, 0, 0, 1]
    fb = lat.forward_backward(b, want_occ=True)
    assert np.isneginf(fb["logp"][1]) and np.isneginf(fb["logp"][3]) and np.isneginf(fb["logp"][2])   # 1 frame < 3 states
    assert np.isfinite(fb["logp"][0]) and np.isfinite(fb["logp"][4])
    np.testing.assert_allclose(fb["occ"][:7].sum(axis=1), 1.0, rtol=1e-9)
    b.close()
    lat.close()
    gmm.close()


@pytest.mark.parametrize("S,M,D", [(50, 8, 39), (12, 3, 13), (40, 1, 13), (9, 32, 7), (30, 16, 24)])
def test_loglik_subset_equals_full_on_the_ranges(hip, ctx, S, M, D):
    """gh_loglik_subset: every utterance gets the likelihoods of its own state range, bit-identical to gh_loglik."""
    rng = np.random.default_rng(S + M)
    gmm = hip.PackedGMM(ctx, rng.normal(size=(S, M, D)), rng.uniform(0.5, 1.5, size=(S, M, D)), rng.dirichlet(np.ones(M), size=S))
    lens = [1, 31, 32, 33, 64, 100, 7, 150]
    xs = [rng.normal(size=(t, D)) for t in lens]
    lo = rng.integers(0, S - 1, size=len(xs)).astype(np.int32)
    hi = np.minimum(S, lo + rng.integers(1, 7, size=len(xs))).astype(np.int32)
    lo[0], hi[0] = 0, min(S, 5)
    lo[-1], hi[-1] = max(0, S - 5), S
    for dt in (np.float64, np.float32):
        full = hip.Batch(ctx, xs, dtype=dt)
        ref = full.loglik(gmm)
        sub = hip.Batch(ctx, xs, dtype=dt)
        got = sub.loglik(gmm, state_ranges=(lo, hi))
        for u in range(len(xs)):
            a, b_ = sub.offsets[u], sub.offsets[u + 1]
            np.testing.assert_array_equal(got[a:b_, lo[u]:hi[u]], ref[a:b_, lo[u]:hi[u]])
        full.close()
        sub.close()
    gmm.close()


def test_loglik_extreme_parameter_ranges(hip, ctx):
    """Tiny and huge variances, offsets far from zero, weights spanning 12 decades, far-away frames: the scaled-log-domain
    epilogue of the MFMA kernel against the oracle's log-domain arithmetic (fp64).  With the log domain kept throughout
    (set_compat(underflow=False)) every cost is finite; with the reference's rule (a context's default) exactly the costs
    beyond -ln 2^-1075 = 745.13 are +inf and the others are the same numbers."""
    rng = np.random.default_rng(77)
    S, M, D, N = 6, 8, 13, 200
    means = rng.normal(size=(S, M, D)) * np.array([1e-3, 1.0, 30.0, 1e3, 1.0, 1.0])[:, None, None]
    vars_ = 10.0 ** rng.uniform(-6, 6, size=(S, M, D))
    vars_[4] = 10.0 ** rng.uniform(-1, 1, size=(M, D))
    w = 10.0 ** rng.uniform(-12, 0, size=(S, M))
    X = rng.normal(size=(N, D)) * 3.0
    X[:50] += means[3, 0]            # near the far-away state
    X[50:60] *= 1e3                  # very unlikely frames: costs of ~1e6 and more
    ref = O.gmm_neg_loglik_batch(X, means, vars_, w)
    gmm = hip.PackedGMM(ctx, means, vars_, w)
    b = hip.Batch(ctx, feats=X, offsets=[0, N])
    ruled = b.loglik(gmm).copy()
    ctx.set_compat(underflow=False)
    try:
        got = b.loglik(gmm).copy()
    finally:
        ctx.set_compat(underflow=True)
    assert np.all(np.isfinite(got))
    # the contraction sum_k P_k z_k cancels terms of size ~max|x^2/var|: compare to that scale
    scale = np.maximum(np.abs(ref), ((X[:, None, None, :] ** 2 + means[None] ** 2) / vars_[None]).sum(axis=3).max(axis=2))
    assert np.max(np.abs(got - ref) / scale) < 1e-12
    # the reference's rule: +inf where the LARGEST weighted density of the state is below 2^-1075 (the sum's other terms are
    # smaller still), the log-domain number elsewhere; costs within rounding of the threshold may fall on either side
    logc = np.log(w) - 0.5 * (D * np.log(2 * np.pi) + np.log(vars_).sum(axis=2))
    best = (logc[None] - 0.5 * (((X[:, None, None, :] - means[None]) ** 2) / vars_[None]).sum(axis=3)).max(axis=2)
    thr = 1075 * np.log(2.0)
    clear = np.abs(best + thr) > 1e-6 * scale
    np.testing.assert_array_equal(np.isinf(ruled)[clear], (best < -thr)[clear])
    np.testing.assert_array_equal(ruled[np.isfinite(ruled)], got[np.isfinite(ruled)])
    assert np.isinf(ruled).any() and np.isfinite(ruled).any()
    b.close()
    gmm.close()


@pytest.mark.parametrize("W,n,M,D", [(10, 5, 8, 39), (4, 3, 4, 13), (3, 8, 8, 6), (5, 2, 1, 7), (3, 5, 16, 13), (2, 4, 5, 24)])
def test_bw_statistics_fused_matrix_core_path_equals_generic(hip, ctx, W, n, M, D):
    """One-word chain graphs: when nobody asks for the [N, S] occupancy matrix the forward-backward keeps gamma compact
    and gh_bw_accumulate runs the fused kernel (densities and accumulation on the matrix cores, gh_bw_fused.hip);
    asking for the matrix takes the generic pair of kernels.  Same statistics (1e-9), also for mixtures padded to 8
    components, odd numbers of states, a zero-weight component, and -- M = 16 -- shapes the fused kernel hands back."""
    rng = np.random.default_rng(W * 100 + n * 10 + M)
    S = W * n
    means = rng.normal(size=(S, M, D)) * 2.0
    vars_ = rng.uniform(0.5, 1.5, size=(S, M, D))
    w = rng.dirichlet(np.ones(M), size=S)
    if M > 1:
        w[1, 0] = 0.0                                     # a switched-off component
    trans = np.full((n, n), np.inf)
    for i in range(n):
        trans[i, i] = 0.4 if i < n - 1 else 0.0
        if i < n - 1:
            trans[i + 1, i] = 1.1
    xs, words = [], []
    for u in range(70):
        wd = int(rng.integers(0, W))
        T = int(rng.integers(n + 1, 150))
        st = np.minimum(np.arange(T) * n // T, n - 1)
        xs.append(means[wd * n + st, rng.integers(0, M, T)] + rng.normal(size=(T, D)))
        words.append(wd)
    gmm = hip.PackedGMM(ctx, means, vars_, w)
    lat = hip.Lattices(ctx, [graph(np.arange(n) + i * n, trans, [0], [n - 1]) for i in range(W)])
    b = hip.Batch(ctx, xs)
    b.loglik(gmm, fetch=False)
    r_full = lat.forward_backward(b, utt_lattice=words, want_occ=True, fetch_occ=True)
    generic = b.bw_accumulate(gmm)
    r_cmp = lat.forward_backward(b, utt_lattice=words, want_occ=True, fetch_occ=False)
    fused = b.bw_accumulate(gmm)
    np.testing.assert_array_equal(r_cmp["logp"], r_full["logp"])
    scale = np.abs(generic).max(axis=2, keepdims=True) + 1e-300
    assert np.max(np.abs(fused - generic) / scale) < 1e-9
    np.testing.assert_allclose(fused[:, :, 0].sum(), sum(len(x) for x in xs), rtol=1e-9)
    visited = sorted(set(words))
    assert np.all(fused[[s for s in range(S) if s // n not in visited]] == 0)
    # the same again through a device-side destination (what the trainer hands to RCCL) and with an occupancy floor
    floor = 1e-3
    lat.forward_backward(b, utt_lattice=words, want_occ=True, fetch_occ=True)
    g2 = b.bw_accumulate(gmm, occ_floor=floor)
    lat.forward_backward(b, utt_lattice=words, want_occ=True, fetch_occ=False)
    f2 = b.bw_accumulate(gmm, occ_floor=floor)
    assert np.max(np.abs(f2 - g2) / scale) < 1e-9
    b.close()
    lat.close()
    gmm.close()


def test_viterbi_lattice_beam(hip, ctx):
    """N4, second half: rank beam per column in gh_viterbi (gh_lattices_set_beam).  (i) On a chain graph it reproduces the
    reference's own dtw beam results (G5: surviving cells, costs, path); (ii) on K-layer word lattices and a loop
    grammar it equals the oracle's lattice beam cell for cell, paths bit-exact; (iii) beam >= rows == no beam
    (bit-identical to the unpruned kernels), pruned cost >= unpruned cost, labels follow."""
    from sr.recognition.continuous_speech import packed_lattice, packed_loop_lattice
    g = load_golden("G5_dtw")
    x, y, var, trans = g["x"], g["y"], g["var"], g["trans"]
    n, D = y.shape
    gmm = hip.PackedGMM(ctx, y[:, None, :], var[:, None, :], np.ones((n, 1)))
    b = hip.Batch(ctx, [x])
    b.loglik(gmm, fetch=False)
    lat = hip.Lattices(ctx, [graph(np.arange(n), trans, [0], [n - 1])])
    lat.set_beam(3)
    r = lat.viterbi(b, want_path=True, want_costs=True)
    ref = g["costs_beam3"]                                        # mahalanobis distance == -log N(x; y_i, var_i)
    alive = np.isfinite(ref) & (ref != -1)
    np.testing.assert_array_equal(np.isfinite(r["costs"][0][:, :-1]), alive[:, :-1])
    np.testing.assert_allclose(r["costs"][0][:, :-1][alive[:, :-1]], ref[:, :-1][alive[:, :-1]], rtol=1e-10)
    np.testing.assert_array_equal(r["paths"][0], g["path_beam3"])
    lat.close(); b.close(); gmm.close()
    # ---- lattices against the oracle
    rng = np.random.default_rng(41)
    W, ns, M, D = 6, 3, 2, 5
    means = rng.normal(size=(W, ns, M, D)) * 2.0
    vars_ = rng.uniform(0.5, 1.5, size=(W, ns, M, D))
    w = rng.dirichlet(np.ones(M), size=(W, ns))
    tr = np.full((ns, ns), np.inf)
    for i in range(ns):
        tr[i, i] = 0.3 if i < ns - 1 else 0.0
        if i < ns - 1:
            tr[i + 1, i] = 1.3
    gmm = hip.PackedGMM(ctx, means.reshape(W * ns, M, D), vars_.reshape(W * ns, M, D), w.reshape(W * ns, M))
    xs = []
    for u in range(12):
        segs = []
        for wd in rng.integers(0, W, size=3):
            Tw = int(rng.integers(ns + 1, 12))
            st = np.minimum(np.arange(Tw) * ns // Tw, ns - 1)
            segs.append(means[wd, st, rng.integers(0, M, Tw)] + rng.normal(size=(Tw, D)))
        xs.append(np.concatenate(segs))
    b = hip.Batch(ctx, xs)
    nll = b.loglik(gmm, fetch=True)
    import warnings as _w
    for gr in (packed_lattice([tr] * W, ns, [list(range(W))] * 3)[0], packed_loop_lattice([tr] * W, ns, 0.5)[0]):
        R = len(gr["row_state"])
        dense = np.full((R, R), np.inf)
        dense[gr["arc_to"], gr["arc_from"]] = gr["arc_cost"]
        is_nes = gr["row_state"] < 0
        lat = hip.Lattices(ctx, [gr])
        base = lat.viterbi(b, want_path=True)
        row_word = np.where(gr["row_state"] >= 0, gr["row_state"] // ns, -1).astype(np.int32)
        for beam in (4, 9, R):
            lat.set_beam(beam)
            r = lat.viterbi(b, want_path=True, want_costs=True)
            rl = lat.viterbi_labels(b, row_word)
            for u in range(b.U):
                E = np.zeros((R, len(xs[u])))
                E[~is_nes] = nll[b.offsets[u]:b.offsets[u + 1]][:, gr["row_state"][~is_nes]].T
                with _w.catch_warnings():
                    _w.simplefilter("ignore")
                    try:
                        costs, path = O.decode_states(E, is_nes, dense, end_points=[[int(e), -1] for e in gr["end_rows"]], beam=beam)
                    except RuntimeError:
                        continue                                  # every end pruned away: the reference-style walk does not terminate
                np.testing.assert_array_equal(np.isfinite(r["costs"][u]), np.isfinite(costs))
                fin = np.isfinite(costs)
                np.testing.assert_allclose(r["costs"][u][fin], costs[fin], rtol=1e-12)
                if np.isfinite(costs[np.asarray(gr["end_rows"]), -1]).any():
                    np.testing.assert_array_equal(r["paths"][u], path)
                    from sr.recognition.batch import path_to_words
                    assert [int(v) for v in rl["labels"][u]] == path_to_words(path, gr["row_state"], ns)
                assert np.min(r["end_cost"][u]) >= np.min(base["end_cost"][u])
            if beam == R:
                np.testing.assert_array_equal(r["end_cost_flat"], base["end_cost_flat"])
                for u in range(b.U):
                    np.testing.assert_array_equal(r["paths"][u], base["paths"][u])
        lat.set_beam(None)
        again = lat.viterbi(b, want_path=True)
        np.testing.assert_array_equal(again["end_cost_flat"], base["end_cost_flat"])
        lat.close()
    b.close()
    gmm.close()


def test_decode_hmm_states_beam_kwarg(hip, ctx):
    """The mirror API's extension kwarg: decode_hmm_states(..., beam=k); the default is the reference's decode."""
    import sr.recognition as R
    g = load_golden("G5_dtw")
    x, y, var, trans = g["x"], g["y"], g["var"], g["trans"]
    states = [R.GMM(y[i], var[i], 1) for i in range(len(y))]
    c0, p0 = R.decode_hmm_states(x, states, trans)
    c3, p3 = R.decode_hmm_states(x, states, trans, beam=3)
    np.testing.assert_array_equal(p3, g["path_beam3"])
    assert np.isinf(c3[:, :-1]).sum() > np.isinf(c0[:, :-1]).sum() and c3[-1, -1] >= c0[-1, -1]
    cb, pb = R.decode_hmm_states(x, states, trans, beam=np.inf)
    np.testing.assert_array_equal(cb, c0)
    np.testing.assert_array_equal(pb, p0)


@pytest.mark.parametrize("n,beam", [(65, 0), (100, 12), (300, 0), (300, 40)])
def test_dtw_long_templates(n, beam):
    """gh_dtw beyond one wave: templates of 65..300 rows (several waves per utterance, 16-bit back-pointers above 255
    rows) against the oracle's reference-shaped DP -- costs (with the beam's -1 / +inf marks), paths."""
    from sr.recognition import _hip
    from oracle import ref_numpy as O
    ctx = _hip.default_context()
    rng = np.random.default_rng(n + beam)
    D = 5
    y = rng.normal(size=(n, D))
    trans = np.full((n, n), np.inf)
    for i in range(n):
        trans[i, i] = rng.uniform(0.1, 0.5)
        if i + 1 < n:
            trans[i + 1, i] = rng.uniform(0.3, 1.0)
        if i + 2 < n:
            trans[i + 2, i] = rng.uniform(0.8, 2.0)
    xs = [y[np.minimum(np.arange(T) * n // T, n - 1)] + 0.3 * rng.normal(size=(T, D)) for T in (n // 2 + 3, n + 7)]
    b = _hip.Batch(ctx, xs)
    costs, paths = b.dtw(trans, y=y, beam=beam)
    for u, x in enumerate(xs):
        E = np.array([[O.euclid(x[j], y[i]) for j in range(len(x))] for i in range(n)])
        with np.errstate(invalid="ignore"):
            rc, rp = O.dtw(E, trans, beam=beam if beam else np.inf)
        np.testing.assert_allclose(costs[u], rc, rtol=1e-12)
        np.testing.assert_array_equal(paths[u], np.asarray(rp).reshape(-1, 2))
    b.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("M,D", [(4, 5), (1, 13), (8, 39), (32, 39)])
def test_compat_linear_domain_underflow(hip, dtype, M, D):
    """GMM.evaluate sums w pdf in the LINEAR domain (hmm_state.py:114-120): once every term underflows fp64 the state
    costs -log 0 = +inf.  A context reproduces the +inf by default (gh_ctx_set_compat bit 0; VERDICT r2 'missing' 5) --
    checked against the oracle's linear-domain evaluate frame by frame, together with the frames that do not underflow;
    set_compat(underflow=False) keeps the kernels in the log domain: the finite cost everywhere."""
    ctx = hip.Context(0)                        # a context of its own: the switch is a property of the context
    rng = np.random.default_rng(M * 100 + D)
    S, N = 6, 96
    means, vars_ = rng.normal(size=(S, M, D)), rng.uniform(0.5, 1.5, size=(S, M, D))
    w = rng.dirichlet(np.ones(M), size=S)
    X = rng.normal(size=(N, D))
    X[::3] += 30.0 * np.sign(rng.normal(size=(N // 3, D)))     # a third of the frames far from every mean: log p << -745
    X = X.astype(dtype).astype(np.float64)                      # (what both sides see)
    lin = np.array([[O.gmm_evaluate(x, means[s], vars_[s], w[s]) for s in range(S)] for x in X])
    logdom = O.gmm_neg_loglik_batch(X, means, vars_, w)
    assert np.isinf(lin[::3]).all() and np.isfinite(lin[1::3]).all() and np.isfinite(logdom).all()
    gmm = hip.PackedGMM(ctx, means, vars_, w)
    b = hip.Batch(ctx, feats=X, offsets=[0, N], dtype=dtype)
    rtol = 1e-10 if dtype == np.float64 else 1e-3
    got = b.loglik(gmm).copy()                                               # default: the reference's rule
    np.testing.assert_array_equal(np.isinf(got), np.isinf(lin))
    fin = np.isfinite(lin)
    np.testing.assert_allclose(got[fin], lin[fin], rtol=rtol)
    ctx.set_compat(underflow=False)
    np.testing.assert_allclose(b.loglik(gmm), logdom, rtol=rtol)             # log domain: finite everywhere
    ctx.set_compat(underflow=True)
    np.testing.assert_array_equal(b.loglik(gmm), got)
    b.close(); gmm.close(); ctx.close()


@pytest.mark.parametrize("M,D", [(1, 13), (4, 13), (8, 39)])
def test_compat_underflow_with_normalisers_above_one(hip, M, D):
    """the reference loses a term already in np.exp(-q/2) -- below ln 2^-1075 WHATEVER the normaliser is
    (hmm_state.py:36-45) -- so with tight variances (log(w norm) > 0) a frame costs +inf although the total logarithm of
    its largest term is still representable.  Frames with -q/2 in (-790, -746), variances of 0.01: +inf exactly where the
    oracle's linear-domain evaluate says so (the band the test on the total logarithm alone would miss), finite elsewhere."""
    ctx = hip.Context(0)
    rng = np.random.default_rng(7 * M + D)
    S, N = 5, 120
    means = rng.normal(size=(S, M, D)) * 0.002
    vars_ = np.full((S, M, D), 0.01) * rng.uniform(0.9, 1.1, size=(S, M, D))
    w = rng.dirichlet(np.ones(M), size=S)
    X = rng.normal(size=(N, D)) * 0.1
    for i in range(0, N, 2):                                    # every other frame: -q/2 between -700 and -790
        d = rng.normal(size=D)
        X[i] = d / np.linalg.norm(d) * np.sqrt(2 * (700.0 + 90.0 * i / N) * 0.01)
    with np.errstate(divide="ignore"):
        lin = np.array([[O.gmm_evaluate(x, means[s], vars_[s], w[s]) for s in range(S)] for x in X])
    logdom = O.gmm_neg_loglik_batch(X, means, vars_, w)
    lognorm = -0.5 * (D * np.log(2 * np.pi) + np.sum(np.log(vars_), axis=2)) + np.log(w)
    assert (lognorm > 5).all()
    band = np.isinf(lin) & (logdom < 745.0)                     # +inf in the reference, total logarithm above the threshold
    assert band.sum() >= 20 and np.isfinite(lin).sum() >= N * S // 2
    gmm = hip.PackedGMM(ctx, means, vars_, w)
    b = hip.Batch(ctx, feats=X, offsets=[0, 40, N])
    got = b.loglik(gmm)
    np.testing.assert_array_equal(np.isinf(got), np.isinf(lin))
    # (values: where the reference's exp(-q/2) is a normal number -- beyond ~708 it works on denormals, whose few bits
    #  show in the cost; the +inf pattern above is exact everywhere)
    q2min = np.array([[0.5 * np.min(np.sum((x - means[s]) ** 2 / vars_[s], axis=1)) for s in range(S)] for x in X])
    fin = np.isfinite(lin) & (q2min < 700)
    assert fin.sum() >= N * S // 3
    np.testing.assert_allclose(got[fin], lin[fin], rtol=1e-10)
    # the same entries through a subset launch and after an in-place parameter update (device-side re-pack)
    lo, hi = np.zeros(2, dtype=np.int32), np.full(2, S, dtype=np.int32)
    got2 = b.loglik(gmm, state_ranges=(lo, hi))
    np.testing.assert_array_equal(np.isinf(got2), np.isinf(lin))
    gmm.update(means, vars_ * 100.0, w)                         # ordinary variances: nothing underflows early any more
    got3 = b.loglik(gmm)
    assert np.isfinite(got3).all()
    gmm.update(means, vars_, w)
    np.testing.assert_array_equal(np.isinf(b.loglik(gmm)), np.isinf(lin))
    b.close(); gmm.close(); ctx.close()


@pytest.mark.parametrize("n", [9, 12, 16])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_chain_forward_backward_with_16_lanes_equals_generic(hip, ctx, n, dtype, monkeypatch):
    """Chains of 9-16 rows (configs[3]: 16 states per word) run the chain forward-backward with 16 lanes per utterance;
    log P, the [N, S] occupancy matrix and the expected self transitions == the any-graph kernel, skip arcs included,
    on fp64 and fp32 likelihoods."""
    from sr.recognition.continuous_speech import packed_lattice
    rng = np.random.default_rng(40 + n)
    W, M, D, U = 3, 4, 13, 40
    means = rng.normal(size=(W * n, M, D)) * 2
    vars_ = rng.uniform(0.5, 1.5, size=(W * n, M, D))
    w = rng.dirichlet(np.ones(M), size=W * n)
    trans = np.full((n, n), np.inf)
    for i in range(n):
        trans[i, i] = 0.3
        if i:
            trans[i, i - 1] = 1.2
        if i > 1:
            trans[i, i - 2] = 3.0
    words = rng.integers(0, W, size=U)
    xs = []
    for wd in words:
        T = int(rng.integers(n, 5 * n))
        st = wd * n + np.minimum(np.arange(T) * n // T, n - 1)
        xs.append(means[st, rng.integers(0, M, size=T)] + rng.normal(size=(T, D)))
    lat = hip.Lattices(ctx, [packed_lattice([trans] * W, n, [[k]])[0] for k in range(W)])
    assert "fb_chain" in lat.forms()
    gmm = hip.PackedGMM(ctx, means, vars_, w)
    b = hip.Batch(ctx, xs, dtype=dtype)
    b.loglik(gmm, fetch=False)
    ul = words.astype(np.int32)
    got = lat.forward_backward(b, utt_lattice=ul, want_occ=True, want_self_xi=True)
    monkeypatch.setenv("GMMHMM_FB", "generic")
    ref = lat.forward_backward(b, utt_lattice=ul, want_occ=True, want_self_xi=True)
    monkeypatch.delenv("GMMHMM_FB")
    np.testing.assert_allclose(got["logp"], ref["logp"], rtol=1e-10)
    np.testing.assert_allclose(got["occ"], ref["occ"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(got["self_xi"], ref["self_xi"], rtol=1e-8, atol=1e-10)
    for h in (b, lat, gmm):
        h.close()


def test_batch_jitter_makes_the_copies_of_a_tiled_batch_distinct():
    """gh_batch_jitter: independent N(0, scale^2) noise per feature from a counter-based generator -- deterministic for a
    seed, different for another, mean / variance as asked, fp32 and fp64; the likelihoods of the jittered batch are those
    of its own (changed) frames."""
    from sr.recognition import _hip
    ctx = _hip.default_context()
    rng = np.random.default_rng(4)
    D, T = 7, 500
    X = rng.normal(size=(T, D))
    gm = _hip.PackedGMM(ctx, np.zeros((1, 1, D)), np.ones((1, 1, D)), np.ones((1, 1)))

    def frames_of(b):        # recover sum x^2 per frame from the unit Gaussian's cost
        return 2.0 * (b.loglik(gm)[:, 0].astype(np.float64) - 0.5 * D * np.log(2 * np.pi))
    for dt, tol in ((np.float64, 1e-12), (np.float32, 2e-5)):
        base = _hip.Batch(ctx, feats=X.astype(dt), offsets=[0, T], dtype=dt)
        a = base.tile(40).jitter(seed=11, scale=0.5)
        b = base.tile(40).jitter(seed=11, scale=0.5)
        c = base.tile(40).jitter(seed=12, scale=0.5)
        qa, qb, qc, q0 = frames_of(a), frames_of(b), frames_of(c), frames_of(base)
        np.testing.assert_array_equal(qa, qb)
        assert np.mean(qa != qc) > 0.999
        # E[sum (x + e)^2 - sum x^2] = D scale^2; copies differ from each other
        d = qa.reshape(40, T) - q0[None, :]
        np.testing.assert_allclose(d.mean(), D * 0.25, rtol=0.05)
        assert np.mean(qa.reshape(40, T)[0] != qa.reshape(40, T)[1]) > 0.999
        # scale 0 changes nothing
        z = base.tile(3).jitter(seed=5, scale=0.0)
        np.testing.assert_allclose(frames_of(z).reshape(3, T), np.tile(q0, (3, 1)), rtol=tol, atol=tol)
        for h in (a, b, c, z, base):
            h.close()
    gm.close()
