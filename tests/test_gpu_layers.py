# -*- coding: utf-8 -*-
"""The layer-form lattice kernel (gh_viterbi_layers.hip: one wave per utterance, lane = (layer, word), states in
registers, decision bits as back-pointers) against the reference's goldens, the oracle and the row-per-lane lean
kernel, which implements the same decode_hmm_states semantics (decode.py:80-146) by an entirely different route.

`Lattices.viterbi(batch)` with ONE graph for the whole batch takes the layer-form kernel when the graph is a K-layer
word lattice; passing `utt_lattice` (all zeros) makes the call non-uniform and routes it to the lean kernel."""
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


def word_trans(rng, n, skip=False, last_self=0.0):
    t = np.full((n, n), np.inf)
    for i in range(n):
        t[i, i] = rng.uniform(0.05, 0.6) if i < n - 1 else last_self
        if i < n - 1:
            t[i + 1, i] = rng.uniform(0.8, 2.5)
        if skip and i < n - 2 and rng.random() < 0.6:
            t[i + 2, i] = rng.uniform(1.5, 4.0)
    return t


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_layers_kernel_reference_goldens(hip, ctx, dtype):
    """G4 (captured from the reference's decode_hmm_states): end costs, chosen end and BIT-EXACT paths for the
    K = 1, 2, 3, 7 lattices, each decoded as a one-graph batch (layer-form kernel)."""
    g = load_golden("G4_lattice_decode")
    means, vars_, w = g["means"], g["vars"], g["w"]
    W, n, M, D = means.shape
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, M, D), vars_.reshape(W * n, M, D), w.reshape(W * n, M))
    for K in (1, 2, 3, 7):
        p = "K%d_" % K
        rw, rs = g[p + "row_word"], g[p + "row_state"]
        graph = dict(row_state=np.where(rw < 0, -1, rw * n + rs), arc_to=g[p + "arc_to"], arc_from=g[p + "arc_from"],
                     arc_cost=g[p + "arc_cost"], start_rows=[0], end_rows=g[p + "ends"])
        lat = hip.Lattices(ctx, [graph])
        b = hip.Batch(ctx, [g[p + "x"]], dtype=dtype)
        b.loglik(gmm, fetch=False)
        r = lat.viterbi(b, want_path=True)
        ref = g[p + "costs"]
        ends = np.asarray(g[p + "ends"])
        np.testing.assert_allclose(r["end_cost"][0], ref[ends, -1], rtol=1e-10 if dtype == np.float64 else 1e-5)
        np.testing.assert_array_equal(r["paths"][0], g[p + "path"])
        assert O.path_to_words(r["paths"][0], rw < 0, rw) == list(g[p + "digits"])
        row_word = np.where(rw < 0, -1, rw).astype(np.int32)
        rl = lat.viterbi_labels(b, row_word, max_labels=K + 1)
        assert [int(v) for v in rl["labels"][0]] == list(g[p + "digits"])
        b.close()
        lat.close()


@pytest.mark.parametrize("W,n,K,skip", [(10, 5, 7, False), (11, 5, 7, False), (1, 2, 1, False), (3, 2, 8, False),
                                        (16, 3, 4, True), (5, 8, 5, True), (7, 4, 3, False), (10, 5, 2, True),
                                        (4, 6, 6, False), (2, 7, 8, True),
                                        # wide word models: 12 states without skip arcs still pack two register sets into
                                        # a 32-bit decision word; the others take 64-bit words
                                        (4, 12, 3, False), (3, 12, 8, True), (5, 16, 2, False), (2, 16, 8, True),
                                        (6, 16, 5, False),
                                        # more than 8 layers (ten-digit strings): four register sets, up to 8 states per word
                                        (10, 5, 10, False), (4, 3, 16, True), (3, 8, 12, True), (2, 2, 9, False),
                                        (5, 8, 16, False),
                                        # more than 16 words per layer (up to 64): the wide kernel, lane = word, the layers one
                                        # after the other in the lane's registers
                                        (17, 5, 7, False), (33, 3, 8, True), (64, 5, 3, False), (40, 8, 3, True), (20, 2, 1, False)])
def test_layers_kernel_equals_lean_kernel(hip, ctx, W, n, K, skip):
    """Random word models (per-word transition costs, optional skip arcs), utterances from far too short to long:
    end costs BITWISE equal, same chosen end, same paths -- including the unreachable cases, where every candidate is
    +inf and the winner is decided by the candidate order alone."""
    from sr.recognition.continuous_speech import packed_lattice
    rng = np.random.default_rng(1000 * W + 10 * n + K)
    M, D = 2, 6
    means = rng.normal(size=(W, n, M, D)) * 2.0
    vars_ = rng.uniform(0.5, 1.5, size=(W, n, M, D))
    w = rng.dirichlet(np.ones(M), size=(W, n))
    wt = [word_trans(rng, n, skip, last_self=rng.uniform(0.0, 0.3)) for _ in range(W)]
    xs = []
    for u in range(60):
        if u < 12:
            T = int(rng.integers(2, K * (n - 1) + 2))              # too short for K words: unreachable ends
            xs.append(rng.normal(size=(T, D)) * 2.0)
            continue
        words = rng.integers(0, W, size=K)
        segs = []
        for wd in words:
            Tw = int(rng.integers(n, 3 * n + 4))
            st = np.minimum(np.arange(Tw) * n // Tw, n - 1)
            comp = rng.integers(0, M, size=Tw)
            segs.append(means[wd, st, comp] + np.sqrt(vars_[wd, st, comp]) * rng.normal(size=(Tw, D)))
        xs.append(np.concatenate(segs))
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, M, D), vars_.reshape(W * n, M, D), w.reshape(W * n, M))
    graph = packed_lattice(wt, n, [list(range(W))] * K)[0]
    lat = hip.Lattices(ctx, [graph])
    assert "layers" in lat.forms()
    b = hip.Batch(ctx, xs)
    b.loglik(gmm, fetch=False)
    U = b.U
    lean = lat.viterbi(b, utt_lattice=np.zeros(U, dtype=np.int32), want_path=True)
    fast = lat.viterbi(b, want_path=True)
    np.testing.assert_array_equal(fast["end_cost_flat"], lean["end_cost_flat"])
    np.testing.assert_array_equal(fast["best_end"], lean["best_end"])
    assert np.isfinite(lean["end_cost_flat"]).any()
    if K * (n - 1) >= 4:
        assert np.isinf(lean["end_cost_flat"]).any()          # the short utterances cannot hold K words
    for u in range(U):
        np.testing.assert_array_equal(fast["paths"][u], lean["paths"][u])
    nopath = lat.viterbi(b, want_path=False)                          # the variant without decision bits
    np.testing.assert_array_equal(nopath["end_cost_flat"], lean["end_cost_flat"])
    np.testing.assert_array_equal(nopath["best_end"], lean["best_end"])
    row_word = np.where(graph["row_state"] >= 0, graph["row_state"] // n, -1).astype(np.int32)
    la = lat.viterbi_labels(b, row_word)
    lb = lat.viterbi_labels(b, row_word, utt_lattice=np.zeros(U, dtype=np.int32))
    for u in range(U):
        np.testing.assert_array_equal(la["labels"][u], lb["labels"][u])
    # and against the oracle's reference-shaped DP on a few utterances (costs 1e-12, paths exact)
    nll = b.loglik(gmm, fetch=True)
    R = len(graph["row_state"])
    dense = np.full((R, R), np.inf)
    dense[graph["arc_to"], graph["arc_from"]] = graph["arc_cost"]
    is_nes = graph["row_state"] < 0
    for u in (0, 13, 40, U - 1):
        E = np.zeros((R, len(xs[u])))
        E[~is_nes] = nll[b.offsets[u]:b.offsets[u + 1]][:, graph["row_state"][~is_nes]].T
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            costs, path = O.decode_states(E, is_nes, dense, end_points=[[int(e), -1] for e in graph["end_rows"]])
        ec = costs[np.asarray(graph["end_rows"]), -1]
        fin = np.isfinite(ec)
        np.testing.assert_array_equal(np.isfinite(fast["end_cost"][u]), fin)
        np.testing.assert_allclose(fast["end_cost"][u][fin], ec[fin], rtol=1e-12)
        if fin.any():
            np.testing.assert_array_equal(fast["paths"][u], path)
    b.close()
    lat.close()
    gmm.close()


def test_layers_kernel_fp32_likelihoods_and_long_utterances(hip, ctx):
    """fp32 resident likelihoods (the DP itself stays fp64), utterances of several hundred frames (many decision
    words, several register chunks in the back-trace), ragged lengths in one launch."""
    from sr.recognition.continuous_speech import packed_lattice
    from sr.recognition.batch import path_to_words
    rng = np.random.default_rng(5)
    W, n, K, M, D = 10, 5, 7, 4, 13
    means = rng.normal(size=(W, n, M, D)) * 2.0
    vars_ = rng.uniform(0.5, 1.5, size=(W, n, M, D))
    w = rng.dirichlet(np.ones(M), size=(W, n))
    trans = word_trans(rng, n)
    xs, truth = [], []
    for u in range(96):
        words = rng.integers(0, W, size=K)
        segs = []
        for wd in words:
            Tw = int(rng.integers(6, 140 if u % 5 == 0 else 40))
            st = np.minimum(np.arange(Tw) * n // Tw, n - 1)
            comp = rng.integers(0, M, size=Tw)
            segs.append(means[wd, st, comp] + np.sqrt(vars_[wd, st, comp]) * rng.normal(size=(Tw, D)))
        xs.append(np.concatenate(segs))
        truth.append([int(v) for v in words])
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, M, D), vars_.reshape(W * n, M, D), w.reshape(W * n, M))
    graph = packed_lattice([trans] * W, n, [list(range(W))] * K)[0]
    lat = hip.Lattices(ctx, [graph])
    for dtype in (np.float32, np.float64):
        b = hip.Batch(ctx, xs, dtype=dtype)
        b.loglik(gmm, fetch=False)
        lean = lat.viterbi(b, utt_lattice=np.zeros(b.U, dtype=np.int32), want_path=True)
        fast = lat.viterbi(b, want_path=True)
        np.testing.assert_array_equal(fast["end_cost_flat"], lean["end_cost_flat"])
        np.testing.assert_array_equal(fast["best_end"], lean["best_end"])
        ok = 0
        for u in range(b.U):
            np.testing.assert_array_equal(fast["paths"][u], lean["paths"][u])
            ok += path_to_words(fast["paths"][u], graph["row_state"], n) == truth[u]
        assert ok >= 0.9 * b.U
        assert max(len(x) for x in xs) > 400
        b.close()
    lat.close()
    gmm.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_loop_kernel_reference_goldens(hip, ctx, dtype):
    """G14: the word-loop graph decoded by the reference's own decode_hmm_states -- end costs, BIT-EXACT paths and
    digits through the loop-form kernel (four utterances per wave, one launch for all utterances of a penalty)."""
    from sr.recognition.continuous_speech import packed_loop_lattice
    g = load_golden("G14_loop_grammar")
    means, vars_, w, wt = g["means"], g["vars"], g["w"], g["word_trans"]
    W, n, M, D = means.shape
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, M, D), vars_.reshape(W * n, M, D), w.reshape(W * n, M))
    U = int(g["n_utts"])
    for pen in (0, 1):
        graph = packed_loop_lattice([wt] * W, n, float(g["p%d_penalty" % pen]))[0]
        lat = hip.Lattices(ctx, [graph])
        b = hip.Batch(ctx, [g["p%d_x%d" % (pen, u)] for u in range(U)], dtype=dtype)
        b.loglik(gmm, fetch=False)
        r = lat.viterbi(b, want_path=True)
        rw = g["p%d_row_word" % pen]
        ends = np.asarray(graph["end_rows"])
        for u in range(U):
            ref = g["p%d_costs%d" % (pen, u)]
            np.testing.assert_allclose(r["end_cost"][u], ref[ends, -1], rtol=1e-10 if dtype == np.float64 else 1e-5)
            np.testing.assert_array_equal(r["paths"][u], g["p%d_path%d" % (pen, u)])
            assert O.path_to_words(r["paths"][u], rw < 0, rw) == list(g["p%d_digits%d" % (pen, u)])
        b.close()
        lat.close()


@pytest.mark.parametrize("W,n,skip,penalty", [(10, 5, False, 0.0), (11, 5, False, 2.5), (1, 2, False, 0.0), (16, 3, True, 1.0),
                                              (5, 8, True, 0.0), (7, 4, False, 0.7), (3, 6, True, 3.0), (2, 7, False, 0.0),
                                              (4, 12, False, 0.5), (3, 12, True, 1.0), (5, 16, False, 0.0), (2, 16, True, 2.0),
                                              # more than 16 words: the wide loop kernel (one utterance per wave, lane = word)
                                              (17, 5, False, 0.5), (40, 3, True, 1.0), (64, 5, False, 0.0), (33, 8, True, 2.0)])
def test_loop_kernel_equals_lean_kernel(hip, ctx, W, n, skip, penalty):
    """Random word models through the loop grammar: ragged utterances (1 .. 6 words, a few too short for even one
    word, counts that are not a multiple of the four utterances a wave holds): end costs BITWISE, chosen ends, paths
    and labels equal to the row-per-lane lean kernel."""
    from sr.recognition.continuous_speech import packed_loop_lattice
    rng = np.random.default_rng(77 * W + n)
    M, D = 2, 6
    means = rng.normal(size=(W, n, M, D)) * 2.0
    vars_ = rng.uniform(0.5, 1.5, size=(W, n, M, D))
    w = rng.dirichlet(np.ones(M), size=(W, n))
    wt = [word_trans(rng, n, skip, last_self=rng.uniform(0.0, 0.3)) for _ in range(W)]
    xs = []
    for u in range(61):
        if u % 9 == 0:
            xs.append(rng.normal(size=(int(rng.integers(2, max(3, n))), D)) * 2.0)      # shorter than any word
            continue
        segs = []
        for wd in rng.integers(0, W, size=rng.integers(1, 7)):
            Tw = int(rng.integers(n, 3 * n + 4))
            st = np.minimum(np.arange(Tw) * n // Tw, n - 1)
            comp = rng.integers(0, M, size=Tw)
            segs.append(means[wd, st, comp] + np.sqrt(vars_[wd, st, comp]) * rng.normal(size=(Tw, D)))
        xs.append(np.concatenate(segs))
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, M, D), vars_.reshape(W * n, M, D), w.reshape(W * n, M))
    graph = packed_loop_lattice(wt, n, penalty)[0]
    lat = hip.Lattices(ctx, [graph])
    assert "loop" in lat.forms()
    for dtype in (np.float64, np.float32):
        b = hip.Batch(ctx, xs, dtype=dtype)
        b.loglik(gmm, fetch=False)
        U = b.U
        lean = lat.viterbi(b, utt_lattice=np.zeros(U, dtype=np.int32), want_path=True)
        fast = lat.viterbi(b, want_path=True)
        np.testing.assert_array_equal(fast["end_cost_flat"], lean["end_cost_flat"])
        np.testing.assert_array_equal(fast["best_end"], lean["best_end"])
        for u in range(U):
            np.testing.assert_array_equal(fast["paths"][u], lean["paths"][u])
        nopath = lat.viterbi(b, want_path=False)
        np.testing.assert_array_equal(nopath["end_cost_flat"], lean["end_cost_flat"])
        np.testing.assert_array_equal(nopath["best_end"], lean["best_end"])
        row_word = np.where(graph["row_state"] >= 0, graph["row_state"] // n, -1).astype(np.int32)
        la = lat.viterbi_labels(b, row_word)
        lb = lat.viterbi_labels(b, row_word, utt_lattice=np.zeros(U, dtype=np.int32))
        lc = lat.viterbi_labels(b, row_word, as_lists=False)                    # packed result (gh_viterbi_labels_packed)
        ld = lat.viterbi_labels(b, row_word, utt_lattice=np.zeros(U, dtype=np.int32), as_lists=False)
        for u in range(U):
            np.testing.assert_array_equal(la["labels"][u], lb["labels"][u])
            for pk in (lc, ld):
                np.testing.assert_array_equal(pk["labels_flat"][pk["label_off"][u]:pk["label_off"][u] + pk["n_labels"][u]], la["labels"][u])
        assert len(lc["labels_flat"]) == sum(len(l) for l in la["labels"])
        b.close()
    lat.close()
    gmm.close()


def test_layers_kernel_is_not_taken_for_other_graphs(hip, ctx):
    """Graphs that are not in layer form (different words per layer, a non-emitting end row, the loop grammar) keep
    working through the other kernels -- and a layer-form batch with a one-frame utterance falls back as a whole
    (the reference's column wrap at T == 1)."""
    from sr.recognition.continuous_speech import packed_lattice, packed_loop_lattice
    rng = np.random.default_rng(9)
    W, n, M, D = 4, 3, 1, 4
    means = rng.normal(size=(W, n, M, D)) * 2.0
    vars_ = np.ones((W, n, M, D))
    w = np.ones((W, n, M))
    wt = [word_trans(rng, n) for _ in range(W)]
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, M, D), vars_.reshape(W * n, M, D), w.reshape(W * n, M))
    xs = [rng.normal(size=(int(T), D)) for T in (9, 1, 14, 30)]
    b = hip.Batch(ctx, xs)
    b.loglik(gmm, fetch=False)
    for graph in (packed_lattice(wt, n, [[0, 1], [2, 3, 1]])[0], packed_loop_lattice(wt, n)[0],
                  packed_lattice(wt, n, [list(range(W))] * 3)[0]):
        lat = hip.Lattices(ctx, [graph])
        a = lat.viterbi(b, want_path=True)
        c = lat.viterbi(b, utt_lattice=np.zeros(b.U, dtype=np.int32), want_path=True)
        np.testing.assert_array_equal(a["end_cost_flat"], c["end_cost_flat"])
        for u in range(b.U):
            np.testing.assert_array_equal(a["paths"][u], c["paths"][u])
        lat.close()
    b.close()
    gmm.close()


def test_label_decode_without_the_end_costs():
    """viterbi_labels(want_end_cost=False): same labels and chosen end rows, nothing but them comes back (packed and
    per-utterance form, layer form and loop form)."""
    from sr.recognition import _hip
    from sr.recognition.continuous_speech import packed_lattice, packed_loop_lattice
    ctx = _hip.default_context()
    rng = np.random.default_rng(21)
    W, n, M, D, K = 5, 4, 2, 6, 3
    means = rng.normal(size=(W * n, M, D)) * 2
    gmm = _hip.PackedGMM(ctx, means, rng.uniform(0.5, 1.5, size=(W * n, M, D)), rng.dirichlet(np.ones(M), size=W * n))
    wt = [word_trans(rng, n) for _ in range(W)]
    xs = [means[rng.integers(0, W * n, size=int(rng.integers(n * K, 5 * n * K))), 0] + rng.normal(size=(1, D)) for _ in range(40)]
    b = _hip.Batch(ctx, xs)
    b.loglik(gmm, fetch=False)
    for graph in (packed_lattice(wt, n, [list(range(W))] * K)[0], packed_loop_lattice(wt, n)[0]):
        lat = _hip.Lattices(ctx, [graph])
        row_word = np.where(graph["row_state"] >= 0, graph["row_state"] // n, -1).astype(np.int32)
        for as_lists in (True, False):
            full = lat.viterbi_labels(b, row_word, as_lists=as_lists)
            lean = lat.viterbi_labels(b, row_word, as_lists=as_lists, want_end_cost=False)
            assert lean["end_cost_flat"] is None and full["end_cost_flat"] is not None
            np.testing.assert_array_equal(lean["best_end"], full["best_end"])
            np.testing.assert_array_equal(lean["n_labels"], full["n_labels"])
            np.testing.assert_array_equal(lean["labels_flat"], full["labels_flat"])
        lat.close()
    b.close(); gmm.close()
