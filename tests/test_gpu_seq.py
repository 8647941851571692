# -*- coding: utf-8 -*-
"""Forced-alignment lattices (continuous_speech.py:80-89: one word per layer, a graph per distinct transcript) through
the sequence-form kernels (gh_seq.hip: four utterances per wave, lane = layer) against the row-per-lane lean kernel
(GMMHMM_VITERBI=lean, read per call), the generic forward-backward and the oracle's reference-shaped DP."""
import os
import warnings

import numpy as np
import pytest

from oracle import ref_numpy as O
from test_gpu_layers import word_trans

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from sr.recognition import _hip
    return _hip


@pytest.fixture(scope="module")
def ctx(hip):
    return hip.default_context()


class forced:
    """GMMHMM_VITERBI / GMMHMM_FB for the duration of a with block."""

    def __init__(self, **env):
        self.env = env

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.env}
        os.environ.update(self.env)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def make_task(rng, W, n, skip, Kmax, U, M=2, D=6, short=8):
    from sr.recognition.continuous_speech import packed_lattice
    means = rng.normal(size=(W, n, M, D)) * 2.0
    vars_ = rng.uniform(0.5, 1.5, size=(W, n, M, D))
    w = rng.dirichlet(np.ones(M), size=(W, n))
    wt = [word_trans(rng, n, skip, last_self=rng.uniform(0.0, 0.3)) for _ in range(W)]
    xs, labels = [], []
    for u in range(U):
        K = int(rng.integers(1, Kmax + 1))
        words = [int(v) for v in rng.integers(0, W, size=K)]
        if u < short:
            T = int(rng.integers(2, max(3, K * (n - 1) + 1)))              # (mostly) too short for the transcript
            xs.append(rng.normal(size=(T, D)) * 2.0)
        else:
            segs = []
            for wd in words:
                Tw = int(rng.integers(n, 3 * n + 4))
                st = np.minimum(np.arange(Tw) * n // Tw, n - 1)
                comp = rng.integers(0, M, size=Tw)
                segs.append(means[wd, st, comp] + np.sqrt(vars_[wd, st, comp]) * rng.normal(size=(Tw, D)))
            xs.append(np.concatenate(segs))
        labels.append(words)
    keys, graphs, utt_graph = {}, [], np.empty(U, dtype=np.int32)
    for u, l in enumerate(labels):
        if tuple(l) not in keys:
            keys[tuple(l)] = len(graphs)
            graphs.append(packed_lattice(wt, n, [[x] for x in l])[0])
        utt_graph[u] = keys[tuple(l)]
    return means, vars_, w, wt, xs, labels, graphs, utt_graph


@pytest.mark.parametrize("W,n,skip,Kmax", [(10, 5, False, 7), (4, 2, False, 16), (6, 3, True, 9), (3, 8, True, 5),
                                           (5, 4, False, 12), (2, 7, True, 3), (7, 6, False, 1),
                                           # 12 and 16 states per word (BASELINE configs[3]'s word models): round 5
                                           (3, 12, False, 5), (2, 12, True, 4), (3, 16, False, 4), (2, 16, True, 6)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sequence_kernel_equals_lean_kernel(hip, ctx, W, n, skip, Kmax, dtype):
    """end costs BITWISE, chosen end and paths equal -- reachable and unreachable utterances, transcripts of 1..Kmax
    words, ragged lengths in one launch; then the oracle on a few utterances."""
    rng = np.random.default_rng(77 * W + 5 * n + Kmax)
    U = 90
    means, vars_, w, wt, xs, labels, graphs, utt_graph = make_task(rng, W, n, skip, Kmax, U)
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, -1, means.shape[-1]), vars_.reshape(W * n, -1, means.shape[-1]),
                        w.reshape(W * n, -1))
    lat = hip.Lattices(ctx, graphs)
    assert "sequence" in lat.forms()
    b = hip.Batch(ctx, xs, dtype=dtype)
    b.loglik(gmm, fetch=False)

    def run(want_path):
        try:
            return lat.viterbi(b, utt_lattice=utt_graph, want_path=want_path), None
        except hip.BackendError as e:
            return None, str(e)

    with forced(GMMHMM_VITERBI="lean"):
        lean, lean_err = run(True)
    fast, fast_err = run(True)
    if lean is None:
        # an unreachable utterance whose back-trace runs into the start row: both routes report the same thing
        assert fast is None and "without predecessor" in lean_err and "without predecessor" in fast_err
        keep = [u for u in range(U) if u >= 8]
        b.close()
        xs = [xs[u] for u in keep]
        utt_graph = utt_graph[keep]
        b = hip.Batch(ctx, xs, dtype=dtype)
        b.loglik(gmm, fetch=False)
        with forced(GMMHMM_VITERBI="lean"):
            lean, lean_err = run(True)
        fast, fast_err = run(True)
        assert lean is not None and fast is not None, (lean_err, fast_err)
    np.testing.assert_array_equal(fast["end_cost_flat"], lean["end_cost_flat"])
    np.testing.assert_array_equal(fast["best_end"], lean["best_end"])
    assert np.isfinite(lean["end_cost_flat"]).any()
    for u in range(b.U):
        np.testing.assert_array_equal(fast["paths"][u], lean["paths"][u])
    nopath, _ = run(False)
    np.testing.assert_array_equal(nopath["end_cost_flat"], lean["end_cost_flat"])
    nll = b.loglik(gmm, fetch=True)
    for u in (0, b.U // 2, b.U - 1):
        graph = graphs[utt_graph[u]]
        R = len(graph["row_state"])
        dense = np.full((R, R), np.inf)
        dense[graph["arc_to"], graph["arc_from"]] = graph["arc_cost"]
        is_nes = graph["row_state"] < 0
        E = np.zeros((R, len(xs[u])))
        E[~is_nes] = nll[b.offsets[u]:b.offsets[u + 1]][:, graph["row_state"][~is_nes]].T
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            costs, path = O.decode_states(E, is_nes, dense, end_points=[[int(e), -1] for e in graph["end_rows"]])
        ec = costs[np.asarray(graph["end_rows"]), -1]
        if np.isfinite(ec).all():
            np.testing.assert_allclose(fast["end_cost"][u], ec, rtol=1e-12 if dtype == np.float64 else 1e-5)
            if dtype == np.float64:
                np.testing.assert_array_equal(fast["paths"][u], path)
    b.close()
    lat.close()
    gmm.close()


def test_sequence_kernel_long_utterances_many_graphs(hip, ctx):
    """Several hundred frames per utterance (many decision words), 300 utterances with ~300 distinct transcripts, the
    shape of `continuous_train`'s alignment step (continuous_speech.py:80-106)."""
    from sr.recognition.continuous_speech import packed_lattice
    rng = np.random.default_rng(11)
    W, n, M, D = 10, 5, 4, 13
    means = rng.normal(size=(W, n, M, D)) * 2.0
    vars_ = rng.uniform(0.5, 1.5, size=(W, n, M, D))
    w = rng.dirichlet(np.ones(M), size=(W, n))
    wt = [word_trans(rng, n) for _ in range(W)]
    xs, graphs = [], []
    for u in range(300):
        K = int(rng.integers(4, 8))
        words = rng.integers(0, W, size=K)
        segs = []
        for wd in words:
            Tw = int(rng.integers(6, 120 if u % 7 == 0 else 40))
            st = np.minimum(np.arange(Tw) * n // Tw, n - 1)
            comp = rng.integers(0, M, size=Tw)
            segs.append(means[wd, st, comp] + np.sqrt(vars_[wd, st, comp]) * rng.normal(size=(Tw, D)))
        xs.append(np.concatenate(segs))
        graphs.append(packed_lattice(wt, n, [[int(x)] for x in words])[0])
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, M, D), vars_.reshape(W * n, M, D), w.reshape(W * n, M))
    lat = hip.Lattices(ctx, graphs)
    assert "sequence" in lat.forms()
    ug = np.arange(300, dtype=np.int32)
    b = hip.Batch(ctx, xs)
    b.loglik(gmm, fetch=False)
    with forced(GMMHMM_VITERBI="lean"):
        lean = lat.viterbi(b, utt_lattice=ug, want_path=True)
    fast = lat.viterbi(b, utt_lattice=ug, want_path=True)
    np.testing.assert_array_equal(fast["end_cost_flat"], lean["end_cost_flat"])
    for u in range(b.U):
        np.testing.assert_array_equal(fast["paths"][u], lean["paths"][u])
    assert max(len(x) for x in xs) > 300
    b.close()
    lat.close()
    gmm.close()


@pytest.mark.parametrize("W,n,skip,Kmax", [(10, 5, False, 7), (4, 2, False, 16), (6, 3, True, 9), (3, 8, True, 5), (7, 6, False, 1),
                                           (3, 8, True, 12), (5, 4, True, 16)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sequence_forward_backward_equals_generic_kernel(hip, ctx, W, n, skip, Kmax, dtype):
    """log P, frame x state occupancies and expected self transitions of the sequence-form kernels -- lane = cell, one
    utterance per wave (graphs of <= 64 cells) and lane = layer, four per wave (the last shape: up to 96 cells; and
    GMMHMM_FBSEQ=layer) -- against the generic row-per-lane forward-backward (GMMHMM_FB=generic): reachable and
    unreachable utterances, repeated words in one transcript (the occupancies of their layers add up), single-frame
    utterances."""
    rng = np.random.default_rng(31 * W + 7 * n + Kmax)
    U = 70
    means, vars_, w, wt, xs, labels, graphs, utt_graph = make_task(rng, W, n, skip, Kmax, U)
    xs[9] = xs[9][:1]                                                  # T == 1
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, -1, means.shape[-1]), vars_.reshape(W * n, -1, means.shape[-1]),
                        w.reshape(W * n, -1))
    lat = hip.Lattices(ctx, graphs)
    assert "sequence" in lat.forms()
    b = hip.Batch(ctx, xs, dtype=dtype)
    b.loglik(gmm, fetch=False)
    with forced(GMMHMM_FB="generic"):
        ref = lat.forward_backward(b, utt_lattice=utt_graph, want_occ=True, want_self_xi=True)
    got = lat.forward_backward(b, utt_lattice=utt_graph, want_occ=True, want_self_xi=True)
    fin = np.isfinite(ref["logp"])
    assert fin.any() and (Kmax < 3 or (~fin).any())
    np.testing.assert_array_equal(np.isfinite(got["logp"]), fin)
    np.testing.assert_allclose(got["logp"][fin], ref["logp"][fin], rtol=1e-12)
    np.testing.assert_allclose(got["occ"], ref["occ"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(got["self_xi"], ref["self_xi"], rtol=1e-9, atol=1e-12)
    # every frame of a reachable utterance is in exactly one state of its transcript... or, on a word boundary, in two:
    # the last state of a word and the first of the next share the boundary frame (decode.py:109-111)
    rows = got["occ"].sum(axis=1)
    for u in np.nonzero(fin)[0][:10]:
        r = rows[b.offsets[u]:b.offsets[u + 1]]
        assert (r > 1 - 1e-9).all() and (r < 2 + 1e-9).all()
    only = lat.forward_backward(b, utt_lattice=utt_graph)            # log P alone: no backward sweep
    np.testing.assert_array_equal(only["logp"], got["logp"])
    # the two lane mappings do the same arithmetic per cell in the same order: same bits (occupancies of a repeated
    # word's layers meet in one LDS cell in either order: last bits there)
    with forced(GMMHMM_FBSEQ="layer"):
        lay = lat.forward_backward(b, utt_lattice=utt_graph, want_occ=True, want_self_xi=True)
    np.testing.assert_array_equal(lay["logp"], got["logp"])
    np.testing.assert_allclose(lay["occ"], got["occ"], rtol=1e-14, atol=1e-300)
    np.testing.assert_allclose(lay["self_xi"], got["self_xi"], rtol=1e-13)
    b.close()
    lat.close()
    gmm.close()


@pytest.mark.parametrize("W,n,skip,Kmax", [(10, 5, False, 7), (6, 3, True, 9), (4, 2, False, 16), (3, 16, True, 4), (2, 12, False, 5)])
def test_align_segments_equals_cut_segments_of_the_paths(hip, ctx, W, n, skip, Kmax):
    """gh_align_segments (alignment + the regrouping loop of continuous_speech.py:90-106 on the device, one int per
    frame back) against the ORACLE's restatement of that loop (O.cut_segments) over the paths of gh_viterbi -- per
    frame the same state, per state the same number of segments."""

    def cut_segments(path, row_state):
        rs = np.asarray(row_state)
        for row, lo, hi in O.cut_segments(path, rs < 0):
            yield int(rs[row]), lo, hi
    rng = np.random.default_rng(5 * W + n + Kmax)
    U = 80
    means, vars_, w, wt, xs, labels, graphs, utt_graph = make_task(rng, W, n, skip, Kmax, U, short=0)
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, -1, means.shape[-1]), vars_.reshape(W * n, -1, means.shape[-1]),
                        w.reshape(W * n, -1))
    lat = hip.Lattices(ctx, graphs)
    b = hip.Batch(ctx, xs)
    b.loglik(gmm, fetch=False)
    paths = lat.viterbi(b, utt_lattice=utt_graph, want_path=True)
    got = lat.align_segments(b, utt_lattice=utt_graph)
    np.testing.assert_array_equal(got["end_cost_flat"], paths["end_cost_flat"])
    want = np.full(b.N, -1, dtype=np.int32)
    want_start = np.zeros(b.N, dtype=bool)
    for u in range(U):
        for sid, lo, hi in cut_segments(paths["paths"][u], graphs[utt_graph[u]]["row_state"]):
            want[b.offsets[u] + lo:b.offsets[u] + hi] = sid
            want_start[b.offsets[u] + lo] = True
    np.testing.assert_array_equal(got["frame_state"], want)
    np.testing.assert_array_equal(got["segment_start"], want_start)
    assert (want >= 0).mean() > 0.5 and want_start.sum() >= U
    # the same alignment as RUNS (gh_align_runs): expanded, they are the frame labels; gathered run by run, the same rows
    runs = lat.align_runs(b, utt_lattice=utt_graph)
    np.testing.assert_array_equal(runs["end_cost_flat"], got["end_cost_flat"])
    exp = np.full(b.N, -1, dtype=np.int32)
    exp_start = np.zeros(b.N, dtype=bool)
    for sid, lo, ln in zip(runs["state"], runs["start"], runs["length"]):
        assert ln > 0 and (exp[lo:lo + ln] == -1).all()
        exp[lo:lo + ln] = sid
        exp_start[lo] = True
    np.testing.assert_array_equal(exp, want)
    np.testing.assert_array_equal(exp_start, want_start)
    assert (np.diff(runs["start"]) > 0).all()                                  # utterance / time order
    order = np.argsort(runs["state"], kind="stable")
    rl = runs["length"][order]
    g_runs = b.gather_runs(runs["start"][order], rl, np.cumsum(rl) - rl, int(rl.sum()))
    rows = np.concatenate([np.arange(lo, lo + ln) for lo, ln in zip(runs["start"][order], rl)])
    g_rows = b.gather(rows)
    Xall = np.concatenate(xs)
    gm = hip.PackedGMM(ctx, np.zeros((1, 1, Xall.shape[1])), np.ones((1, 1, Xall.shape[1])), np.ones((1, 1)))
    np.testing.assert_array_equal(g_runs.loglik(gm), g_rows.loglik(gm))        # (the same rows in the same order)
    np.testing.assert_allclose(g_runs.loglik(gm)[:, 0],
                               0.5 * (Xall[rows] ** 2).sum(axis=1) + 0.5 * Xall.shape[1] * np.log(2 * np.pi), rtol=1e-12)
    gm.close(); g_runs.close(); g_rows.close()
    # the same through the layer-form and the lean kernels (one graph for the whole batch)
    one = hip.Lattices(ctx, [graphs[utt_graph[0]]])
    p1 = one.viterbi(b, want_path=True)
    g1 = one.align_segments(b)
    w1 = np.full(b.N, -1, dtype=np.int32)
    for u in range(U):
        for sid, lo, hi in cut_segments(p1["paths"][u], graphs[utt_graph[0]]["row_state"]):
            w1[b.offsets[u] + lo:b.offsets[u] + hi] = sid
    np.testing.assert_array_equal(g1["frame_state"], w1)
    one.close()
    b.close()
    lat.close()
    gmm.close()


@pytest.mark.parametrize("W,n,skip,Kmax", [(10, 5, False, 7), (6, 3, True, 9), (3, 8, True, 5), (3, 16, True, 4), (4, 12, False, 3)])
def test_transcripts_handle_equals_arc_list_handle(hip, ctx, W, n, skip, Kmax):
    """gh_lattices_create_transcripts (graphs from W word models + label strings, sequence form written directly, the
    row-per-lane arrays expanded lazily) against gh_lattices_create on the arc lists of `packed_lattice`: Viterbi
    (costs, ends, paths), segments, forward-backward (log P, occupancies, xi) identical; and every fallback -- a
    single-frame utterance, GMMHMM_VITERBI=lean, alpha / beta / gamma matrices, a beam -- lands on the same numbers."""
    rng = np.random.default_rng(3 * W + n + 100 * Kmax)
    U = 60
    means, vars_, w, wt, xs, labels, graphs, utt_graph = make_task(rng, W, n, skip, Kmax, U, short=0)
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, -1, means.shape[-1]), vars_.reshape(W * n, -1, means.shape[-1]),
                        w.reshape(W * n, -1))
    transcripts = [None] * len(graphs)
    for u, l in enumerate(labels):
        transcripts[utt_graph[u]] = l
    a = hip.Lattices(ctx, graphs)
    t = hip.Lattices.from_transcripts(ctx, wt, n, transcripts)
    assert t.forms() == {"sequence"} and "sequence" in a.forms()
    assert list(t.R) == list(a.R) and list(t.n_end) == list(a.n_end)

    def same(b, beam=None, paths=True):
        a.set_beam(beam), t.set_beam(beam)
        ra = a.viterbi(b, utt_lattice=utt_graph, want_path=paths)
        rt = t.viterbi(b, utt_lattice=utt_graph, want_path=paths)
        np.testing.assert_array_equal(rt["end_cost_flat"], ra["end_cost_flat"])
        np.testing.assert_array_equal(rt["best_end"], ra["best_end"])
        for u in range(b.U if paths else 0):
            np.testing.assert_array_equal(rt["paths"][u], ra["paths"][u])
        a.set_beam(None), t.set_beam(None)

    b = hip.Batch(ctx, xs)
    b.loglik(gmm, fetch=False)
    same(b)
    np.testing.assert_array_equal(t.align_segments(b, utt_lattice=utt_graph)["frame_state"],
                                  a.align_segments(b, utt_lattice=utt_graph)["frame_state"])
    # label mode (main.py:59-67 on the device) through the sequence-form kernels: the transcript comes back
    row_word = [np.where(g["row_state"] >= 0, g["row_state"] // n, -1).astype(np.int32) for g in graphs]
    la = a.viterbi_labels(b, row_word, utt_lattice=utt_graph, max_labels=Kmax + 1)
    lt = t.viterbi_labels(b, row_word, utt_lattice=utt_graph, max_labels=Kmax + 1)
    for u in range(b.U):
        np.testing.assert_array_equal(lt["labels"][u], la["labels"][u])
        assert [int(v) for v in lt["labels"][u]] == list(transcripts[utt_graph[u]])
    fa = a.forward_backward(b, utt_lattice=utt_graph, want_occ=True, want_self_xi=True)
    ft = t.forward_backward(b, utt_lattice=utt_graph, want_occ=True, want_self_xi=True)
    np.testing.assert_array_equal(ft["logp"], fa["logp"])
    np.testing.assert_array_equal(ft["occ"], fa["occ"])
    np.testing.assert_allclose(ft["self_xi"], fa["self_xi"], rtol=1e-12)     # (atomics: summation order)
    with forced(GMMHMM_VITERBI="lean"):
        same(b)
    same(b, beam=6)
    ma = a.forward_backward(b, utt_lattice=utt_graph, want_matrices=True)
    mt = t.forward_backward(b, utt_lattice=utt_graph, want_matrices=True)
    for u in (0, U - 1):
        np.testing.assert_array_equal(mt["gamma"][u], ma["gamma"][u])
    b.close()
    xs1 = list(xs)
    xs1[5] = xs1[5][:1]                                                # a single-frame utterance: the reference's column wrap
    utt1 = utt_graph.copy()
    k1 = [g for g in range(len(transcripts)) if len(transcripts[g]) == 1]
    if k1:
        utt1[5] = k1[0]
    b = hip.Batch(ctx, xs1)
    b.loglik(gmm, fetch=False)
    ra = a.viterbi(b, utt_lattice=utt1, want_path=False)
    rt = t.viterbi(b, utt_lattice=utt1, want_path=False)
    np.testing.assert_array_equal(rt["end_cost_flat"], ra["end_cost_flat"])
    b.close()
    a.close()
    t.close()
    gmm.close()


def test_transcripts_handle_isolated_words_and_odd_models(hip, ctx):
    """Transcripts of one word each take the ordinary handle (one-word chain forms); word models with an arc the
    sequence form cannot hold (a backward arc) are expanded eagerly -- same results as the arc-list handle."""
    from sr.recognition.continuous_speech import packed_lattice
    rng = np.random.default_rng(2)
    W, n, M, D = 4, 3, 2, 5
    means = rng.normal(size=(W, n, M, D)) * 2.0
    vars_ = rng.uniform(0.5, 1.5, size=(W, n, M, D))
    w = rng.dirichlet(np.ones(M), size=(W, n))
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, M, D), vars_.reshape(W * n, M, D), w.reshape(W * n, M))
    wt = [word_trans(rng, n) for _ in range(W)]
    xs = [rng.normal(size=(int(rng.integers(4, 30)), D)) for _ in range(12)]
    b = hip.Batch(ctx, xs)
    b.loglik(gmm, fetch=False)
    single = hip.Lattices.from_transcripts(ctx, wt, n, [[i] for i in range(W)])
    assert "fb_chain" in single.forms()
    ug = (np.arange(12) % W).astype(np.int32)
    ref = hip.Lattices(ctx, [packed_lattice(wt, n, [[i]])[0] for i in range(W)])
    np.testing.assert_array_equal(single.viterbi(b, utt_lattice=ug)["end_cost_flat"], ref.viterbi(b, utt_lattice=ug)["end_cost_flat"])
    np.testing.assert_array_equal(single.forward_backward(b, utt_lattice=ug)["logp"], ref.forward_backward(b, utt_lattice=ug)["logp"])
    odd = [m.copy() for m in wt]
    odd[1][0, 2] = 1.7                                                 # state 2 -> state 0: not left-to-right
    seqs = [[0, 1], [1, 2, 3], [2, 1]]
    o = hip.Lattices.from_transcripts(ctx, odd, n, seqs)
    r = hip.Lattices(ctx, [packed_lattice(odd, n, [[l] for l in s])[0] for s in seqs])
    ug = (np.arange(12) % 3).astype(np.int32)
    ro, rr = o.viterbi(b, utt_lattice=ug), r.viterbi(b, utt_lattice=ug)
    np.testing.assert_array_equal(ro["end_cost_flat"], rr["end_cost_flat"])
    for u in range(12):
        np.testing.assert_array_equal(ro["paths"][u], rr["paths"][u])
    with pytest.raises(hip.BackendError):
        hip.Lattices.from_transcripts(ctx, wt, n, [[0], []])
    for h in (single, ref, o, r):
        h.close()
    b.close()
    gmm.close()


def test_sequence_forward_backward_fp32_likelihoods(hip, ctx):
    """fp32 resident likelihoods (the recursion itself is fp64 either way): log P AND the occupancies / expected self
    transitions against the fp64 batch within the north star's fp32 tolerance (1e-3)."""
    rng = np.random.default_rng(8)
    W, n, U = 8, 5, 60
    means, vars_, w, wt, xs, labels, graphs, utt_graph = make_task(rng, W, n, False, 6, U, short=0)
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, -1, means.shape[-1]), vars_.reshape(W * n, -1, means.shape[-1]),
                        w.reshape(W * n, -1))
    lat = hip.Lattices(ctx, graphs)
    out = {}
    for dt in (np.float64, np.float32):
        b = hip.Batch(ctx, xs, dtype=dt)
        b.loglik(gmm, fetch=False)
        out[dt] = lat.forward_backward(b, utt_lattice=utt_graph, want_occ=True, want_self_xi=True)
        b.close()
    a, f = out[np.float64], out[np.float32]
    np.testing.assert_allclose(f["logp"], a["logp"], rtol=1e-3)
    np.testing.assert_allclose(f["occ"], a["occ"], atol=1e-3)
    np.testing.assert_allclose(f["self_xi"], a["self_xi"], rtol=1e-3, atol=1e-3)
    lat.close()
    gmm.close()


@pytest.mark.parametrize("W,n,M,D,Kmax", [(10, 5, 8, 39, 7), (4, 2, 4, 13, 12), (6, 3, 8, 5, 6), (3, 8, 2, 20, 4), (5, 4, 16, 13, 5)])
def test_bw_statistics_over_sequence_segments_equal_generic(hip, ctx, W, n, M, D, Kmax):
    """Multi-word transcripts: after the sequence-form forward-backward gh_bw_accumulate runs the fused matrix-core
    kernel over (utterance, layer) segments grouped by word, gamma read from the occupancy matrix -- against the generic
    statistics kernel on the same occupancies (GMMHMM_BW=generic).  Repeated words in a transcript, ragged lengths,
    unreachable utterances, words that never occur; M = 16 is a shape the fused kernel hands back."""
    rng = np.random.default_rng(W + 10 * n + M)
    U = 50
    means, vars_, w, wt, xs, labels, graphs, utt_graph = make_task(rng, W, n, False, Kmax, U, M=M, D=D, short=4)
    labels = [[l % (W - 1) for l in ls] for ls in labels] if W > 2 else labels      # the last word never occurs
    transcripts, keys = [], {}
    for u, l in enumerate(labels):
        if tuple(l) not in keys:
            keys[tuple(l)] = len(transcripts)
            transcripts.append(l)
        utt_graph[u] = keys[tuple(l)]
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, M, D), vars_.reshape(W * n, M, D), w.reshape(W * n, M))
    lat = hip.Lattices.from_transcripts(ctx, wt, n, transcripts)
    b = hip.Batch(ctx, xs)
    b.loglik(gmm, fetch=False)
    r = lat.forward_backward(b, utt_lattice=utt_graph, want_occ=True, fetch_occ=True)
    fused = b.bw_accumulate(gmm)
    with forced(GMMHMM_BW="generic"):
        generic = b.bw_accumulate(gmm)
    scale = np.maximum(np.abs(generic).max(axis=(1, 2), keepdims=True), 1e-300)
    assert np.max(np.abs(fused - generic) / scale) < 1e-9
    reach = np.isfinite(r["logp"])
    np.testing.assert_allclose(fused[:, :, 0].sum(), r["occ"].sum(), rtol=1e-9)
    assert fused[:, :, 0].sum() >= sum(len(xs[u]) for u in range(U) if reach[u]) * (1 - 1e-9)
    if W > 2:
        assert np.all(fused[(W - 1) * n:] == 0)
    b.close()
    lat.close()
    gmm.close()


def test_sequence_forward_backward_many_states_and_state_base(hip, ctx):
    """More than 256 states (the occupancy rows no longer fit the LDS row buffers: double atomics on the matrix) and a
    transcripts handle with explicit first-state indices per word (`state_base`), against the generic kernels."""
    from sr.recognition.continuous_speech import packed_lattice
    rng = np.random.default_rng(21)
    W, n, M, D, U = 60, 5, 1, 4, 24
    base = (np.arange(W)[::-1] * n).astype(np.int32)                    # word w owns states base[w] .. base[w] + n - 1
    means = rng.normal(size=(W * n, M, D)) * 2.0
    vars_ = rng.uniform(0.5, 1.5, size=(W * n, M, D))
    wgt = np.ones((W * n, M))
    wt = [word_trans(rng, n) for _ in range(W)]
    seqs = [[int(v) for v in rng.integers(0, W, size=int(rng.integers(1, 9)))] for _ in range(U)]
    xs = []
    for l in seqs:
        segs = []
        for wd in l:
            Tw = int(rng.integers(n, 3 * n))
            st = np.minimum(np.arange(Tw) * n // Tw, n - 1)
            segs.append(means[base[wd] + st, 0] + rng.normal(size=(Tw, D)))
        xs.append(np.concatenate(segs))
    gmm = hip.PackedGMM(ctx, means, vars_, wgt)
    t = hip.Lattices.from_transcripts(ctx, wt, n, seqs, state_base=base)
    a = hip.Lattices(ctx, [packed_lattice(wt, n, [[l] for l in s], state_base=base)[0] for s in seqs])
    assert t.forms() == {"sequence"}
    ug = np.arange(U, dtype=np.int32)
    b = hip.Batch(ctx, xs)
    b.loglik(gmm, fetch=False)
    rt, ra = t.viterbi(b, utt_lattice=ug), a.viterbi(b, utt_lattice=ug)
    np.testing.assert_array_equal(rt["end_cost_flat"], ra["end_cost_flat"])
    for u in range(U):
        np.testing.assert_array_equal(rt["paths"][u], ra["paths"][u])
    with forced(GMMHMM_FB="generic"):
        ref = a.forward_backward(b, utt_lattice=ug, want_occ=True, want_self_xi=True)
    got = t.forward_backward(b, utt_lattice=ug, want_occ=True, want_self_xi=True)
    np.testing.assert_allclose(got["logp"], ref["logp"], rtol=1e-12)
    np.testing.assert_allclose(got["occ"], ref["occ"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(got["self_xi"], ref["self_xi"], rtol=1e-9, atol=1e-12)
    fused = b.bw_accumulate(gmm)
    with forced(GMMHMM_BW="generic"):
        generic = b.bw_accumulate(gmm)
    scale = np.maximum(np.abs(generic).max(axis=(1, 2), keepdims=True), 1e-300)
    assert np.max(np.abs(fused - generic) / scale) < 1e-9
    for h in (a, t):
        h.close()
    b.close()
    gmm.close()


def test_new_entry_points_edge_cases(hip, ctx):
    """Error paths and edge cases of the round-2 entry points: labels out of range, an empty label string, rows out of
    range in gh_batch_gather, empty gathers, align_segments with single-frame and unreachable utterances (the expanded
    twin takes over: same answers as the arc-list handle), forms() of an expanded handle."""
    from sr.recognition.continuous_speech import packed_lattice
    rng = np.random.default_rng(4)
    W, n, M, D = 3, 3, 2, 4
    wt = [word_trans(rng, n) for _ in range(W)]
    with pytest.raises(hip.BackendError):
        hip.Lattices.from_transcripts(ctx, wt, n, [[0, 3]])
    with pytest.raises(hip.BackendError):
        hip.Lattices.from_transcripts(ctx, wt, n, [[0, -1]])
    with pytest.raises(hip.BackendError):
        hip.Lattices.from_transcripts(ctx, wt, n, [[]])
    means = rng.normal(size=(W * n, M, D))
    gmm = hip.PackedGMM(ctx, means, np.ones_like(means), np.full((W * n, M), 0.5))
    xs = [rng.normal(size=(T, D)) for T in (1, 2, 9, 30, 5)]
    seqs = [[0], [1, 2], [2, 0], [0, 1, 2, 1], [1, 1, 1]]           # utterance 1 is too short for its two words (4 states)
    b = hip.Batch(ctx, xs)
    b.loglik(gmm, fetch=False)
    t = hip.Lattices.from_transcripts(ctx, wt, n, seqs)
    a = hip.Lattices(ctx, [packed_lattice(wt, n, [[l] for l in s])[0] for s in seqs])
    ug = np.arange(5, dtype=np.int32)
    try:
        ra = a.align_segments(b, utt_lattice=ug)
    except hip.BackendError as e:          # an unreachable utterance: the back-trace runs into the start row
        with pytest.raises(hip.BackendError):
            t.align_segments(b, utt_lattice=ug)
        assert "predecessor" in str(e)
    else:
        rt = t.align_segments(b, utt_lattice=ug)
        np.testing.assert_array_equal(rt["frame_state"], ra["frame_state"])
        np.testing.assert_array_equal(rt["segment_start"], ra["segment_start"])
    keep = [0, 2, 3, 4]                                               # without the unreachable one: T == 1 is still in
    b2 = b.gather(np.concatenate([np.arange(b.offsets[u], b.offsets[u + 1]) for u in keep]),
                  offsets=np.concatenate([[0], np.cumsum([len(xs[u]) for u in keep])]))
    b2.loglik(gmm, fetch=False)
    ug2 = np.array(keep, dtype=np.int32)
    rt, ra = t.align_segments(b2, utt_lattice=ug2), a.align_segments(b2, utt_lattice=ug2)
    np.testing.assert_array_equal(rt["frame_state"], ra["frame_state"])
    np.testing.assert_array_equal(rt["end_cost_flat"], ra["end_cost_flat"])
    assert rt["frame_state"][0] == -1                                 # the single frame of a T == 1 utterance joins nothing
    ft, fa = t.forward_backward(b2, utt_lattice=ug2, want_occ=True), a.forward_backward(b2, utt_lattice=ug2, want_occ=True)
    np.testing.assert_array_equal(ft["logp"], fa["logp"])
    np.testing.assert_array_equal(ft["occ"], fa["occ"])
    with pytest.raises(hip.BackendError):
        b.gather([0, b.N])
    e = b.gather(np.zeros(0, dtype=np.int64))
    assert e.N == 0 and e.U == 1
    e.close()
    for h in (t, a):
        h.close()
    b.close(); b2.close()
    gmm.close()


def test_loglik_state_sets(hip, ctx):
    """gh_loglik_sets: several state ranges per utterance (the words of a transcript).  Requested entries are BITWISE those
    of the full matrix, for fp64 and fp32, mixtures of 8 / 1 / 32 components, ranges that share a Gaussian tile, a range
    list that covers everything, and an empty list (nothing requested for that utterance)."""
    from sr.recognition.continuous_speech import transcript_state_sets
    rng = np.random.default_rng(12)
    for W, n, M, D in ((10, 5, 8, 39), (12, 3, 1, 13), (6, 4, 32, 7)):
        S = W * n
        gmm = hip.PackedGMM(ctx, rng.normal(size=(S, M, D)), rng.uniform(0.5, 1.5, size=(S, M, D)), rng.dirichlet(np.ones(M), size=S))
        xs = [rng.normal(size=(int(rng.integers(1, 90)), D)) for _ in range(40)]
        seqs = [[int(v) for v in rng.integers(0, W, size=int(rng.integers(1, 8)))] for _ in xs]
        seqs[3] = list(range(W))
        sets = transcript_state_sets(seqs, n, W)
        for dt in (np.float64, np.float32):
            b = hip.Batch(ctx, xs, dtype=dt)
            full = b.loglik(gmm, fetch=True).copy()
            b2 = hip.Batch(ctx, xs, dtype=dt)
            sub = b2.loglik(gmm, fetch=True, state_sets=sets)
            for u, l in enumerate(seqs):
                cols = np.concatenate([np.arange(w * n, (w + 1) * n) for w in sorted(set(l))])
                np.testing.assert_array_equal(sub[b.offsets[u]:b.offsets[u + 1]][:, cols], full[b.offsets[u]:b.offsets[u + 1]][:, cols])
            b.close(); b2.close()
        gmm.close()
    off, lo, hi = transcript_state_sets([[2, 3, 7, 3], [], [0]], 5, 10)
    assert off.tolist() == [0, 2, 3, 4] and lo.tolist() == [10, 35, 0, 0] and hi.tolist() == [20, 40, 50, 5]
