# -*- coding: utf-8 -*-
"""configs[4] at its OWN model and at size: continuous-digit decode over the flat 10-word grammar with the 8-mixture,
39-dimensional model (10 words x 5 states), 20 000 seven-word utterances (800 distinct ones tiled x25 on the device, as
bench.py's C5 legs do) -- the reference's `main.py:35,59-67` path: `build_state_sequences(models, [[0..9]] * 7)`,
`decode_hmm_states`, path post-processing.

What is pinned here (parity hole 1):
  * the label sequences of `gh_viterbi_labels` / `gh_viterbi_labels_packed` (layer-form kernels, decision bits, per-lane
    back-trace in label mode) == `path_to_words` of the row-per-lane lean kernel's paths on EVERY utterance, for the
    K = 7 lattice and for the word-loop grammar;
  * sampled utterances against the CPU oracle's `decode_states` (costs 1e-12 relative, paths bit-exact);
  * a decode from fp32 likelihoods == the decode from fp64 likelihoods (paths and labels);
  * the x25 tiling changes nothing (copy k of an utterance decodes like the original), chunked launches included.
"""
import os
import warnings

import numpy as np
import pytest

from oracle import ref_numpy as O

pytestmark = pytest.mark.gpu

K, W, n, M, D = 7, 10, 5, 8, 39
U_BASE, REPS = 800, 25


@pytest.fixture(scope="module")
def problem():
    import bench
    from sr.recognition import _hip
    ctx = _hip.default_context()
    rng = np.random.default_rng(1005)
    wl = bench.synth_workload(1005, 1, W=W, n=n, M=M, D=D)
    means, vars_, trans = wl["means"], wl["vars"], wl["trans"]
    S = W * n
    words = rng.integers(0, W, size=(U_BASE, K))
    Tw = rng.integers(30, 61, size=(U_BASE, K))
    seg_len = Tw.reshape(-1)
    seg_off = np.concatenate([[0], np.cumsum(seg_len)])
    Nb = int(seg_off[-1])
    seg = np.repeat(np.arange(len(seg_len)), seg_len)
    t = np.arange(Nb) - seg_off[seg]
    st = np.minimum(t * n // seg_len[seg], n - 1)
    idx = (words.reshape(-1)[seg] * n + st) * M + rng.integers(0, M, size=Nb)
    X = means.reshape(-1, D)[idx] + np.sqrt(vars_).reshape(-1, D)[idx] * rng.standard_normal((Nb, D))
    off = np.concatenate([[0], np.cumsum(Tw.sum(axis=1))]).astype(np.int64)
    gmm = _hip.PackedGMM(ctx, means.reshape(S, M, D), vars_.reshape(S, M, D), wl["w"].reshape(S, M))
    base = _hip.Batch(ctx, feats=X, offsets=off)
    base32 = _hip.Batch(ctx, feats=X.astype(np.float32), offsets=off, dtype=np.float32)
    return dict(ctx=ctx, hip=_hip, gmm=gmm, base=base, base32=base32, X=X, off=off, words=words, trans=trans,
                means=means.reshape(S, M, D), vars=vars_.reshape(S, M, D), w=wl["w"].reshape(S, M))


def graphs(trans):
    from sr.recognition.continuous_speech import packed_lattice, packed_loop_lattice
    return {"layers": (packed_lattice([trans] * W, n, [list(range(W))] * K)[0], K + 1),
            "loop": (packed_loop_lattice([trans] * W, n)[0], None)}


@pytest.mark.parametrize("grammar", ["layers", "loop"])
def test_c5_labels_equal_the_lean_kernels_paths_on_every_utterance(problem, grammar, monkeypatch):
    from sr.recognition.batch import path_to_words
    hip, ctx, base, gmm = problem["hip"], problem["ctx"], problem["base"], problem["gmm"]
    graph, max_labels = graphs(problem["trans"])[grammar]
    lat = hip.Lattices(ctx, [graph])
    assert ("layers" if grammar == "layers" else "loop") in lat.forms()
    row_word = np.where(graph["row_state"] >= 0, graph["row_state"] // n, -1).astype(np.int32)
    b = base.tile(REPS)
    assert b.U == U_BASE * REPS == 20000
    b.loglik(gmm, fetch=False)
    ml = max_labels if max_labels is not None else b.lengths // (n - 1) + 2
    r = lat.viterbi_labels(b, row_word, max_labels=ml, as_lists=False)          # gh_viterbi_labels_packed
    lf, lo, ln = r["labels_flat"], r["label_off"], r["n_labels"]
    got = [lf[lo[u]:lo[u] + ln[u]].tolist() for u in range(b.U)]
    # the same batch through the row-per-lane lean kernel (another kernel, uint16 back-pointers, host post-processing)
    monkeypatch.setenv("GMMHMM_VITERBI", "lean")
    ref = lat.viterbi(b, want_path=True)
    monkeypatch.delenv("GMMHMM_VITERBI")
    ref_words = [path_to_words(p, graph["row_state"], n) for p in ref["paths"]]
    assert got == ref_words
    np.testing.assert_array_equal(r["best_end"], ref["best_end"])
    np.testing.assert_allclose(r["end_cost_flat"], ref["end_cost_flat"], rtol=1e-12)
    # the x25 tiling: copy k of utterance u decodes like utterance u
    for k in (1, 7, REPS - 1):
        assert got[k * U_BASE:(k + 1) * U_BASE] == got[:U_BASE]
    # the unpacked label call and several chunks (a small scratch budget) give the same labels
    base.loglik(gmm, fetch=False)
    r2 = lat.viterbi_labels(base, row_word, max_labels=(max_labels if max_labels is not None else base.lengths // (n - 1) + 2))
    assert [list(map(int, l)) for l in r2["labels"]] == got[:U_BASE]
    monkeypatch.setenv("GMMHMM_SCRATCH_BUDGET", "8M")
    r3 = lat.viterbi_labels(b, row_word, max_labels=ml, as_lists=False)
    monkeypatch.delenv("GMMHMM_SCRATCH_BUDGET")
    assert ctx.last_chunks >= 3
    np.testing.assert_array_equal(r3["n_labels"], ln)
    np.testing.assert_array_equal(r3["labels_flat"][:int(ln.sum())], lf[:int(ln.sum())])
    # accuracy on the synthetic truth (the K-layer lattice decodes exactly K words)
    truth = [list(map(int, w)) for w in problem["words"]]
    acc = np.mean([got[u] == truth[u] for u in range(U_BASE)])
    assert acc > 0.97, acc
    b.close()
    lat.close()


@pytest.mark.parametrize("grammar", ["layers", "loop"])
def test_c5_sampled_utterances_against_the_oracle(problem, grammar):
    """Eight utterances of the configs[4] model through the oracle's decode_states (the restatement of decode.py:80-146
    pinned by G3 / G4 / G14): end costs 1e-12, paths bit-exact, words == main.py:59-67 of the oracle's path."""
    hip, ctx, base, gmm = problem["hip"], problem["ctx"], problem["base"], problem["gmm"]
    graph, _ = graphs(problem["trans"])[grammar]
    lat = hip.Lattices(ctx, [graph])
    base.loglik(gmm, fetch=False)
    r = lat.viterbi(base, want_path=True)
    R = len(graph["row_state"])
    dense = np.full((R, R), np.inf)
    dense[graph["arc_to"], graph["arc_from"]] = graph["arc_cost"]
    is_nes = graph["row_state"] < 0
    off = problem["off"]
    for u in (0, 1, 17, 100, 333, 512, 640, U_BASE - 1):
        x = problem["X"][off[u]:off[u + 1]]
        nll = O.gmm_neg_loglik_batch(x, problem["means"], problem["vars"], problem["w"])
        E = np.zeros((R, len(x)))
        E[~is_nes] = nll[:, graph["row_state"][~is_nes]].T
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            costs, path = O.decode_states(E, is_nes, dense, end_points=[[int(e), -1] for e in graph["end_rows"]])
        ends = costs[np.asarray(graph["end_rows"]), -1]
        got_ends = r["end_cost_flat"][r["end_off"][u]:r["end_off"][u + 1]]
        fin = np.isfinite(ends)
        np.testing.assert_array_equal(np.isfinite(got_ends), fin)
        np.testing.assert_allclose(got_ends[fin], ends[fin], rtol=1e-12)
        np.testing.assert_array_equal(r["paths"][u], path)
    lat.close()


@pytest.mark.parametrize("grammar", ["layers", "loop"])
def test_c5_decode_from_fp32_likelihoods_equals_the_fp64_decode(problem, grammar):
    """The 'fast mode' of the docs -- fp32 likelihoods, fp64 dynamic program -- on configs[4]'s own model: every state path
    and every label sequence of the 800 distinct utterances equals the fp64 decode's (bench.py reports the same rate on
    5 000 utterances per run)."""
    hip, ctx, gmm = problem["hip"], problem["ctx"], problem["gmm"]
    graph, max_labels = graphs(problem["trans"])[grammar]
    lat = hip.Lattices(ctx, [graph])
    row_word = np.where(graph["row_state"] >= 0, graph["row_state"] // n, -1).astype(np.int32)
    res = {}
    for name in ("base", "base32"):
        b = problem[name]
        b.loglik(gmm, fetch=False)
        ml = max_labels if max_labels is not None else b.lengths // (n - 1) + 2
        res[name] = (lat.viterbi(b, want_path=True)["paths"], lat.viterbi_labels(b, row_word, max_labels=ml)["labels"])
    for pa, pb in zip(res["base"][0], res["base32"][0]):
        np.testing.assert_array_equal(pa, pb)
    for la, lb in zip(res["base"][1], res["base32"][1]):
        np.testing.assert_array_equal(la, lb)
    lat.close()
