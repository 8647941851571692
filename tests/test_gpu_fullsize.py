# -*- coding: utf-8 -*-
"""Parity at BASELINE.json's full size (configs[1]: 10 words x 5 states, 8-mix, 39-dim, 10 000 utterances, ~1 M
frames), where the per-cell Python oracle would take hours: sampled rows against the oracle plus the
size-independent properties the domain offers -- batching independence (bitwise), weight-scaling linearity,
agreement of independent kernels (chain vs lean Viterbi) on every utterance, decode accuracy on data drawn
from the models, fp32 against fp64 within the north star's 1e-3."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import ref_numpy as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full():
    import bench
    from sr.recognition import _hip
    ctx = _hip.default_context()
    wl = bench.synth_workload(1002, 10000)
    W, n, M, D = wl["W"], wl["n"], wl["M"], wl["D"]
    S = W * n
    gmm = _hip.PackedGMM(ctx, wl["means"].reshape(S, M, D), wl["vars"].reshape(S, M, D), wl["w"].reshape(S, M))
    batch = _hip.Batch(ctx, feats=wl["X"], offsets=wl["off"])
    nll = batch.loglik(gmm, fetch=True)
    yield dict(hip=_hip, ctx=ctx, wl=wl, gmm=gmm, batch=batch, nll=nll, S=S, graph=bench.stacked_graph(W, n, wl["trans"]))
    batch.close()
    gmm.close()


def test_fullsize_loglik_sampled_rows_vs_oracle(full):
    wl, nll, S = full["wl"], full["nll"], full["S"]
    W, n, M, D = wl["W"], wl["n"], wl["M"], wl["D"]
    assert nll.shape == (wl["X"].shape[0], S) and np.all(np.isfinite(nll))
    rng = np.random.default_rng(3)
    rows = np.concatenate([[0, 31, 32, nll.shape[0] - 1], rng.integers(0, nll.shape[0], size=44)])
    mean, var, w = wl["means"].reshape(S, M, D), wl["vars"].reshape(S, M, D), wl["w"].reshape(S, M)
    for r in rows:
        ref = np.array([O.gmm_evaluate(wl["X"][r], mean[s], var[s], w[s]) for s in range(S)])
        np.testing.assert_allclose(nll[r], ref, rtol=1e-10)


def test_fullsize_loglik_batching_independence_and_weight_linearity(full):
    hip, ctx, wl, nll, S = full["hip"], full["ctx"], full["wl"], full["nll"], full["S"]
    W, n, M, D = wl["W"], wl["n"], wl["M"], wl["D"]
    # a frame's likelihoods do not depend on where it sits in the batch: second half alone, bitwise
    N = nll.shape[0]
    h = N // 2 + 7           # not a multiple of the kernel's 32-frame block
    b2 = hip.Batch(ctx, feats=wl["X"][h:], offsets=np.array([0, N - h], dtype=np.int64))
    np.testing.assert_array_equal(b2.loglik(full["gmm"]), nll[h:])
    b2.close()
    # scaling every weight of a state by c moves its likelihood by -log c (weights need not sum to 1: A3)
    c = np.linspace(0.25, 4.0, S)
    g2 = hip.PackedGMM(ctx, wl["means"].reshape(S, M, D), wl["vars"].reshape(S, M, D), wl["w"].reshape(S, M) * c[:, None])
    bs = hip.Batch(ctx, feats=wl["X"][:100000], offsets=np.array([0, 100000], dtype=np.int64))
    np.testing.assert_allclose(bs.loglik(g2), nll[:100000] - np.log(c)[None, :], rtol=1e-12, atol=1e-11)
    bs.close()
    g2.close()


def test_fullsize_viterbi_kernels_agree_and_decode(full):
    hip, ctx, wl, batch = full["hip"], full["ctx"], full["wl"], full["batch"]
    W, U = wl["W"], batch.U
    lat = hip.Lattices(ctx, [full["graph"]])
    r_chain = lat.viterbi(batch, want_path=False)                                         # chain kernel
    r_lean = lat.viterbi(batch, utt_lattice=np.zeros(U, dtype=np.int32), want_path=False)  # lean kernel, same graph
    np.testing.assert_array_equal(r_chain["end_cost_flat"], r_lean["end_cost_flat"])
    ec = r_chain["end_cost_flat"].reshape(U, W)
    decoded = np.argmin(ec, axis=1)
    assert np.mean(decoded == wl["words"]) == 1.0
    # decode + post-processing on the device (gh_viterbi_labels) through the chain kernel: one label = the word
    row_word = (np.arange(len(full["graph"]["row_state"])) // wl["n"]).astype(np.int32)
    rl = lat.viterbi_labels(batch, row_word)
    np.testing.assert_array_equal(rl["best_end"], r_chain["best_end"])
    assert all(len(l) == 1 for l in rl["labels"])
    np.testing.assert_array_equal(np.array([l[0] for l in rl["labels"]]), decoded)
    # a sample of utterances through the oracle's reference-shaped DP: costs and paths
    nll = full["nll"]
    sub = [0, 1, U // 2, U - 1]
    sb = hip.Batch(ctx, [wl["X"][wl["off"][u]:wl["off"][u + 1]] for u in sub])
    sb.loglik(full["gmm"], fetch=False)
    rp = lat.viterbi(sb, want_path=True)
    g = full["graph"]
    R = len(g["row_state"])
    dense = np.full((R, R), np.inf)
    dense[g["arc_to"], g["arc_from"]] = g["arc_cost"]
    n = wl["n"]
    for k, u in enumerate(sub):
        E = nll[wl["off"][u]:wl["off"][u + 1]].T                                  # [S, T] == rows of the stacked graph
        wd = int(decoded[u])
        sl = slice(wd * n, (wd + 1) * n)
        costs, path = O.decode_states(E[sl], np.zeros(n, dtype=bool), dense[sl, sl])
        np.testing.assert_allclose(ec[u, wd], costs[-1, -1], rtol=1e-12)
        got = rp["paths"][k]
        np.testing.assert_array_equal(got[:, 0] - wd * n, path[:, 0])
        np.testing.assert_array_equal(got[:, 1], path[:, 1])
    sb.close()
    lat.close()


def test_fullsize_fp32_against_fp64(full):
    hip, ctx, wl, nll = full["hip"], full["ctx"], full["wl"], full["nll"]
    b32 = hip.Batch(ctx, feats=wl["X"], offsets=wl["off"], dtype=np.float32)
    n32 = b32.loglik(full["gmm"], fetch=True)
    np.testing.assert_allclose(n32, nll, rtol=1e-3)
    lat = hip.Lattices(ctx, [full["graph"]])
    ec = lat.viterbi(b32, want_path=False)["end_cost_flat"].reshape(b32.U, wl["W"])
    assert np.mean(np.argmin(ec, axis=1) == wl["words"]) == 1.0
    b32.close()
    lat.close()


def test_fullsize_forced_alignment_and_forward_backward_properties():
    """configs[2]'s model on 2000 seven-word transcripts (1.4 M frames, one graph per transcript) -- the training shape of
    `continuous_train` / the soft-EM trainer -- through the sequence-form kernels, checked by properties that need no
    per-cell oracle:
      * Viterbi: costs, ends and paths identical to the row-per-lane lean kernel on every utterance (bitwise);
      * alignment + regrouping: per utterance the states appear in transcript order, every state's frames are one run
        per visit, the frames a run drops are exactly the reference's (one per state entry inside a word, none at a word
        boundary: continuous_speech.py:96-106);
      * forward-backward: log P >= -(best path cost) (the sum over paths contains the best one), occupancies of a frame
        add up to 1 (2 on a word boundary: decode.py:109-111), expected self transitions of a state < its occupancy;
      * sampled utterances against the oracle's reference-shaped DP (costs 1e-12, paths exact)."""
    import warnings
    import bench
    from sr.recognition import _hip
    from sr.recognition.continuous_speech import packed_lattice
    ctx = _hip.default_context()
    U, K = 2000, 7
    wl = bench.synth_workload(1003, U * K)
    W, n, M, D = wl["W"], wl["n"], wl["M"], wl["D"]
    off = wl["off"][::K]
    labels = [[int(w) for w in wl["words"][i * K:(i + 1) * K]] for i in range(U)]
    gmm = _hip.PackedGMM(ctx, wl["means"].reshape(W * n, M, D), wl["vars"].reshape(W * n, M, D), wl["w"].reshape(W * n, M))
    b = _hip.Batch(ctx, feats=wl["X"], offsets=off)
    nll = b.loglik(gmm, fetch=True)
    keys, ug = {}, np.empty(U, dtype=np.int32)
    for u, l in enumerate(labels):
        ug[u] = keys.setdefault(tuple(l), len(keys))
    lat = _hip.Lattices.from_transcripts(ctx, [wl["trans"]] * W, n, list(keys))
    assert lat.forms() == {"sequence"}
    fast = lat.viterbi(b, utt_lattice=ug, want_path=True)
    os.environ["GMMHMM_VITERBI"] = "lean"
    try:
        lean = lat.viterbi(b, utt_lattice=ug, want_path=True)
    finally:
        os.environ.pop("GMMHMM_VITERBI")
    np.testing.assert_array_equal(fast["end_cost_flat"], lean["end_cost_flat"])
    assert np.isfinite(fast["end_cost_flat"]).all()
    for u in range(U):
        np.testing.assert_array_equal(fast["paths"][u], lean["paths"][u])
    seg = lat.align_segments(b, utt_lattice=ug)
    fs, st = seg["frame_state"], seg["segment_start"]
    for u in range(0, U, 7):
        f = fs[off[u]:off[u + 1]]
        used = f[f >= 0]
        runs = used[np.insert(np.diff(used) != 0, 0, True)]                   # the states in the order they are visited
        want = [l * n + s for l in labels[u] for s in range(n)]
        # (the last state's run is never closed, and a state visited for a single frame inside a word leaves no frame)
        it = iter(want)
        assert all(any(r == w for w in it) for r in runs), (u, runs, want)
        assert st[off[u]:off[u + 1]].sum() == len(runs)
    fb = lat.forward_backward(b, utt_lattice=ug, want_occ=True, want_self_xi=True)
    assert np.all(fb["logp"] >= -fast["end_cost_flat"] - 1e-9 * np.abs(fast["end_cost_flat"]))
    rows = fb["occ"].sum(axis=1)
    assert rows.min() > 1 - 1e-9 and rows.max() < 2 + 1e-9
    assert np.all(fb["self_xi"] < fb["occ"].sum(axis=0) + 1e-9)
    is_cache = {}
    for u in (0, 777, U - 1):
        g = packed_lattice([wl["trans"]] * W, n, [[l] for l in labels[u]])[0]
        R = len(g["row_state"])
        dense = np.full((R, R), np.inf)
        dense[g["arc_to"], g["arc_from"]] = g["arc_cost"]
        is_nes = g["row_state"] < 0
        E = np.zeros((R, off[u + 1] - off[u]))
        E[~is_nes] = nll[off[u]:off[u + 1]][:, g["row_state"][~is_nes]].T
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            costs, path = O.decode_states(E, is_nes, dense, end_points=[[int(e), -1] for e in g["end_rows"]])
        np.testing.assert_allclose(fast["end_cost"][u], costs[np.asarray(g["end_rows"]), -1], rtol=1e-12)
        np.testing.assert_array_equal(fast["paths"][u], path)
    b.close()
    lat.close()
    gmm.close()
