# -*- coding: utf-8 -*-
"""The multi-launch ("chunked") form of every dynamic program: the DP scratch of one launch -- back-pointers / decision
words of gh_viterbi*, alpha columns of gh_forward_backward -- is bounded by a budget (a quarter of the free HBM, at most
24 GiB; gh_scratch_budget), and a batch that needs more runs as several launches.  At the sizes of BASELINE's configs
one launch is enough on a 288 GB part, so nothing else in this suite (and nothing in bench.py) reaches the second
chunk: GMMHMM_SCRATCH_BUDGET forces >= 3 chunks through every kernel family here, and every result must equal the
one-launch run BIT FOR BIT ("what's weak" 1)."""
import os

import numpy as np
import pytest

from test_gpu_layers import word_trans
from test_gpu_seq import forced, make_task

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from sr.recognition import _hip
    return _hip


@pytest.fixture(scope="module")
def ctx(hip):
    return hip.default_context()


def _word_lattice_task(hip, ctx, rng, W, n, K, U, loop):
    from sr.recognition.continuous_speech import packed_lattice, packed_loop_lattice
    M, D = 2, 6
    means = rng.normal(size=(W, n, M, D)) * 2.0
    vars_ = rng.uniform(0.5, 1.5, size=(W, n, M, D))
    w = rng.dirichlet(np.ones(M), size=(W, n))
    wt = [word_trans(rng, n) for _ in range(W)]
    xs = []
    for u in range(U):
        words = rng.integers(0, W, size=K)
        segs = []
        for wd in words:
            Tw = int(rng.integers(n, 3 * n + 4))
            st = np.minimum(np.arange(Tw) * n // Tw, n - 1)
            comp = rng.integers(0, M, size=Tw)
            segs.append(means[wd, st, comp] + np.sqrt(vars_[wd, st, comp]) * rng.normal(size=(Tw, D)))
        xs.append(np.concatenate(segs))
    graph = (packed_loop_lattice(wt, n) if loop else packed_lattice(wt, n, [list(range(W))] * K))[0]
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, M, D), vars_.reshape(W * n, M, D), w.reshape(W * n, M))
    b = hip.Batch(ctx, xs)
    b.loglik(gmm, fetch=False)
    lat = hip.Lattices(ctx, [graph])
    row_word = np.where(graph["row_state"] >= 0, graph["row_state"] // n, -1).astype(np.int32)
    return gmm, b, lat, row_word


def _same_decode(a, b):
    np.testing.assert_array_equal(a["end_cost_flat"], b["end_cost_flat"])
    np.testing.assert_array_equal(a["best_end"], b["best_end"])
    if "paths" in a:
        assert len(a["paths"]) == len(b["paths"])
        for x, y in zip(a["paths"], b["paths"]):
            np.testing.assert_array_equal(x, y)
    if "labels_flat" in a:
        np.testing.assert_array_equal(a["n_labels"], b["n_labels"])
        for u in range(len(a["n_labels"])):
            np.testing.assert_array_equal(a["labels_flat"][a["label_off"][u]:a["label_off"][u] + a["n_labels"][u]],
                                          b["labels_flat"][b["label_off"][u]:b["label_off"][u] + b["n_labels"][u]])


@pytest.mark.parametrize("loop", [False, True])
@pytest.mark.parametrize("kernel", ["form", "lean", "generic"])
def test_word_lattice_decodes_in_three_or_more_chunks(hip, ctx, loop, kernel):
    """K-layer lattice / word-loop grammar through the layer-form / loop-form kernel, the row-per-lane lean kernel and the
    generic kernel: paths and on-device label sequences of the chunked run == the one-launch run."""
    rng = np.random.default_rng(11 + loop)
    gmm, b, lat, row_word = _word_lattice_task(hip, ctx, rng, W=6, n=4, K=4, U=48, loop=loop)
    env = {} if kernel == "form" else {"GMMHMM_VITERBI": kernel}
    assert ("loop" if loop else "layers") in lat.forms()
    with forced(**env):
        one_p = lat.viterbi(b, want_path=True)
        assert ctx.last_chunks == 1
        one_l = lat.viterbi_labels(b, row_word, as_lists=False)
        # (decision words: 128 B per column for the 4-layer lattice, 16 B for the loop grammar; row-per-lane kernels:
        #  2 B per lattice row and column)
        with forced(GMMHMM_SCRATCH_BUDGET=("2K" if loop else "24K") if kernel == "form" else ("8K" if loop else "120K")):
            many_p = lat.viterbi(b, want_path=True)
            assert ctx.last_chunks >= 3, ctx.last_chunks
            many_l = lat.viterbi_labels(b, row_word, as_lists=False)
            assert ctx.last_chunks >= 3, ctx.last_chunks
    _same_decode(one_p, many_p)
    _same_decode(one_l, many_l)
    b.close(); lat.close(); gmm.close()


@pytest.mark.parametrize("kernel", ["sequence", "lean"])
def test_forced_alignment_in_three_or_more_chunks(hip, ctx, kernel):
    """Sequence-form forced alignment (one graph per transcript), its lean fallback, and the alignment + regrouping call
    of continuous_train."""
    rng = np.random.default_rng(5)
    W, n = 6, 4
    means, vars_, w, wt, xs, labels, graphs, utt_graph = make_task(rng, W, n, False, 6, 60, short=0)
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, -1, means.shape[-1]), vars_.reshape(W * n, -1, means.shape[-1]), w.reshape(W * n, -1))
    b = hip.Batch(ctx, xs)
    b.loglik(gmm, fetch=False)
    lat = hip.Lattices(ctx, graphs)
    env = {} if kernel == "sequence" else {"GMMHMM_VITERBI": "lean"}
    with forced(**env):
        one = lat.viterbi(b, utt_lattice=utt_graph, want_path=True)
        assert ctx.last_chunks == 1
        seg_one = lat.align_segments(b, utt_lattice=utt_graph)
        with forced(GMMHMM_SCRATCH_BUDGET="2K" if kernel == "sequence" else "12K"):
            many = lat.viterbi(b, utt_lattice=utt_graph, want_path=True)
            assert ctx.last_chunks >= 3, ctx.last_chunks
            seg_many = lat.align_segments(b, utt_lattice=utt_graph)
            assert ctx.last_chunks >= 3
    _same_decode(one, many)
    np.testing.assert_array_equal(seg_one["frame_state"], seg_many["frame_state"])
    np.testing.assert_array_equal(seg_one["segment_start"], seg_many["segment_start"])
    b.close(); lat.close(); gmm.close()


def test_chain_kernel_paths_in_three_or_more_chunks(hip, ctx):
    """Isolated-word decode through the stacked chains (configs[1]'s graph) with back-pointers."""
    import bench
    wl = bench.synth_workload(3, 300)
    W, n, M, D = wl["W"], wl["n"], wl["M"], wl["D"]
    gmm = hip.PackedGMM(ctx, wl["means"].reshape(W * n, M, D), wl["vars"].reshape(W * n, M, D), wl["w"].reshape(W * n, M))
    b = hip.Batch(ctx, feats=wl["X"], offsets=wl["off"])
    b.loglik(gmm, fetch=False)
    lat = hip.Lattices(ctx, [bench.stacked_graph(W, n, wl["trans"])])
    assert "chain" in lat.forms()
    one = lat.viterbi(b, want_path=True)
    assert ctx.last_chunks == 1
    with forced(GMMHMM_SCRATCH_BUDGET="1M"):
        many = lat.viterbi(b, want_path=True)
        assert ctx.last_chunks >= 3, ctx.last_chunks
    _same_decode(one, many)
    b.close(); lat.close(); gmm.close()


@pytest.mark.parametrize("kernel", ["sequence", "generic"])
def test_forward_backward_in_three_or_more_chunks(hip, ctx, kernel):
    """log P, occupancies and expected self transitions of the sequence-form and the generic forward-backward; the
    per-state sums of the self transitions are accumulated per launch, so they agree to rounding, everything per
    utterance / per frame bit for bit."""
    rng = np.random.default_rng(9)
    W, n = 6, 4
    means, vars_, w, wt, xs, labels, graphs, utt_graph = make_task(rng, W, n, False, 6, 60, short=0)
    gmm = hip.PackedGMM(ctx, means.reshape(W * n, -1, means.shape[-1]), vars_.reshape(W * n, -1, means.shape[-1]), w.reshape(W * n, -1))
    b = hip.Batch(ctx, xs)
    b.loglik(gmm, fetch=False)
    lat = hip.Lattices(ctx, graphs)
    env = {} if kernel == "sequence" else {"GMMHMM_FB": "generic"}
    with forced(**env):
        one = lat.forward_backward(b, utt_lattice=utt_graph, want_occ=True, want_self_xi=True)
        assert ctx.last_chunks == 1
        with forced(GMMHMM_SCRATCH_BUDGET="64K"):
            many = lat.forward_backward(b, utt_lattice=utt_graph, want_occ=True, want_self_xi=True)
            assert ctx.last_chunks >= 3, ctx.last_chunks
    np.testing.assert_array_equal(one["logp"], many["logp"])
    np.testing.assert_array_equal(one["occ"], many["occ"])
    np.testing.assert_allclose(one["self_xi"], many["self_xi"], rtol=1e-12, atol=1e-14)
    b.close(); lat.close(); gmm.close()


def test_budget_defaults_to_a_share_of_free_memory(hip, ctx):
    """Without the override a small batch is one launch, and a nonsensical override does not break the call."""
    rng = np.random.default_rng(1)
    gmm, b, lat, row_word = _word_lattice_task(hip, ctx, rng, W=4, n=3, K=3, U=10, loop=False)
    os.environ.pop("GMMHMM_SCRATCH_BUDGET", None)
    one = lat.viterbi(b, want_path=True)
    assert ctx.last_chunks == 1
    with forced(GMMHMM_SCRATCH_BUDGET="1"):          # less than one utterance needs: one utterance per launch
        many = lat.viterbi(b, want_path=True)
        assert ctx.last_chunks == b.U
    _same_decode(one, many)
    b.close(); lat.close(); gmm.close()
