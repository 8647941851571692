# -*- coding: utf-8 -*-
"""Host-side logic of the `sr.recognition` mirror that needs no GPU: segment
bookkeeping, lattice construction (object-level and packed), alignment cutting,
decode post-processing, state packing, sharding and the M-step -- checked against
the reference goldens and the oracle."""
import warnings

import numpy as np
import pytest

from conftest import load_golden
from oracle import ref_numpy as O


@pytest.fixture(scope="module")
def R(built_library):
    import sr.recognition as R
    return R


def make_hmm(R, means, vars_, w, trans):
    h = R.HMM(means.shape[0])
    h.gmm_states = []
    for s in range(means.shape[0]):
        g = R.GMM(means[s, 0].copy(), vars_[s, 0].copy(), means.shape[1])
        g.update_models(means[s].copy(), vars_[s].copy(), w[s].copy())
        h.gmm_states.append(g)
    h.transitions = trans.copy()
    return h


def test_exported_names(R):
    import sr
    for name in ("dtw", "decode_hmm_states", "HMM", "GMM", "NES", "HMMState", "MultivariateNormal", "mahalanobis",
                 "kmeans", "skmeans", "align_gmm_states", "calc_transition_costs", "get_segments_from_path",
                 "segment_data", "combine_templates", "calc_variance", "cluster_centroids", "build_state_sequences",
                 "continuous_train"):
        assert hasattr(R, name), name
    for name in ("HMMState", "HMM", "decode_hmm_states", "GMM", "build_state_sequences", "NES"):
        assert hasattr(sr, name), name  # reference sr/__init__.py:2


def test_segment_helpers_match_golden_and_oracle(R):
    g = load_golden("G9_hmm_fit_single")
    np.testing.assert_array_equal(R.get_segments_from_path(g["gsp_path"], 5), g["gsp_out"])
    np.testing.assert_allclose(R.calc_transition_costs(2, g["ctc_lens"]), g["ctc_out"], rtol=0)
    s5 = load_golden("G5_dtw")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.testing.assert_array_equal(R.calc_transition_costs(2, s5["seg_lens_skip"]), s5["trans_skip"])
    ys = [g["y%d" % i] for i in range(int(g["n"]))]
    starts = np.array([[0, 5, 11, 18, 25]] * len(ys))
    ref = O.segment_data(ys, 5, starts)
    got = R.segment_data(ys, len(ys), 5, starts)
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(a, b)
    mu, var = R.combine_templates(ys, len(ys), 5, starts)
    rmu, rvar = O.combine_templates(ys, 5, starts)
    np.testing.assert_array_equal(mu, rmu)
    np.testing.assert_array_equal(var, rvar)
    np.testing.assert_array_equal(R.calc_variance(ys[0].T), np.cov(ys[0].T).diagonal())
    np.testing.assert_allclose(R.calc_variance(ys[0].T), np.var(ys[0], axis=0, ddof=1), rtol=1e-12)
    cl = np.array([0, 1, 1, 0, 2, 2, 2])
    np.testing.assert_array_equal(R.cluster_centroids(ys[0][:7], cl, 3), O.cluster_centroids(ys[0][:7], cl, 3))


@pytest.mark.parametrize("K", [1, 2, 3, 7])
def test_build_state_sequences_matches_reference(R, K):
    g = load_golden("G4_lattice_decode")
    W, n = g["means"].shape[:2]
    hmms = [make_hmm(R, g["means"][i], g["vars"][i], g["w"][i], g["word_trans"]) for i in range(W)]
    p = "K%d_" % K
    seq, trans, ends = R.build_state_sequences(hmms, [list(range(W))] * K)
    assert len(seq) == int(g[p + "R"]) == 1 + K * (W * n + 1)
    assert list(ends) == list(g[p + "ends"])
    ref = np.full_like(trans, np.inf)
    ref[g[p + "arc_to"], g[p + "arc_from"]] = g[p + "arc_cost"]
    np.testing.assert_array_equal(trans, ref)
    for r, s in enumerate(seq):
        w, st = g[p + "row_word"][r], g[p + "row_state"][r]
        assert (type(s) is R.NES) == (w < 0)
        if w >= 0:
            assert s is hmms[w].gmm_states[st]  # shared objects, not copies
    # the packed builder describes the same graph
    from sr.recognition.continuous_speech import packed_lattice
    from sr.recognition import _pack
    pk, nes_rows = packed_lattice([h.transitions for h in hmms], n, [list(range(W))] * K)
    dense = np.full_like(trans, np.inf)
    dense[pk["arc_to"], pk["arc_from"]] = pk["arc_cost"]
    np.testing.assert_array_equal(dense, ref)
    np.testing.assert_array_equal(pk["end_rows"], g[p + "ends"])
    np.testing.assert_array_equal(pk["row_state"], np.where(g[p + "row_word"] < 0, -1,
                                                            g[p + "row_word"] * n + g[p + "row_state"]))
    row_state, uniq = _pack.pack_states(seq)
    assert len(uniq) == W * n
    np.testing.assert_array_equal(row_state, pk["row_state"])
    g2 = _pack.graph_from_dense(row_state, trans, [0], ends)
    assert sorted(zip(g2["arc_to"], g2["arc_from"])) == sorted(zip(pk["arc_to"], pk["arc_from"]))


def test_path_postprocessing_matches_reference_main(R):
    from sr.recognition.batch import path_to_words
    g = load_golden("G4_lattice_decode")
    n = g["means"].shape[1]
    for K in (1, 2, 3, 7):
        p = "K%d_" % K
        rw, rs = g[p + "row_word"], g[p + "row_state"]
        row_state = np.where(rw < 0, -1, rw * n + rs)
        assert path_to_words(g[p + "path"], row_state, n) == list(g[p + "digits"])
    assert path_to_words(np.array([]), [0], n) == []


def test_sequence_report_is_the_tally_of_reference_main(R):
    """main.py:54-84: sequence accuracy = exact string matches / utterances; digit accuracy = 1 - differing positions of
    the WRONG strings / all label digits."""
    from sr.recognition.batch import sequence_report
    labels = [[1, 2, 3], [4, 5, 6], [7, 8, 9], [0, 0, 0]]
    decoded = [[1, 2, 3], [4, 9, 6], [9, 8, 7], [0, 0, 0]]
    r = sequence_report(decoded, labels)
    # the reference's own arithmetic
    correct = digit_ndiff = n_digits = 0
    for m, l in zip(decoded, labels):
        n_digits += len(l)
        if m == l:
            correct += 1
        else:
            digit_ndiff += np.count_nonzero(np.asarray(m) - np.asarray(l))
    assert r["sequence_accuracy"] == correct / len(labels) == 0.5
    assert r["digit_accuracy"] == (n_digits - digit_ndiff) / n_digits == 9 / 12
    assert r["n_digit_errors"] == 3 and r["n_digits"] == 12 and r["n_correct"] == 2
    # strings of different length (only the loop grammar can produce them): the surplus positions count as errors
    assert sequence_report([[1, 2]], [[1, 2, 3]])["n_digit_errors"] == 1
    assert sequence_report([], [])["sequence_accuracy"] == 0


def test_transcript_helpers_match_the_packed_lattice(R):
    """`transcript_row_state` / `transcript_state_sets` (what the transcripts handle and gh_loglik_sets are fed with)
    against the rows and states of the graph `packed_lattice` builds for the same label string."""
    from sr.recognition.continuous_speech import packed_lattice, transcript_row_state, transcript_state_sets
    rng = np.random.default_rng(0)
    n, W = 4, 9
    wt = [np.where(np.tri(n, n, 0, dtype=bool) & ~np.tri(n, n, -3, dtype=bool), 1.0, np.inf).T for _ in range(W)]
    seqs = [[int(v) for v in rng.integers(0, W, size=int(rng.integers(1, 9)))] for _ in range(20)]
    off, lo, hi = transcript_state_sets(seqs, n, W)
    for u, l in enumerate(seqs):
        g = packed_lattice(wt, n, [[w] for w in l])[0]
        np.testing.assert_array_equal(transcript_row_state(l, n), g["row_state"])
        want = sorted(set(int(s) for s in g["row_state"] if s >= 0))
        got = sorted(s for r in range(off[u], off[u + 1]) for s in range(lo[r], hi[r]))
        assert got == want
        assert all(hi[r] < lo[r + 1] for r in range(off[u], off[u + 1] - 1))        # disjoint, ascending, merged
    assert transcript_state_sets([[]], n, W)[1:] == (0, n * W) or list(transcript_state_sets([[]], n, W)[2]) == [n * W]


def test_cut_segments_matches_reference_rule(R):
    """The frame ranges cut from the reference's own alignment path, against the oracle's
    restatement of continuous_speech.py:90-106."""
    from sr.recognition.continuous_speech import cut_segments, packed_lattice
    g = load_golden("G4_lattice_decode")
    W, n = g["means"].shape[:2]
    labels = list(g["forced_labels"])
    pk, _ = packed_lattice([g["word_trans"]] * W, n, [[l] for l in labels])
    got = list(cut_segments(g["forced_path"], pk["row_state"]))
    # reference rule: the oracle's restatement of the loop
    rw, rs, nes, _, _ = O.build_state_sequences(n, [g["word_trans"]] * W, [[l] for l in labels])
    exp = [(int(rw[cur]) * n + int(rs[cur]), start, c) for cur, start, c in O.cut_segments(g["forced_path"], nes)]
    assert got == exp and len(got) >= 2 * n
    # consequences the docstring promises: entering frame dropped inside a word, final run open
    assert all(b > a for _, a, b in got)
    assert got[-1][0] != labels[-1] * n + n - 1 or got[-1][2] < len(g["forced_x"])


def test_pack_states_pads_mixed_mixture_sizes(R):
    from sr.recognition import _pack
    a = R.GMM(np.zeros(3), np.ones(3), 2)
    b = R.GMM(np.ones(3), np.full(3, 2.0), 4)
    row_state, uniq = _pack.pack_states([R.NES(), a, b, a, R.NES()])
    np.testing.assert_array_equal(row_state, [-1, 0, 1, 0, -1])
    means, vars_, w = _pack.stack_gmms(uniq)
    assert means.shape == (2, 4, 3)
    np.testing.assert_array_equal(w[0], [0.5, 0.5, 0, 0])
    np.testing.assert_array_equal(vars_[0, 2:], np.ones((2, 3)))
    c = R.GMM(np.zeros(5), np.ones(5), 2)
    with pytest.raises(NameError):
        _pack.stack_gmms([a, c])


def test_gmm_object_semantics_without_gpu(R):
    g = R.GMM(np.arange(3.0), np.ones(3), 4)
    assert len(g) == 4 and g.w.tolist() == [0.25] * 4 and g.mu_old.shape == (4, 3)
    assert sorted(g.__dict__) == ["dists", "id", "mu_old", "n_gaussians", "parent", "sigma_old", "w", "w_old"]
    assert sorted(g.dists[0].__dict__) == ["_cov", "inv_cov", "mean"]
    g.update_models(np.ones((2, 3)), np.full((2, 3), 2.0), np.array([0.7, 0.1]))
    assert g.w.tolist() == [0.7, 0.1, 0.25, 0.25]  # not renormalised
    np.testing.assert_array_equal(g.dists[1].inv_cov, np.diag([0.5] * 3))
    with pytest.raises(np.linalg.LinAlgError):
        g.update_models(np.ones((1, 3)), np.zeros((1, 3)), np.array([1.0]))
    # a zero variance in the SECOND component: the first is updated, the second keeps its inverse but has the new mean and
    # the bad covariance -- the state the reference's component-after-component assignment leaves (hmm_state.py:29-30)
    bad = np.array([[4.0, 4.0, 4.0], [1.0, 0.0, 1.0], [8.0, 8.0, 8.0]])
    with pytest.raises(np.linalg.LinAlgError):
        g.update_models(np.arange(9.0).reshape(3, 3), bad, np.array([0.2, 0.3, 0.5]))
    np.testing.assert_array_equal(g.dists[0].inv_cov, np.diag([0.25] * 3))
    np.testing.assert_array_equal(g.dists[1].mean, [3.0, 4.0, 5.0])
    np.testing.assert_array_equal(g.dists[1].cov, bad[1])
    np.testing.assert_array_equal(g.dists[1].inv_cov, np.diag([0.5] * 3))
    np.testing.assert_array_equal(g.dists[2].mean, np.arange(3.0))          # never reached
    assert all(sorted(d.__dict__) == ["_cov", "inv_cov", "mean"] for d in g.dists)
    import copy
    h = copy.deepcopy(g)
    assert h.id == g.id and hash(h) == hash(g)
    a = R.HMM(5)
    with pytest.raises(AssertionError):
        a.use_gmm = False
        a[0]


def test_shard_utterances_balances_frames(R):
    from sr.recognition.parallel import shard_utterances
    rng = np.random.default_rng(0)
    lens = rng.integers(50, 151, size=1000)
    shards = shard_utterances(lens, 8)
    assert sorted(np.concatenate(shards).tolist()) == list(range(1000))
    loads = np.array([lens[s].sum() for s in shards])
    assert loads.max() - loads.min() <= lens.max()
    assert [len(s) for s in shard_utterances([5, 4], 4)] == [1, 1, 0, 0]


def test_m_step_from_centred_statistics_matches_reference_em(R):
    """One reference EM iteration (oracle) == m_step on centred sufficient statistics."""
    from sr.recognition.parallel import m_step
    g = load_golden("G7_gmm_em")
    data, k = g["data"], 3
    means, vars_, w = g["init_means"].copy(), g["init_vars"].copy(), g["init_w"].copy()
    p = np.array([O.gmm_evaluate(x, means, vars_, w, neg_log=False)[:k] for x in data])
    r = p / p.sum(axis=1, keepdims=True)
    D = data.shape[1]
    stats = np.zeros((1, k, 1 + 2 * D))
    for c in range(k):
        d = data - means[c]
        stats[0, c, 0] = r[:, c].sum()
        stats[0, c, 1:1 + D] = (r[:, [c]] * d).sum(axis=0)
        stats[0, c, 1 + D:] = (r[:, [c]] * d * d).sum(axis=0)
    mu, sigma, wn = m_step(stats, [len(data)], means[None, :k])
    st = O.new_gmm_state(g["mu0"], g["var0"], len(w))
    st["means"][:], st["vars"][:], st["w"][:] = means, vars_, w
    O.gmm_em(data, st["means"], st["vars"], st["w"], k, max_iteration=1,
             old=(st["mu_old"], st["sigma_old"], st["w_old"]))
    np.testing.assert_allclose(mu[0], st["means"][:k], rtol=1e-12)
    np.testing.assert_allclose(sigma[0], st["vars"][:k], rtol=1e-11)
    np.testing.assert_allclose(wn[0], st["w"][:k], rtol=1e-12)


def test_reference_pickles_load_into_the_mirror(R):
    """N2: models pickled by the reference's own classes (module path sr.recognition.*, attribute
    set of SURVEY.md section 5) unpickle as the mirror's classes with every parameter intact."""
    import pickle
    g = load_golden("G12_reference_pickle")
    hmms = pickle.loads(g["pickle"].tobytes())
    assert len(hmms) == 2 and all(type(h) is R.HMM for h in hmms)
    for h, ids in zip(hmms, g["ids"]):
        assert [type(s) for s in h.gmm_states] == [R.GMM] * 3
        assert [str(s.id) for s in h.gmm_states] == list(ids)
        assert sorted(h.__dict__) == ["gmm_states", "mu", "n_segments", "segments", "sigma", "transitions",
                                      "use_em", "use_gmm"]
        d = h.gmm_states[0].dists[0]
        assert type(d) is R.MultivariateNormal and sorted(d.__dict__) == ["_cov", "inv_cov", "mean"]
        np.testing.assert_allclose(np.diag(d.inv_cov), 1.0 / d.cov, rtol=1e-14)
    # and they round-trip through the mirror's own pickling
    again = pickle.loads(pickle.dumps(hmms))
    assert again[0] == hmms[0] and again[1].gmm_states[2].id == hmms[1].gmm_states[2].id


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


def test_deepcopy_of_models_equals_the_generic_deepcopy():
    """MultivariateNormal / HMMState.__deepcopy__ copy their arrays directly (continuous_train copies every model once
    per outer iteration): same values, same ids and hashes, fresh arrays, aliasing between objects preserved, parent
    cycles survive -- as with the generic copy.deepcopy the reference relies on (continuous_speech.py:62,101)."""
    import copy
    import pickle
    from sr.recognition.hmm_state import GMM, NES, MultivariateNormal
    rng = np.random.default_rng(0)
    mu, sigma = rng.normal(size=5), rng.uniform(0.5, 2, size=5)
    g = GMM(mu, sigma, 4)                      # all four components share the SAME mean / cov arrays (hmm_state.py:110)
    g.update_models(rng.normal(size=(2, 5)), rng.uniform(0.5, 2, size=(2, 5)), np.array([0.3, 0.7]))
    g.parent = [g]                             # a cycle through a non-array attribute
    nes = NES()
    models = [g, nes, g]
    c = copy.deepcopy(models)
    assert c[0] is c[2] and c[0] is not g and c[0].parent[0] is c[0]
    assert c[0].id == g.id and hash(c[0]) == hash(g) and c[1].id == nes.id and c[1] == nes
    assert c[0] == g
    for a, b in zip(c[0].dists, g.dists):
        assert a is not b and a.mean is not b.mean
        np.testing.assert_array_equal(a.mean, b.mean)
        np.testing.assert_array_equal(a.cov, b.cov)
        np.testing.assert_array_equal(a.inv_cov, b.inv_cov)
    assert c[0].dists[2].mean is c[0].dists[3].mean            # untouched components still share one array
    assert c[0].dists[0].mean is not c[0].dists[1].mean
    assert set(vars(c[0])) == set(vars(g)) and set(vars(c[0].dists[0])) == set(vars(g.dists[0]))
    c[0].dists[0].mean[0] = 99.0
    assert g.dists[0].mean[0] != 99.0
    r = pickle.loads(pickle.dumps(c[0]))
    assert r == g or r.dists[0].mean[0] == 99.0
    m = MultivariateNormal(mu, np.diag(sigma))
    mc = copy.deepcopy(m)
    np.testing.assert_array_equal(mc.inv_cov, m.inv_cov)
    assert mc.inv_cov is not m.inv_cov


def test_segment_order_regroups_like_segment_data():
    """kmeans.segment_order / gather_rows / split_segments (train_words: the frames of all words regrouped by (word, segment)
    with one sort and one gather) against segment_data, the reference's slices (kmeans.py:33-50), word by word -- including
    empty segments, which stay `np.array([])`."""
    import importlib
    km = importlib.import_module("sr.recognition.kmeans")
    rng = np.random.default_rng(0)
    W, n = 3, 5
    tw = [[rng.normal(size=(int(rng.integers(5, 30)), 4)) for _ in range(int(rng.integers(1, 6)))] for _ in range(W)]
    n_temps = np.array([len(t) for t in tw])
    lengths = np.array([len(t) for ts in tw for t in ts])
    starts = np.array([np.sort(np.concatenate([[0], rng.integers(0, L, n - 1)])) for L in lengths])
    starts[0, 2] = starts[0, 1]                                   # an empty segment
    X = np.concatenate([t for ts in tw for t in ts])
    order, counts = km.segment_order(lengths, n_temps, starts, n)
    segs = km.split_segments(km.gather_rows(X, order), counts)
    off = np.concatenate([[0], np.cumsum(n_temps)])
    for w in range(W):
        ref = km.segment_data(tw[w], len(tw[w]), n, starts[off[w]:off[w + 1]])
        assert len(ref) == len(segs[w]) == n
        for a, b in zip(ref, segs[w]):
            assert a.shape == b.shape
            np.testing.assert_array_equal(a, b)
    big = rng.normal(size=(40000, 39))
    idx = rng.permutation(40000)
    np.testing.assert_array_equal(km.gather_rows(big, idx), big[idx])      # (the threaded copy)


def test_segment_order_by_brute_force():
    """segment_order builds the order from the R x n runs: against a per-frame count of the segments that started, for
    ragged words, one-segment models, segments that start at the template's end (empty) and several equal starts."""
    import importlib
    km = importlib.import_module("sr.recognition.kmeans")
    rng = np.random.default_rng(1)
    for trial in range(120):
        W, n = int(rng.integers(1, 5)), int(rng.integers(1, 7))
        n_temps = rng.integers(1, 5, size=W)
        lengths = rng.integers(1, 12, size=int(n_temps.sum()))
        starts = np.zeros((len(lengths), n), dtype=np.int64)
        for r, L in enumerate(lengths):
            starts[r, 1:] = np.sort(rng.integers(0, L + 1, size=n - 1))
        order, counts = km.segment_order(lengths, n_temps, starts, n)
        groups = [[] for _ in range(W * n)]
        f = r = 0
        for w in range(W):
            for _ in range(n_temps[w]):
                for t in range(lengths[r]):
                    groups[w * n + sum(1 for s in range(1, n) if starts[r, s] <= t)].append(f)
                    f += 1
                r += 1
        assert order.tolist() == [x for g in groups for x in g]
        assert counts.reshape(-1).tolist() == [len(g) for g in groups]
        ids, seg_lens = km._uniform_segments(lengths, n)           # kmeans.py:122-127: T // n frames each, the rest to the last
        want = np.concatenate([np.minimum(np.arange(L) // max(L // n, 1), n - 1) if L // n else np.full(L, n - 1) for L in lengths])
        np.testing.assert_array_equal(ids, want)
        assert ids.dtype == np.int32 and seg_lens.tolist() == [[L // n] * n for L in lengths]


def test_host_workspace_is_kept_and_never_shared():
    import importlib
    km = importlib.import_module("sr.recognition.kmeans")
    a, rel_a = km.host_workspace((100, 39))
    b, rel_b = km.host_workspace((10, 3))                          # while the first one is out: an ordinary array
    assert a.shape == (100, 39) and a.dtype == np.float64 and not np.shares_memory(a, b)
    rel_b()
    rel_a()
    c, rel_c = km.host_workspace((50, 39))
    assert np.shares_memory(a, c)                                  # the kept buffer again
    rel_c()
    d, rel_d = km.host_workspace((400, 39))                        # grows
    assert d.shape == (400, 39)
    rel_d()
    import os
    os.environ["GMMHMM_HOST_WORKSPACE_MB"] = "0"
    try:
        e, rel_e = km.host_workspace((400, 39))
        assert not np.shares_memory(d, e)
        rel_e()
    finally:
        del os.environ["GMMHMM_HOST_WORKSPACE_MB"]


def test_threaded_concat_rows_is_np_concatenate():
    """kmeans.concat_rows: C-contiguous float64 pieces through csrc/hostcopy.c (threads, GIL released), everything else through
    numpy -- same bytes either way; sizes that do not add up are an error, not a partial copy."""
    import importlib
    import importlib.util
    import os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gmmhmm_build_for_test", os.path.join(here, "speech-recognition_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build_hostcopy(verbose=False)
    km = importlib.import_module("sr.recognition.kmeans")
    from sr.recognition import _hostcopy
    rng = np.random.default_rng(5)
    parts = [rng.normal(size=(int(n), 39)) for n in rng.integers(0, 150, size=700)]
    ref = np.concatenate(parts)
    for threads in (1, 3, 8, 64):
        out = np.full(ref.shape, np.nan)
        _hostcopy.concat_rows(parts, out, threads)
        np.testing.assert_array_equal(out, ref)
    big = [rng.normal(size=(20000, 39)) for _ in range(3)] + [np.zeros((0, 39))]          # above the one-thread size
    out = np.empty((60000, 39))
    assert km.concat_rows(big, out) is out
    np.testing.assert_array_equal(out, np.concatenate(big))
    out = np.empty((4, 2))
    km.concat_rows([np.ones((2, 2), dtype=np.float32), [[2.0, 3.0]], rng.normal(size=(3, 4))[:1, ::2] * 0], out)   # numpy's route
    assert out.tolist() == [[1, 1], [1, 1], [2, 3], [0, 0]]
    with pytest.raises(TypeError):
        _hostcopy.concat_rows([np.ones((2, 2), dtype=np.float32)], np.empty((2, 2)))
    with pytest.raises(TypeError):
        _hostcopy.concat_rows([np.ones((4, 4))[:, :2]], np.empty((4, 2)))
    with pytest.raises(TypeError):                                  # widths that only ADD UP to the right size
        _hostcopy.concat_rows([np.ones((2, 3)), np.ones((3, 2)), np.ones((3, 4))], np.empty((8, 3)))
    with pytest.raises(ValueError):
        _hostcopy.concat_rows(parts, np.empty((len(ref) + 1, 39)))
    with pytest.raises(ValueError):
        _hostcopy.concat_rows(parts, np.empty((len(ref) - 1, 39)))
    _hostcopy.concat_rows([], np.empty((0, 39)))


def test_fast_allclose_is_numpys(R):
    """GMM.__eq__ (hmm_state.py:161-170) compares with np.allclose; `_allclose` is its test for finite arrays and calls it
    for everything else."""
    from sr.recognition.hmm_state import _allclose
    rng = np.random.default_rng(2)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                             # (inf - inf must stay silent, as in np.isclose)
        for _ in range(4000):
            n = int(rng.integers(1, 6))
            a = rng.normal(size=n) * 10.0 ** rng.integers(-9, 3)
            b = a + rng.normal(size=n) * 10.0 ** rng.integers(-12, -3) if rng.random() < 0.7 else rng.normal(size=n)
            if rng.random() < 0.25:
                i = rng.integers(0, n)
                a[i] = rng.choice([np.inf, -np.inf, np.nan])
                if rng.random() < 0.5:
                    b[i] = a[i]
            assert _allclose(a, b) == bool(np.allclose(a, b))
    assert _allclose([1.0, 2.0], np.array([1.0, 2.0 + 1e-9])) and not _allclose(np.ones(3), np.ones(1) * 5) and _allclose(np.ones(3), 1.0)


def test_id_hand_over_serves_eight_ranks_and_ignores_strangers():
    """parallel.exchange_from_rank0 at the world size the scaling run uses (8): rank 0 hands its 128-byte payload to seven
    peers that arrive in any order, a connection that does not open with the magic word / world size / token is not
    counted, and a peer that arrives before rank 0 listens keeps retrying."""
    import socket
    import threading
    import time
    from sr.recognition.parallel import exchange_from_rank0
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world, payload, got, errs = 8, bytes(range(128)), {}, []

    def run(rank, delay):
        try:
            time.sleep(delay)
            got[rank] = exchange_from_rank0(rank, world, lambda: payload, addr="127.0.0.1", ports=[port], timeout=60.0, token=b"run-42")
        except Exception as e:            # noqa: BLE001
            errs.append((rank, repr(e)))

    def stranger():
        time.sleep(0.4)
        for _ in range(3):
            try:
                with socket.create_connection(("127.0.0.1", port), timeout=2.0) as c:
                    c.sendall(b"GET / HTTP/1.0\r\n\r\n" + b"x" * 64)
                    time.sleep(0.05)
            except OSError:
                pass

    thr = [threading.Thread(target=run, args=(r, 0.3 if r == 0 else 0.05 * (r % 3))) for r in range(world)]
    thr.append(threading.Thread(target=stranger))
    for t in thr:
        t.start()
    for t in thr:
        t.join(90)
    assert not errs, errs
    assert sorted(got) == list(range(world)) and all(v == payload for v in got.values())
    # another run's token is not served: the peer gives up at its deadline, rank 0 at its own
    got.clear()

    def wrong():
        try:
            exchange_from_rank0(1, 2, lambda: b"", addr="127.0.0.1", ports=[port], timeout=3.0, token=b"other-run")
            errs.append("a foreign token was served")
        except RuntimeError:
            pass

    def right0():
        try:
            exchange_from_rank0(0, 2, lambda: payload, addr="127.0.0.1", ports=[port], timeout=4.0, token=b"run-42")
            errs.append("rank 0 counted a foreign peer")
        except RuntimeError:
            pass
    ts = [threading.Thread(target=right0), threading.Thread(target=wrong)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(30)
    assert not errs, errs
