# -*- coding: utf-8 -*-
"""Batched drivers on top of the HIP kernels -- the call patterns of the reference's
`sr/core.py:test` (isolated-word arg-min over models, core.py:63-94) and `main.py`
(continuous-digit decode through a K-layer lattice + path post-processing,
main.py:35,59-67), over whole utterance batches resident in HBM.

These are what the benchmark and a serving caller use; the per-utterance functions
in `decode.py` give the same numbers one utterance at a time.
"""
import numpy as np

from . import _hip
from . import _pack
from .continuous_speech import packed_lattice, packed_loop_lattice

__all__ = ["IsolatedWordRecognizer", "ContinuousDecoder", "InFlight", "path_to_words", "sequence_report", "train_words"]


def _stack_models(ctx, models):
    """(states per word, packed Gaussians of all words, single-Gaussian flag).  A model trained with use_gmm=False has no
    `gmm_states` (hmm.py:57-76): its states are the rows of `mu` / `sigma`, scored by mahalanobis() (hmm.py:133-134)."""
    single = [not getattr(m, "use_gmm", True) or m.gmm_states is None for m in models]
    assert all(single) or not any(single), "word models with and without mixtures cannot share one recogniser"
    if single[0]:
        n = len(models[0].mu)
        for m in models:
            assert len(m.mu) == n
        mean = np.concatenate([np.asarray(m.mu, dtype=np.float64) for m in models])[:, None, :]
        var = np.concatenate([np.asarray(m.sigma, dtype=np.float64) for m in models])[:, None, :]
        key = (id(ctx), _pack._digest(mean, var, np.zeros(1)))
        gmm = _pack._gmm_cache.lookup(key, lambda: _hip.PackedGMM(ctx, mean, var, np.ones((len(mean), 1))))
        return n, gmm, True
    n = len(models[0].gmm_states)
    for m in models:
        assert len(m.gmm_states) == n
    return n, _pack.device_gmm(ctx, [s for m in models for s in m.gmm_states]), False


def train_words(templates_by_word, n_segments, n_gaussians=4, use_gmm=True, use_em=True):
    """`HMM(n_segments).fit(ys, n_gaussians, use_gmm, use_em)` for every word -- what sr/core.py:47-60 (make_HMM /
    train) does digit after digit -- in ONE pass over all words:

      1. segmental k-means of all words in lock-step (`kmeans.skmeans_multi`: one alignment launch and one
         (word, segment) reduction per iteration for all templates of all words),
      2. the mixtures of all W x n states refit in one lock-step session (`lockstep.LockstepFitter`; numpy's global
         generator is consumed word after word, state after state -- the order of the sequential loop),
      3. all templates re-aligned against their word's mixtures in one launch (align_gmm_states, hmm.py:95).

    templates_by_word: list over words of lists of [T_r, D] arrays.  Returns the list of trained `HMM`s; under the same
    numpy seed they equal the models of the word-after-word loop (alignment paths and cluster ids exactly, parameters to
    rounding: the words only share launches)."""
    import importlib
    _km = importlib.import_module(__package__ + ".kmeans")    # (the package re-exports the FUNCTION kmeans under that name)
    from .hmm import HMM
    from .hmm_state import GMM
    from .lockstep import LockstepFitter
    import os
    import sys
    import time
    marks = [("start", time.perf_counter())] if os.environ.get("GMMHMM_TRAIN_WORDS_TIMES") else None

    def mark(name):
        if marks is not None:
            marks.append((name, time.perf_counter()))
    W = len(templates_by_word)
    models = [HMM(n_segments) for _ in range(W)]
    for h in models:
        h.use_gmm, h.use_em = use_gmm, use_em
    if use_gmm:
        print('Doing segmental k-means')
    n = n_segments
    kmax = 2 ** max(1, int(np.log(n_gaussians)))
    on_device = bool(W) and all(len(ts) for ts in templates_by_word) and _km._device_skmeans_possible(templates_by_word[0]) and 2 <= n <= 32
    if not on_device:
        return _train_words_by_segments(templates_by_word, models, n, n_gaussians, use_gmm, use_em)
    # ONE resident batch of all templates for the three stages (round 4 uploaded the frames three times and regrouped them
    # on the host twice: 28 of the 50 ms of ten words x 200 templates)
    ctx = _hip.default_context()
    n_temps = np.array([len(ts) for ts in templates_by_word], dtype=np.int64)
    lengths = np.array([len(t) for ts in templates_by_word for t in ts], dtype=np.int64)
    D = np.asarray(templates_by_word[0][0]).shape[1]
    X, release = _km.host_workspace((int(lengths.sum()), D))
    try:
        _km.concat_rows([t for ts in templates_by_word for t in ts], X)
    except BaseException:
        release()
        raise
    off_t = np.concatenate([[0], np.cumsum(lengths)])
    mark("concatenate")
    try:
        return _train_words_resident(ctx, X, off_t, templates_by_word, models, n, n_gaussians, use_gmm, use_em, kmax, lengths, n_temps, mark, marks)
    finally:
        release()


def _train_words_resident(ctx, X, off_t, templates_by_word, models, n, n_gaussians, use_gmm, use_em, kmax, lengths, n_temps, mark, marks):
    """The three stages of `train_words` on ONE resident batch of all templates (X: the caller's frames back to back)."""
    import importlib
    import sys
    _km = importlib.import_module(__package__ + ".kmeans")
    from .hmm_state import GMM
    from .lockstep import LockstepFitter
    W, D = len(templates_by_word), X.shape[1]
    frames = _hip.Batch(ctx, feats=X, offsets=off_t)
    mark("upload")
    try:
        fitted = _km.skmeans_multi(templates_by_word, n, frames=frames)
        mark("segmental k-means")
        tpl_off = np.concatenate([[0], np.cumsum(n_temps)])
        for w, h in enumerate(models):
            h.mu, h.sigma, h.transitions, _ = fitted[w]
        starts = np.concatenate([f[3] for f in fitted])
        order, counts = _km.segment_order(lengths, n_temps, starts, n)
        if not use_gmm:
            for h, segs in zip(models, _km.split_segments(_km.gather_rows(X, order), counts)):
                h.segments = segs
            return models
        for h in models:
            h.gmm_states = [GMM(m, s, n_gaussians) for m, s in zip(h.mu, h.sigma)]
        mark("segment order, GMM objects")
        # the mixtures of all W x n states: their frames gathered on the device from the resident batch
        fitter = LockstepFitter(None, ctx=ctx, source=(frames, order), lengths=[int(c) for c in counts.reshape(-1)], dim=D, kmax=kmax)
        try:
            fitter.split_and_fit([g for h in models for g in h.gmm_states],
                                 start_centroids=np.concatenate([h.mu for h in models]),
                                 weight_divisor=[int(c) for c in counts.reshape(-1)],     # hmm.py:108,118: len(segment)
                                 n_gaussians=n_gaussians, use_em=use_em)
        finally:
            fitter.close()
        mark("refit")
        # re-alignment of every template against its word's mixtures (hmm.py:95), all words in one launch
        utt_word = np.repeat(np.arange(W), n_temps).astype(np.int32)
        gmm = _pack.device_gmm(ctx, [g for h in models for g in h.gmm_states])
        lat = _hip.Lattices(ctx, [_pack.graph_from_dense(np.arange(n) + w * n, h.transitions, [0], [n - 1]) for w, h in enumerate(models)])
        try:
            frames.loglik(gmm, fetch=False, state_ranges=(utt_word * n, utt_word * n + n))
            res = lat.viterbi(frames, utt_lattice=utt_word, want_path=True, flat_paths=True)
        finally:
            lat.close()
        mark("re-alignment")
    finally:
        frames.close()
    # get_segments_from_path for all templates at once: visits of every chain row on the path, cumulated (kmeans.py:98-108)
    tid = np.repeat(np.arange(len(lengths)), res["path_len"])
    visits = np.bincount(tid * n + res["path_flat"][:, 0], minlength=len(lengths) * n).reshape(len(lengths), n)
    starts_all = np.zeros((len(lengths), n), dtype=np.int64)
    np.cumsum(visits[:, :-1], axis=1, out=starts_all[:, 1:])
    order, counts = _km.segment_order(lengths, n_temps, starts_all, n)
    mark("segment order")
    for h, segs in zip(models, _km.split_segments(_km.gather_rows(X, order), counts)):
        h.segments = segs
    mark("segments")
    if marks is not None:
        sys.stderr.write("train_words [ms]: " + ", ".join("%s %.2f" % (b[0], (b[1] - a[1]) * 1e3) for a, b in zip(marks, marks[1:])) + "\n")
    return models


def _starts_from_paths(paths, n):
    """get_segments_from_path for all templates at once: visits of every chain row on the path, cumulated (kmeans.py:98-108)."""
    plen = np.array([len(p) for p in paths], dtype=np.int64)
    rows = (np.concatenate([np.asarray(p)[:, 0] for p in paths]) if len(paths) and plen.sum() else np.zeros(0, dtype=np.int64)).astype(np.int64)
    tid = np.repeat(np.arange(len(paths)), plen)
    counts = np.bincount(tid * n + rows, minlength=len(paths) * n).reshape(len(paths), n)   # (rows: chain rows 0 .. n-1 of the word's own graph)
    starts_all = np.zeros((len(paths), n), dtype=np.int64)
    np.cumsum(counts[:, :-1], axis=1, out=starts_all[:, 1:])
    return starts_all


def _train_words_by_segments(templates_by_word, models, n_segments, n_gaussians, use_gmm, use_em):
    """train_words with the frames regrouped on the host (one feature dimension, word lists with an empty entry, the test
    double of the binding): the stages of `HMM.fit`, the refit and the re-alignment still one launch sequence for all words."""
    import importlib
    _km = importlib.import_module(__package__ + ".kmeans")
    from .hmm_state import GMM
    from .lockstep import LockstepFitter
    W = len(templates_by_word)
    for w, h in enumerate(models):
        h.mu, h.sigma, h.transitions, h.segments = _km.skmeans(templates_by_word[w], n_segments, return_segmented_data=True)
    if not use_gmm:
        return models
    for h in models:
        h.gmm_states = [GMM(m, s, n_gaussians) for m, s in zip(h.mu, h.sigma)]
    fitter = LockstepFitter([seg for h in models for seg in h.segments], kmax=2 ** max(1, int(np.log(n_gaussians))))
    try:
        fitter.split_and_fit([g for h in models for g in h.gmm_states],
                             start_centroids=np.concatenate([h.mu for h in models]),
                             weight_divisor=[len(seg) for h in models for seg in h.segments],     # hmm.py:108,118
                             n_gaussians=n_gaussians, use_em=use_em)
    finally:
        fitter.close()
    ctx = _hip.default_context()
    flat = [np.asarray(t, dtype=np.float64) for ts in templates_by_word for t in ts]
    utt_word = np.repeat(np.arange(W), [len(ts) for ts in templates_by_word]).astype(np.int32)
    n = n_segments
    gmm = _pack.device_gmm(ctx, [g for h in models for g in h.gmm_states])
    lat = _hip.Lattices(ctx, [_pack.graph_from_dense(np.arange(n) + w * n, h.transitions, [0], [n - 1]) for w, h in enumerate(models)])
    frames = _hip.Batch(ctx, flat)
    try:
        frames.loglik(gmm, fetch=False, state_ranges=(utt_word * n, utt_word * n + n))
        res = lat.viterbi(frames, utt_lattice=utt_word, want_path=True)
    finally:
        frames.close()
        lat.close()
    off = np.concatenate([[0], np.cumsum([len(ts) for ts in templates_by_word])])
    starts_all = _starts_from_paths(res["paths"], n)
    for w, h in enumerate(models):
        h.segments = _km.segment_data_fast(templates_by_word[w], n, starts_all[off[w]:off[w + 1]])
    return models


class IsolatedWordRecognizer:
    """Scores every utterance against every word model and returns the arg-min word
    (core.py:82-87: `costs = [m.evaluate(x) for m in models]; argmin`).

    The W word chains are stacked into one graph (W start rows, W end rows), so one
    gh_loglik + one gh_viterbi launch replaces U x W calls of HMM.evaluate."""

    def __init__(self, models, device=None, dtype=np.float64, ctx=None):
        self.ctx = ctx if ctx is not None else _hip.default_context(device)
        self.dtype = dtype
        self.W = len(models)
        self.n, self.gmm, self.single = _stack_models(self.ctx, models)
        n, W = self.n, self.W
        to, frm, cost = [], [], []
        for i, m in enumerate(models):
            t = np.asarray(m.transitions, dtype=np.float64)
            a, b = np.nonzero(~np.isinf(t))
            to.append(a + i * n)
            frm.append(b + i * n)
            cost.append(t[a, b])
        graph = dict(row_state=np.arange(W * n, dtype=np.int32), arc_to=np.concatenate(to),
                     arc_from=np.concatenate(frm), arc_cost=np.concatenate(cost),
                     start_rows=np.arange(W) * n, end_rows=np.arange(W) * n + n - 1)
        self.lat = _hip.Lattices(self.ctx, [graph])

    def costs(self, batch):
        """[U, W] matrix of HMM.evaluate values for a resident `_hip.Batch`.  Models with ONE Gaussian per state
        (use_gmm=False, or one-component mixtures) are scored inside the dynamic program -- the reference's own
        single-Gaussian path, hmm.py:133-134 (`dtw` with `mahalanobis`): no [N, S] likelihood matrix exists
        (gh_viterbi_fused); mixtures run gh_loglik + gh_viterbi."""
        if self.gmm.M == 1:
            r = self.lat.viterbi(batch, want_path=False, fused_gmm=self.gmm, log_domain=self.single)
        else:
            batch.loglik(self.gmm, fetch=False)
            r = self.lat.viterbi(batch, want_path=False)
        return r["end_cost_flat"].reshape(batch.U, self.W)

    def recognize(self, xs):
        """xs: list of [T_u, D] arrays -> (words [U], costs [U, W])."""
        batch = _hip.Batch(self.ctx, xs, dtype=self.dtype)
        try:
            c = self.costs(batch)
        finally:
            batch.close()
        return np.argmin(c, axis=1), c

    def accuracy(self, xs, words, verbose=False):
        """The report of `sr/core.py:test` (core.py:63-94) as a call: fraction of utterances whose cheapest model is
        the labelled word.  The reference keeps the FIRST of equal minima (`if cost < c`), which is np.argmin's rule.
        Returns (n_passed / n_tests, recognised words [U])."""
        got, _ = self.recognize(xs)
        words = np.asarray(words)
        if verbose:
            for w in words[got != words]:
                print("Digit:", int(w), "is wrong")                       # core.py:93
        return float(np.sum(got == words)) / len(words), got


def sequence_report(decoded, labels, verbose=False):
    """The tally at the end of the reference's `main.py` (:69-84): an utterance is correct when its decoded word
    string equals the label string; for a wrong one the number of differing positions (np.count_nonzero(matched - l),
    strings of equal length -- the K-layer lattice always decodes exactly K words) counts against the digit accuracy.
    Returns dict(sequence_accuracy, digit_accuracy, n_correct, n_digits, n_digit_errors)."""
    correct = n_digits = n_diff = 0
    for got, want in zip(decoded, labels):
        got, want = [int(v) for v in got], [int(v) for v in want]
        n_digits += len(want)
        if got == want:
            correct += 1
            if verbose:
                print('Correct:', got)
            continue
        if verbose:
            print('Incorrect:', got, want)
        if len(got) != len(want):
            # main.py:79 subtracts the two arrays, which numpy refuses for different lengths (the loop grammar can
            # produce them): every position beyond the shorter string counts as a difference here
            k = min(len(got), len(want))
            d = int(np.count_nonzero(np.asarray(got[:k]) - np.asarray(want[:k]))) + max(len(got), len(want)) - k
        else:
            d = int(np.count_nonzero(np.asarray(got) - np.asarray(want)))
        if verbose:
            print('Diff:', d)
        n_diff += d
    n = max(len(labels), 1)
    return dict(sequence_accuracy=correct / n, digit_accuracy=(n_digits - n_diff) / max(n_digits, 1), n_correct=correct,
                n_digits=n_digits, n_digit_errors=n_diff)


def path_to_words(path, row_state, n_per_word):
    """main.py:59-67: reversed path rows -> drop consecutive duplicates -> first emitting
    row of every run between non-emitting rows -> word index of that row."""
    if len(path) == 0:
        return []
    rows = np.asarray(path)[:, 0][::-1]
    rows = rows[np.insert(np.diff(rows) != 0, 0, True)]
    st = np.asarray(row_state)[rows]
    emitting = st >= 0
    # first emitting row of each maximal emitting run
    first = emitting & np.insert(~emitting[:-1], 0, True)
    return [int(s) // n_per_word for s in st[first]]


class ContinuousDecoder:
    """Continuous-word decode over all `models`, then `path_to_words`.

    grammar="layers": the reference's K-layer lattice (main.py:35: build_state_sequences(models,
    [[0..W-1]] * K)) -- exactly `n_layers` words; end points = last layer's final states in the last
    column (main.py:60).
    grammar="loop": the word-loop grammar (`build_loop_grammar`) -- any number of
    words, cost = min over K of the K-layer costs, on a graph of 2 + W*n rows instead of
    1 + K*(W*n + 1)."""

    def __init__(self, models, n_layers=7, device=None, dtype=np.float64, grammar="layers", word_penalty=0.0, ctx=None):
        self.ctx = ctx if ctx is not None else _hip.default_context(device)
        self.dtype = dtype
        # (single-Gaussian word models decode through the likelihood kernel here: mahalanobis() is the one-component
        #  GMM.evaluate with weight 1, the same number up to rounding)
        self.n, self.gmm, _ = _stack_models(self.ctx, models)
        W = len(models)
        wt = [m.transitions for m in models]
        if grammar == "layers":
            graph, self.nes_rows = packed_lattice(wt, self.n, [list(range(W))] * n_layers)
            self._max_labels = lambda T: n_layers + 1
        elif grammar == "loop":
            graph, self.nes_rows = packed_loop_lattice(wt, self.n, word_penalty)
            self._max_labels = lambda T: T // max(1, self.n - 1) + 2      # a word spans at least n - 1 column steps
        else:
            raise ValueError("grammar must be 'layers' or 'loop', not %r" % (grammar,))
        self.grammar = grammar
        self.row_state = graph["row_state"]
        self.lat = _hip.Lattices(self.ctx, [graph])

    def decode_batch(self, batch, want_path=False):
        """Word-index lists of every utterance (+ the raw result dict).  By default the paths stay on the device
        and only the decoded word sequences come back (gh_viterbi_labels); want_path=True also returns the
        reference-style (row, column) paths in `r["paths"]` and derives the words from them on the host."""
        batch.loglik(self.gmm, fetch=False)
        if want_path:
            r = self.lat.viterbi(batch, want_path=True)
            return [path_to_words(p, self.row_state, self.n) for p in r["paths"]], r
        row_word = np.where(self.row_state >= 0, self.row_state // self.n, -1).astype(np.int32)
        r = self.lat.viterbi_labels(batch, row_word, max_labels=self._max_labels(batch.lengths))
        return [[int(w) for w in l] for l in r["labels"]], r

    def decode(self, xs):
        """xs: list of [T_u, D] arrays -> list of word-index lists."""
        batch = _hip.Batch(self.ctx, xs, dtype=self.dtype)
        try:
            return self.decode_batch(batch)[0]
        finally:
            batch.close()

    def accuracy(self, xs, labels, verbose=False):
        """Decode `xs` and tally against the label strings like main.py:54-84 -- see `sequence_report`."""
        return sequence_report(self.decode(xs), labels, verbose=verbose)


class InFlight:
    """Several batches in flight on ONE GPU: `n_lanes` contexts (HIP stream, scratch, pinned buffers), one host
    thread each, every lane with its own resident copy of whatever `make_worker(ctx)` builds (e.g. a recogniser).
    The host side of a batch (result copy-back, arg-min, Python) then overlaps the other lanes' kernels -- what
    `bench.py --inflight 2` measures (+17 % throughput on configs[1]).  ctypes releases the GIL during calls.

        pool = InFlight(lambda ctx: IsolatedWordRecognizer(models, ctx=ctx))
        results = pool.map(lambda rec, xs: rec.recognize(xs), list_of_utterance_lists)      # in input order
    """

    def __init__(self, make_worker, n_lanes=2, device=None):
        dev = _hip.default_context(device).device
        self.ctxs = [_hip.Context(dev) for _ in range(max(1, int(n_lanes)))]
        self.workers = [make_worker(c) for c in self.ctxs]

    def map(self, fn, items):
        import threading
        items = list(items)
        out = [None] * len(items)
        lock, state = threading.Lock(), {"next": 0, "err": None}

        def run(worker):
            try:
                while True:
                    with lock:
                        k = state["next"]
                        if k >= len(items) or state["err"] is not None:
                            return
                        state["next"] = k + 1
                    out[k] = fn(worker, items[k])
            except BaseException as e:  # surfaced in the caller's thread
                state["err"] = e
        threads = [threading.Thread(target=run, args=(w,)) for w in self.workers]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if state["err"] is not None:
            raise state["err"]
        return out

    def close(self):
        for w in self.workers:
            h = getattr(w, "lat", None)
            if h is not None and hasattr(h, "close"):
                h.close()
        for c in self.ctxs:
            _pack._gmm_cache.purge(c)     # handles are bound to the context they were created with
            _pack._lat_cache.purge(c)
            _pack._normal_cache.purge(c)
            c.close()
