# -*- coding: utf-8 -*-
"""Word lattices and embedded (Viterbi) training on the MI355X.

Mirror of the reference's `sr/recognition/continuous_speech.py`:
`build_state_sequences` (:13-53) and `continuous_train` (:56-179) with the same
signatures, side effects (one pickle per model and iteration) and convergence
rule.  The alignment of ALL utterances of an outer iteration is one batched
gh_loglik + gh_viterbi launch over features that stay resident in HBM for the
whole training run (one graph per distinct label sequence); the per-state refit
uses the HIP k-means / EM kernels through `kmeans` and `GMM.em`.
"""
import copy
import os
import pickle
import threading
import warnings
from typing import List, AnyStr

import numpy as np

from . import _hip
from . import _pack
from .kmeans import kmeans
from .hmm_state import GMM, NES, mahalanobis
from .hmm import HMM
from .lockstep import LockstepFitter, PartitionStream, fast_partition_ok

__all__ = ["build_state_sequences", "build_loop_grammar", "continuous_train", "forced_alignments", "aligned_frame_states",
           "cut_segments"]


def _layout(n_per_word, label_matrix):
    """Row bookkeeping shared by the object-level and the packed lattice builders.
    Returns (R, [(layer, label, first_row)], nes_rows)."""
    blocks, nes_rows, r = [], [0], 1
    for k, labels in enumerate(label_matrix):
        for l in labels:
            blocks.append((k, l, r))
            r += n_per_word
        nes_rows.append(r)
        r += 1
    return r, blocks, nes_rows


def build_state_sequences(hmms: List[HMM], label_matrix: List[List[int]]):
    """K-layer word lattice (continuous_speech.py:13-53).

    Row 0 is a non-emitting state; layer k lists the states of every word in
    `label_matrix[k]` (the state OBJECTS are shared between layers) and is closed by one
    more non-emitting state.  Zero-cost arcs lead from the non-emitting state in front of a
    layer to the first state of each of its words, and from each word's last state to the
    non-emitting state behind the layer.  All words must have the same number of states.

    :return: (states [R], dense transition costs [R,R] with +inf holes,
              rows of the last layer's final emitting states)"""
    n = len(hmms[0].gmm_states)
    R, blocks, nes_rows = _layout(n, label_matrix)
    trans = np.full((R, R), np.inf)
    seq = [None] * R
    for r in nes_rows:
        seq[r] = NES()
    last_layer = len(label_matrix) - 1
    ends = []
    for k, l, r0 in blocks:
        word = hmms[l]
        assert n == len(word.gmm_states)
        seq[r0:r0 + n] = word.gmm_states
        trans[r0:r0 + n, r0:r0 + n] = word.transitions
        trans[r0, nes_rows[k]] = 0
        trans[nes_rows[k + 1], r0 + n - 1] = 0
        if k == last_layer:
            ends.append(r0 + n - 1)
    return seq, trans, ends


def packed_lattice(word_transitions, n_per_word, label_matrix, state_base=None):
    """The same lattice as a graph dict for `_hip.Lattices`, without building R x R
    matrices or state objects: row_state = word * n + state (or state_base[word] + state)."""
    R, blocks, nes_rows = _layout(n_per_word, label_matrix)
    row_state = np.full(R, -1, dtype=np.int32)
    to, frm, cost, ends = [], [], [], []
    last_layer = len(label_matrix) - 1
    for k, l, r0 in blocks:
        base = l * n_per_word if state_base is None else state_base[l]
        row_state[r0:r0 + n_per_word] = base + np.arange(n_per_word)
        wt = np.asarray(word_transitions[l])
        i, j = np.nonzero(~np.isinf(wt))
        to += list(i + r0) + [r0, nes_rows[k + 1]]
        frm += list(j + r0) + [nes_rows[k], r0 + n_per_word - 1]
        cost += list(wt[i, j]) + [0.0, 0.0]
        if k == last_layer:
            ends.append(r0 + n_per_word - 1)
    return dict(row_state=row_state, arc_to=np.asarray(to, dtype=np.int32), arc_from=np.asarray(frm, dtype=np.int32),
                arc_cost=np.asarray(cost, dtype=np.float64), start_rows=np.array([0], dtype=np.int32),
                end_rows=np.asarray(ends, dtype=np.int32)), nes_rows


def _loop_layout(W, n):
    """Row bookkeeping of the word-loop grammar: (R, loop_row, row_of [W, n])."""
    assert n >= 2, "a loop grammar needs words of at least two states"
    loop_row = 1 + W * (n - 1)
    row_of = np.empty((W, n), dtype=np.int64)
    row_of[:, 1:] = 1 + np.arange(W)[:, None] * (n - 1) + np.arange(n - 1)[None, :]
    row_of[:, 0] = loop_row + 1 + np.arange(W)
    return 2 + W * n, loop_row, row_of


def build_loop_grammar(hmms: List[HMM], word_penalty=0.0):
    """Word LOOP grammar over `hmms` (the reference only has the exactly-K-words
    lattice of `build_state_sequences`): any number of words, each followed by any word.

    Same return convention as `build_state_sequences` -- (states [R], dense costs [R,R], end rows) --
    and decodable by `decode_hmm_states` as is, because the loop is expressed by ROW ORDER: the DP
    reads same-column origins only from rows it has already visited in the column
    (decode.py:97-98,109-111), so rows are laid out as
        0: non-emitting start | states 1..n-1 of every word | one non-emitting loop row |
        state 0 of every word.
    Word ends feed the loop row, the loop row (cost `word_penalty`) and the start row (cost 0) feed
    the first states, all as same-column hops; a word boundary therefore scores its frame with both
    words, exactly like the K-layer lattice, and the decode cost equals the minimum over K of the
    K-layer decode costs."""
    n = len(hmms[0].gmm_states)
    W = len(hmms)
    R, loop_row, row_of = _loop_layout(W, n)
    trans = np.full((R, R), np.inf)
    seq = [None] * R
    seq[0], seq[loop_row] = NES(), NES()
    ends = []
    for w, word in enumerate(hmms):
        assert n == len(word.gmm_states)
        rows = row_of[w]
        for i in range(n):
            seq[rows[i]] = word.gmm_states[i]
        trans[np.ix_(rows, rows)] = word.transitions
        trans[rows[0], 0] = 0
        trans[rows[0], loop_row] = word_penalty
        trans[loop_row, rows[n - 1]] = 0
        ends.append(int(rows[n - 1]))
    return seq, trans, ends


def packed_loop_lattice(word_transitions, n_per_word, word_penalty=0.0, state_base=None):
    """`build_loop_grammar` as a graph dict for `_hip.Lattices` (row_state = word * n + state)."""
    W = len(word_transitions)
    R, loop_row, row_of = _loop_layout(W, n_per_word)
    row_state = np.full(R, -1, dtype=np.int32)
    to, frm, cost, ends = [], [], [], []
    for w in range(W):
        base = w * n_per_word if state_base is None else state_base[w]
        rows = row_of[w]
        row_state[rows] = base + np.arange(n_per_word)
        wt = np.asarray(word_transitions[w])
        i, j = np.nonzero(~np.isinf(wt))
        to += list(rows[i]) + [rows[0], rows[0], loop_row]
        frm += list(rows[j]) + [0, loop_row, rows[n_per_word - 1]]
        cost += list(wt[i, j]) + [0.0, float(word_penalty), 0.0]
        ends.append(int(rows[n_per_word - 1]))
    return dict(row_state=row_state, arc_to=np.asarray(to, dtype=np.int32), arc_from=np.asarray(frm, dtype=np.int32),
                arc_cost=np.asarray(cost, dtype=np.float64), start_rows=np.array([0], dtype=np.int32),
                end_rows=np.asarray(ends, dtype=np.int32)), [0, loop_row]


def transcript_state_sets(label_seqs, n, n_words):
    """(range_off [U+1], lo, hi) for `Batch.loglik(state_sets=...)`: the state ranges of the DISTINCT words of every
    transcript (word w owns the states [w * n, (w + 1) * n)), consecutive words merged; an empty transcript asks for
    every state."""
    off, lo, hi = [0], [], []
    for labels in label_seqs:
        words = sorted(set(int(l) for l in labels)) if len(labels) else list(range(n_words))
        start = prev = None
        for w in words:
            if prev is not None and w == prev + 1:
                prev = w
                continue
            if start is not None:
                lo.append(start * n); hi.append((prev + 1) * n)
            start = prev = w
        if start is not None:
            lo.append(start * n); hi.append((prev + 1) * n)
        off.append(len(lo))
    return np.asarray(off, dtype=np.int64), np.asarray(lo, dtype=np.int32), np.asarray(hi, dtype=np.int32)


class AlignmentPlan:
    """What the alignment step of every outer iteration needs and that never changes between iterations: the distinct
    transcripts (one forced-alignment graph each), every utterance's graph, the flattened label strings the library
    takes, the state sets whose likelihoods an utterance needs -- and ONE packed model that is re-packed in place
    (gh_gmm_update) instead of being rebuilt.  Built once per `continuous_train` call (it used to be rebuilt, label
    string by label string, in every iteration)."""

    def __init__(self, label_seqs, n, n_words):
        self.n, self.n_words = n, n_words
        for u, labels in enumerate(label_seqs):
            if len(labels) == 0:     # (no layer, no end row: the library rejects such a graph; say which utterance it is)
                raise ValueError("utterance %d has an empty transcript: a forced alignment needs at least one word" % u)
        self.state_sets = transcript_state_sets(label_seqs, n, n_words)
        keys, self.transcripts, self.utt_graph = {}, [], np.empty(len(label_seqs), dtype=np.int32)
        for u, labels in enumerate(label_seqs):
            key = tuple(int(l) for l in labels)
            if key not in keys:
                keys[key] = len(self.transcripts)
                self.transcripts.append(key)
            self.utt_graph[u] = keys[key]
        self.flat = _hip.Lattices.flatten_transcripts(self.transcripts) if hasattr(_hip.Lattices, "flatten_transcripts") else None
        self.gmm = None
        self.frame_state_buf = None       # int32 [N]: the alignment's result lands in the same pages every iteration

    def packed_model(self, ctx, models):
        states = [s for m in models for s in m.gmm_states]
        if not hasattr(_hip.PackedGMM, "update"):
            return _pack.device_gmm(ctx, states)
        means, vars_, w = _pack.stack_gmms(states)
        if self.gmm is None or (self.gmm.S, self.gmm.M, self.gmm.D) != means.shape:
            self.close()
            self.gmm = _hip.PackedGMM(ctx, means, vars_, w)
        else:
            self.gmm.update(means, vars_, w)
        return self.gmm

    def close(self):
        if self.gmm is not None:
            self.gmm.close()
            self.gmm = None


def _alignment_lattices(frames, models, label_seqs, plan=None):
    """Likelihoods of every utterance against the states of its own words, and its one-word-per-layer lattice
    (continuous_speech.py:80-82): (device lattices, distinct transcripts, transcript index per utterance)."""
    ctx = frames.ctx
    n = len(models[0].gmm_states)
    own = plan is None
    if own:
        plan = AlignmentPlan(label_seqs, n, len(models))
    gmm = plan.packed_model(ctx, models) if not own else _pack.device_gmm(ctx, [s for m in models for s in m.gmm_states])
    # an utterance's lattice only contains the states of its own words: likelihoods for those states only
    frames.loglik(gmm, fetch=False, state_sets=plan.state_sets)
    kw = {"flat": plan.flat} if plan.flat is not None else {}
    lat = _hip.Lattices.from_transcripts(ctx, [m.transitions for m in models], n, plan.transcripts, **kw)
    return lat, plan.transcripts, plan.utt_graph


def transcript_row_state(labels, n):
    """row_state of the forced-alignment graph of `labels` (see `packed_lattice`): -1 on non-emitting rows, else
    word * n + state."""
    rs = np.full(len(labels) * (n + 1) + 1, -1, dtype=np.int32)
    for k, l in enumerate(labels):
        rs[k * (n + 1) + 1:k * (n + 1) + 1 + n] = l * n + np.arange(n)
    return rs


def forced_alignments(frames, models, label_seqs):
    """Viterbi alignment of every utterance through its own one-word-per-layer lattice
    (continuous_speech.py:80-89), all utterances in one launch.
    Returns (paths, row_state per utterance) with row_state[r] = -1 on non-emitting rows,
    else word * n + state."""
    lat, transcripts, utt_graph = _alignment_lattices(frames, models, label_seqs)
    try:
        res = lat.viterbi(frames, utt_lattice=utt_graph, want_path=True)
    finally:
        lat.close()
    n = len(models[0].gmm_states)
    row_states = [transcript_row_state(t, n) for t in transcripts]
    return res["paths"], [row_states[g] for g in utt_graph]


def aligned_frame_states(frames, models, label_seqs, plan=None):
    """Alignment AND regrouping (continuous_speech.py:80-106) in one launch sequence, nothing but one int per frame
    coming back: (frame_state [N]: word * n + state of the training data the frame joins, -1 for none;
    segment_start bool [N]: first frame of a segment).  Equals `cut_segments` applied to `forced_alignments`."""
    lat, _, utt_graph = _alignment_lattices(frames, models, label_seqs, plan)
    try:
        if plan is not None:
            if plan.frame_state_buf is None or plan.frame_state_buf.shape != (frames.N,):
                plan.frame_state_buf = np.empty(frames.N, dtype=np.int32)
            try:
                res = lat.align_segments(frames, utt_lattice=utt_graph, out=plan.frame_state_buf)
            except TypeError:          # (a test double of the binding without `out`)
                res = lat.align_segments(frames, utt_lattice=utt_graph)
        else:
            res = lat.align_segments(frames, utt_lattice=utt_graph)
    finally:
        lat.close()
    return res["frame_state"], res["segment_start"]


def aligned_runs(frames, models, label_seqs, plan=None):
    """The same alignment with the RUNS as the result (`Lattices.align_runs`, gh_align_runs): dict(state [R], start [R] (row
    of `frames`), length [R]) in utterance / time order, or None when the binding has no such call (the test double)."""
    lat, _, utt_graph = _alignment_lattices(frames, models, label_seqs, plan)
    try:
        if not hasattr(lat, "align_runs"):
            return None
        return lat.align_runs(frames, utt_lattice=utt_graph)
    finally:
        lat.close()


def cut_segments(path, row_state):
    """Frame ranges per visited state from one alignment (continuous_speech.py:90-106).

    Walking the path from start to end: a run opens at the first cell of an emitting row
    seen while no run is open; a cell of a different row closes the open run as
    [start, c) -- c being that cell's column -- provided start < c, and does not itself
    open a run.  Consequences kept from the reference: the frame on which a state is
    entered inside a word is dropped, the frame on a word boundary belongs to the next
    word only, and the final state's last run is never closed.
    Yields (row_state value, start, stop)."""
    open_row, open_at = None, None
    for r, c in path[::-1]:
        if open_at is None and row_state[r] >= 0:
            open_row, open_at = r, c
        if r != open_row and open_at is not None and open_at < c:
            yield int(row_state[open_row]), int(open_at), int(c)
            open_row, open_at = None, None


def _clone_models(models):
    """copy.deepcopy(models) for lists of GMM word models, without the generic walk: new HMM / GMM / MultivariateNormal
    objects with their own parameter arrays, state ids kept (a copy hashes like its original, as with deepcopy).  The
    D x D `inv_cov` matrices are SHARED with the originals: nothing writes into one (assigning `cov` installs a new
    matrix), and copying 400 of them was most of the 3.7 ms a deep copy of ten 5-state 8-mixture words took.
    Anything that is not such a model goes through copy.deepcopy."""
    if not all(type(m) is HMM and m.use_gmm and m.gmm_states is not None and all(type(g) is GMM for g in m.gmm_states)
               for m in models):
        return copy.deepcopy(models)
    memo = {}
    out = []
    for m in models:
        nm = HMM.__new__(HMM)
        memo[id(m)] = nm
        for key, v in m.__dict__.items():
            if key != "gmm_states":
                nm.__dict__[key] = v.copy() if type(v) is np.ndarray and v.dtype != object else copy.deepcopy(v, memo)
        states = []
        for g in m.gmm_states:
            ng = GMM.__new__(GMM)
            memo[id(g)] = ng
            for key, v in g.__dict__.items():
                if key == "dists":
                    continue
                if key == "parent":
                    ng.__dict__[key] = memo.get(id(v)) if v is not None and id(v) in memo else copy.deepcopy(v, memo)
                else:
                    ng.__dict__[key] = v.copy() if type(v) is np.ndarray and v.dtype != object else copy.deepcopy(v, memo)
            dists = []
            for d in g.dists:
                nd = d.__class__.__new__(d.__class__)
                dd = nd.__dict__
                for key, v in d.__dict__.items():
                    if key == "inv_cov":
                        dd[key] = v
                    else:
                        dd[key] = v.copy() if type(v) is np.ndarray and v.dtype != object else copy.deepcopy(v, memo)
                dists.append(nd)
            ng.__dict__["dists"] = dists
            states.append(ng)
        nm.__dict__["gmm_states"] = states
        out.append(nm)
    return out


class _PickleWriter:
    """The per-iteration pickles of continuous_train (continuous_speech.py:167-170) written on a worker thread while
    the next iteration's alignment runs on the device: the models handed over are not touched again (the next
    iteration trains a copy).  `wait()` before the next hand-over and before returning: every iteration's files exist
    when the next one's are written, errors surface there."""

    def __init__(self):
        self.thread = None
        self.error = None

    def write(self, models, output_path):
        self.wait()
        if os.environ.get("GMMHMM_CTRAIN_PICKLE_THREAD", "1") == "0":      # (A/B switch: write them here and now)
            self._dump(models, output_path)
            return

        def run():
            try:
                self._dump(models, output_path)
            except BaseException as e:
                self.error = e

        self.thread = threading.Thread(target=run, name="gmmhmm-pickles", daemon=True)
        self.thread.start()

    @staticmethod
    def _dump(models, output_path):
        for i, m in enumerate(models):
            # (written under a private name and renamed: the ranks of a sharded run may share `output_path`, and
            #  they all hold the same models)
            final = os.path.join(output_path, str(i) + '.pkl')
            tmp = final + '.%d.tmp' % os.getpid()
            with open(tmp, 'wb') as f:
                pickle.dump(m, f)
            os.replace(tmp, final)

    def wait(self):
        if self.thread is not None:
            self.thread.join()
            self.thread = None
        if self.error is not None:
            e, self.error = self.error, None
            raise e


def continuous_train(data: List[np.ndarray], models: List[HMM], label_seqs: List[List[int]], output_path: AnyStr,
                     n_gaussians: int = 4,
                     n_segments: int = 5,
                     max_iteration: int = 1000,
                     reducer=None,
                     compat_cov: bool = False):
    """Embedded Viterbi training (continuous_speech.py:56-179).

    Per outer iteration: forced alignment of every utterance, frames regrouped per visited
    state, every visited state refit (binary-split k-means + EM, in first-visit order --
    numpy's global RNG is consumed in that order), transition costs re-estimated from the
    segment counts, every model pickled to `output_path/<index>.pkl`; stops when all models
    compare equal (mixtures allclose) to the previous iteration's.

    reducer (extension; the reference is single-process): a `parallel.StatsAllReducer` of a process group with more
    than one rank shards the training by utterance -- `data` / `label_seqs` are THIS rank's utterances, alignment and
    regrouping stay local, the refit runs in lock-step over the ranks (one collective per k-means / EM iteration,
    `lockstep.LockstepFitter`), segment / frame counts for the transition costs are all-reduced, and every rank ends
    each iteration with the same models (states are then visited in ascending order; parity with the single-process
    run is statistical).

    compat_cov (extension): the k-means variances of the random partitions are `np.cov(...).diagonal()` on the host, as
    in the reference (kmeans.py:6-12,171-177), instead of the two-pass per-dimension variance on the device -- the same
    number up to the summation order of the BLAS product behind np.cov (parameters agree to ~1e-13, cluster ids are
    identical on every golden); it brings the frames and the refit loop back to the host (round 2's path)."""
    sharded = bool(reducer is not None and getattr(reducer, "enabled", False) and reducer.world_size > 1)
    old_models = models
    new_models = _clone_models(models)
    n_models = len(new_models)
    # The reference keeps the state objects of this first copy in a dict for the whole run
    # (:64-71).  Later iterations train fresh deep copies with the same uuid hash, so its
    # `gmm_data.get(state)` (:149) falls through to GMM.__eq__(current, first copy): a state
    # whose mixture moved since iteration 0 is reported as "No MFCC data" and keeps its
    # transition costs.  `first_copy` reproduces that lookup.
    first_copy = [list(m.gmm_states) for m in new_models]
    n = len(new_models[0].gmm_states)

    # feature dimension from the MODELS: a rank of a sharded run may hold no utterances at all, and it still has to take
    # part in every collective with buffers of the same shape as everybody else's
    dim = len(np.asarray(models[0].gmm_states[0].dists[0].mean).reshape(-1))
    ctx = _hip.default_context()
    frames = _hip.Batch(ctx, data) if len(data) else _hip.Batch(ctx, feats=np.zeros((0, dim)), offsets=[0])
    # The refit of the states runs as a device-resident session (lockstep.LockstepFitter / gh_fit_*) on frames gathered
    # on the device from the resident utterance batch: no frame visits the host.  compat_cov (np.cov for the partition
    # variances, the host loop of round 2), one feature dimension, the test double of the binding, or a process group
    # that is not the library's own communicator keep the frames on the host as well.
    kmax = 2 ** max(1, int(np.log(n_gaussians)))
    native = bool(getattr(reducer, "native", False))
    on_device = (not compat_cov and 2 <= dim <= 64 and kmax <= 32 and getattr(getattr(_hip, "FitSession", None), "available", False)
                 and frames.np_dtype == np.float64 and (not sharded or native))
    all_frames = None
    if not on_device:
        all_frames = np.concatenate([np.asarray(x, dtype=np.float64) for x in data]) if len(data) else np.zeros((0, dim))
    plan = AlignmentPlan(label_seqs, n, n_models)
    use_runs = os.environ.get("GMMHMM_CTRAIN_RUNS", "1") != "0"      # (0: one label per frame comes back, regrouped with numpy)
    writer = _PickleWriter()
    n_splits = int(np.log(n_gaussians))
    ahead = on_device and n_splits > 0 and os.environ.get("GMMHMM_CTRAIN_AHEAD", "1") != "0" and fast_partition_ok()
    stream = None
    try:
        for it in range(max_iteration):
            print('=' * 25)
            print('Continuous training iteration:', it)
            print('Building state sequences')
            print('Rearranging data, this may take a while...')
            # the refit's random partitions are drawn on a worker thread while the device aligns (lockstep.PartitionStream)
            stream = PartitionStream(int(frames.N) * n_splits) if ahead and frames.N else None
            # alignment + regrouping on the device; per state, its frames in utterance / time order -- what the
            # reference's vstack of the segments holds (:90-113) -- and the states in first-visit order
            runs = aligned_runs(frames, new_models, label_seqs, plan) if (on_device and len(data) and use_runs) else None
            if runs is None:
                use_runs = False
            run_source = None
            if runs is not None:
                # The alignment came back as ~N / 20 runs (state, first row, rows) in utterance / time order instead of one
                # label per frame: the regrouping below is the same arithmetic on the runs -- a stable sort of 70 000 run
                # states instead of 1.4 M frame labels -- and the frames of a run are gathered as one contiguous copy.
                n_states = n_models * n
                rs, rstart, rlen = runs["state"].astype(np.int64), runs["start"], runs["length"]
                per_state = np.bincount(rs, weights=rlen, minlength=n_states).astype(np.int64)
                n_runs = np.bincount(rs, minlength=n_states)
                uniq = np.flatnonzero(per_state)
                order = np.argsort(rs.astype(np.uint16) if n_states < 65536 else rs, kind="stable")   # by state, time order kept
                cuts_r = np.concatenate([[0], np.cumsum(n_runs[uniq])])
                first = rstart[order[cuts_r[:-1]]] if len(uniq) else np.zeros(0, dtype=np.int64)
                runs_of = {int(sid): order[cuts_r[i]:cuts_r[i + 1]] for i, sid in enumerate(uniq)}
                n_of = {int(sid): int(per_state[sid]) for sid in uniq}
                rows_of = {}
                frame_state = seg_start = None
            else:
                if len(data):
                    frame_state, seg_start = aligned_frame_states(frames, new_models, label_seqs, plan)
                else:       # nothing to align on this rank: it only contributes zeros to the collectives below
                    frame_state, seg_start = np.zeros(0, dtype=np.int32), np.zeros(0, dtype=bool)
            used = np.flatnonzero(frame_state >= 0) if runs is None else None
            if runs is None:
                sid_of = frame_state[used]
                # stable grouping by state: 16-bit keys take numpy's radix sort (a comparison sort of 1.4 M int32 keys plus
                # np.unique's second sort were 30 ms of every outer iteration); the first occurrence of a state is the first
                # entry of its group because the sort is stable
                key = sid_of.astype(np.uint16) if n_models * n < 65536 else sid_of
                by_state = np.argsort(key, kind="stable")
                per_state = np.bincount(sid_of, minlength=n_models * n)
                uniq = np.flatnonzero(per_state)
                n_frames = per_state[uniq]
                cuts = np.concatenate([[0], np.cumsum(n_frames)])
                first = by_state[cuts[:-1]] if len(uniq) else np.zeros(0, dtype=np.int64)
                n_runs = np.bincount(frame_state[seg_start], minlength=n_models * n)
                ordered = used[by_state]          # ONE gather of the row indices; a state's rows are a slice of it
                rows_of = {int(sid): ordered[cuts[i]:cuts[i + 1]] for i, sid in enumerate(uniq)}
                n_of = {int(sid): int(c) for sid, c in zip(uniq, n_frames)}
            print('Complete data rearrangement')
            print("=" * 25)

            print('Doing HMM training...')
            # every visited state refit in lock-step (one launch per k-means / EM iteration for all of them); the
            # states are taken in first-visit order, the order in which the reference consumes numpy's global RNG
            keys = [int(sid) for sid in uniq[np.argsort(first, kind="stable")]]
            seg_counts = {sid: (int(n_runs[sid]), n_of[sid]) for sid in keys}
            if sharded:
                # which states were visited anywhere, and their segment / frame counts over all ranks
                loc = np.zeros((n_models * n, 2))
                for sid in keys:
                    loc[sid] = n_of[sid], n_runs[sid]
                glob = reducer(loc)
                keys = [sid for sid in range(n_models * n) if glob[sid, 0] > 0]
                seg_counts = {sid: (glob[sid, 1], glob[sid, 0]) for sid in keys}
            lengths = [n_of.get(sid, 0) for sid in keys]
            if runs is not None:
                ridx = (np.concatenate([runs_of.get(sid, np.zeros(0, dtype=np.int64)) for sid in keys]) if keys
                        else np.zeros(0, np.int64))
                rl = rlen[ridx]
                rows = ("runs", rstart[ridx], rl, np.cumsum(rl) - rl, int(rl.sum()))
            else:
                rows = (np.concatenate([rows_of.get(sid, np.zeros(0, dtype=np.int64)) for sid in keys]) if keys
                        else np.zeros(0, np.int64))
            segs = None if on_device else [all_frames[rows_of[sid]] if sid in rows_of else np.zeros((0, dim)) for sid in keys]
            fitter = LockstepFitter(segs, ctx=frames.ctx, reducer=reducer if sharded else None, source=(frames, rows),
                                    lengths=lengths, dim=dim, kmax=kmax, compat_cov=compat_cov)
            parts = None
            if stream is not None:
                parts, stream = stream.take(lengths, n_splits), None
            try:
                # start centroids: the mean of every state's frames (:116), over all ranks when sharded
                if sharded and keys:
                    if fitter.fit is not None:
                        sums = fitter.fit.segment_means()[0]           # per state, this rank's frames summed on the device
                    else:
                        sums = np.array([seg.sum(axis=0) for seg in fitter.segs])
                    tot = reducer(np.ascontiguousarray(sums))
                    starts = tot / np.array([seg_counts[sid][1] for sid in keys], dtype=np.float64)[:, None]
                else:
                    starts = fitter.segment_means() if keys else np.zeros((0, dim))
                fitter.split_and_fit([new_models[sid // n].gmm_states[sid % n] for sid in keys],
                                     start_centroids=starts,
                                     weight_divisor=[n_segments] * len(keys),          # (:127, :135-137)
                                     n_gaussians=n_gaussians, use_em=True, parts=parts)
            finally:
                fitter.close()

            print('Updating other model parameters...')
            for mi in range(n_models):
                for si in range(n):
                    cnt = seg_counts.get(mi * n + si)
                    current = new_models[mi].gmm_states[si]
                    if cnt is not None and current is not first_copy[mi][si] and not (current == first_copy[mi][si]):
                        cnt = None
                    if cnt is None:
                        warnings.warn("No MFCC data for state", UserWarning)
                        continue
                    p_jump = cnt[0] / cnt[1]
                    if si < n - 1:
                        new_models[mi].transitions[si + 1, si] = -np.log(p_jump)
                    new_models[mi].transitions[si, si] = -np.log(1 - p_jump)

            # (:167-170) this iteration's pickles: on the worker, while the comparison below and the next alignment run
            writer.write(new_models, output_path)
            if all(new_m == old_m for new_m, old_m in zip(new_models, old_models)):
                print('Continuous training converged')
                break
            old_models = new_models
            new_models = _clone_models(old_models)
    finally:
        if stream is not None:
            stream.cancel()
        try:
            writer.wait()
        finally:
            plan.close()
            frames.close()
