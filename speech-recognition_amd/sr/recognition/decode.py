# -*- coding: utf-8 -*-
"""Dynamic programs of the GMM-HMM core on the MI355X.

Mirror of the reference's `sr/recognition/decode.py`: `dtw` (decode.py:7-77) and
`decode_hmm_states` (decode.py:80-146) keep their signatures, return values
(`costs[R,T]` float64, `path[K,2]` int64 from end to start) and exceptions, but
the emission scoring and the DP sweep + back-trace run in HIP kernels
(gh_loglik / gh_viterbi / gh_dtw).  `decode_batch` / `dtw_batch` are the same
operations over many utterances in one launch.
"""
import warnings
from math import isinf

import numpy as np

from . import _hip
from . import _pack
from .hmm_state import mahalanobis, euclidean

__all__ = ["dtw", "decode_hmm_states"]


def _ctx():
    return _hip.default_context()


# ------------------------------------------------------------------------- A6
def decode_batch(xs, states, transitions, end_rows=None, want_costs=False, dtype=np.float64, beam=np.inf):
    """Viterbi of every utterance in `xs` through one state list.

    end_rows: candidate final rows in the last column (default: the last row).
    beam: rank beam per column (see `decode_hmm_states`).
    Returns the dict of `_hip.Lattices.viterbi` (paths, best_end, end_cost[, costs])."""
    ctx = _ctx()
    row_state, uniq = _pack.pack_states(states)
    R = len(states)
    if end_rows is None:
        end_rows = [R - 1]
    graph = _pack.graph_from_dense(row_state, transitions, [0], end_rows)
    pruned = not (beam is None or isinf(beam) or beam <= 0)
    # (a pruned decode gets a graph handle of its own: the beam is a property of the handle, and cached handles are shared)
    lat = _hip.Lattices(ctx, [graph]) if pruned else _pack.device_lattices(ctx, [graph])
    if pruned:
        lat.set_beam(beam)
    batch = _hip.Batch(ctx, xs, dtype=dtype)
    try:
        if uniq:
            gmm = _pack.device_gmm(ctx, uniq)
            if gmm.D != batch.D and batch.N:
                raise NameError("The dimensions of the input don't match")  # hmm_state.py:45
            batch.loglik(gmm, fetch=False)
        else:  # a graph of non-emitting rows only still needs a (never read) likelihood matrix
            dummy = _hip.PackedGMM(ctx, np.zeros((1, 1, batch.D)), np.ones((1, 1, batch.D)), np.ones((1, 1)))
            batch.loglik(dummy, fetch=False)
        return lat.viterbi(batch, want_path=True, want_costs=want_costs)
    finally:
        batch.close()
        if pruned:
            lat.close()


def decode_hmm_states(x, states, transitions, end_points=None, beam=np.inf):
    """
    :param x: an input, array [T, D].
    :param states: a list of hmm states (GMM / NES objects).
    :param transitions: transition matrix, `transitions[i,j]` = cost of going from the jth to the ith state
        (+inf = no arc).
    :param end_points: a list of cells `[r, c]` that may end the state sequence (default: last row, last column).
    :param beam: EXTENSION (the reference's decode_hmm_states has no pruning; its `dtw` has, decode.py:62-68): after
        every column but the last, all but the `beam` cheapest cells of the column -- ranked by (cost, row) -- read
        +inf as origins of the next column and show as +inf in `costs`.  The default (inf) is the reference's decode.
    :return: costs: cost matrix [R, T].
            path: reversed path (from end to start) `[[r_n, c_n], ..., [r_1, c_1]]`, end cell excluded.
    """
    x = np.asarray(x, dtype=np.float64)
    T, R = len(x), len(states)
    if end_points is None:
        end_points = [[R - 1, T - 1]]
    # the reference accepts any cell as end point; its callers only use the last column
    # (hmm.py:132, main.py:60, continuous_speech.py:89)
    last = T - 1
    norm = [(r, c + T if c < 0 else c) for r, c in end_points]
    last_rows = [r for r, c in norm if c == last]
    full = decode_batch([x], states, transitions, end_rows=last_rows or [R - 1], want_costs=True, beam=beam)
    costs = full["costs"][0]
    # choose the end cell like decode.py:129-134 ('>=': the last minimum wins)
    best_cost, best = np.inf, None
    for (r, c), e in zip(norm, end_points):
        if best_cost >= costs[e[0], e[1]]:
            best_cost, best = costs[e[0], e[1]], (r, c)
    if isinf(best_cost):
        warnings.warn("decode_hmm_states: Cannot find a path when decoding sequence")
    br, bc = best
    if bc == 0:
        return costs, np.array([])
    if bc == last and last_rows and last_rows[full["best_end"][0]] == br:
        path = full["paths"][0]
    else:  # an inner column (or a tie resolved across columns): back-trace that prefix on its own
        path = decode_batch([x[:bc + 1]], states, transitions, end_rows=[br], beam=beam)["paths"][0]
    if len(path) == 0:
        return costs, np.array([])
    return costs, path


# ------------------------------------------------------------------------- A5
def dtw_batch(xs, y, dist_fun, transitions, variance=None, beam=np.inf, want_costs=True, batch=None):
    """`dtw` for a list of inputs against one template; one wave per input.
    `batch` may carry an already resident `_hip.Batch` of `xs`."""
    ctx = _ctx()
    y = np.asarray(y, dtype=np.float64)
    transitions = np.asarray(transitions, dtype=np.float64)
    for x in xs:
        assert len(x) > 1 and len(y) > 1  # decode.py:22
    own = batch is None
    if own:
        batch = _hip.Batch(ctx, xs)
    try:
        b = 0 if isinf(beam) else int(beam)
        if dist_fun is euclidean and variance is None:
            return batch.dtw(transitions, y=y, beam=b, want_costs=want_costs)
        if dist_fun is mahalanobis and variance is not None:
            return batch.dtw(transitions, y=y, var=np.asarray(variance, dtype=np.float64), beam=b,
                             want_costs=want_costs)
        # any other callable: the caller's function scores the cells, the DP stays on the GPU
        dist = []
        for x in xs:
            d = np.empty((len(y), len(x)))
            for i in range(len(y)):
                for j in range(len(x)):
                    d[i, j] = dist_fun(x[j], y[i]) if variance is None else dist_fun(x[j], y[i], variance[i])
            dist.append(d)
        return batch.dtw(transitions, beam=b, dist=dist, want_costs=want_costs)
    finally:
        if own:
            batch.close()


def dtw(x, y, dist_fun, transitions, variance=None, beam=np.inf):
    """
    :param x: an input, array [T, D].
    :param y: a template, array [n, D].
    :param dist_fun: distance between two frames; `mahalanobis` and the default Euclidean
        distance of `skmeans`/`kmeans` run fused in HIP, any other callable is evaluated per cell.
    :param transitions: transition matrix, transitions[i,j] = cost of going from the jth to the ith row.
    :param variance: per-row variance handed to `dist_fun` as third argument.
    :param beam: beam size of pruning.
    :return: costs: cost matrix [n, T]; path: reversed path `[[row, col], ...]` from the cell before the
        end back to [0, 0].
    """
    costs, paths = dtw_batch([np.asarray(x, dtype=np.float64)], y, dist_fun, transitions, variance, beam)
    return costs[0], paths[0]
