# -*- coding: utf-8 -*-
"""Segmental k-means, plain k-means and re-alignment on the MI355X.

Mirror of the reference's `sr/recognition/kmeans.py` (same public names and
return values).  The inner loops -- the N x k distance/arg-min sweep of `kmeans`
and the per-template DP of `skmeans` / `align_gmm_states` -- run in HIP
(gh_kmeans_assign, gh_dtw, gh_loglik + gh_viterbi) over frames that stay resident
on the GPU for the whole iteration loop; segment bookkeeping stays on the host.
"""
import numpy as np

from . import _hip
from .decode import dtw_batch, decode_batch
from .hmm_state import mahalanobis, euclidean

__all__ = ["calc_variance", "combine_templates", "segment_data", "calc_transition_costs",
           "get_segments_from_path", "skmeans", "cluster_centroids", "kmeans", "align_gmm_states"]


def _ctx():
    return _hip.default_context()


def calc_variance(data):
    """Per-row sample variance (ddof=1) of a [D, N] array: the diagonal of np.cov (kmeans.py:6-12)."""
    return np.cov(data).diagonal()


def segment_data(templates, n_temps, n_segments, seg_starts):
    """Frames of segment s gathered over all templates (kmeans.py:33-50).

    seg_starts[r, s] is the first frame of segment s in template r; the last segment
    runs to the end of the template."""
    out = []
    for s in range(n_segments):
        pieces = []
        for r in range(n_temps):
            lo = seg_starts[r, s]
            hi = None if s == n_segments - 1 else seg_starts[r, s + 1]
            pieces.append(np.asarray(templates[r])[lo:hi])
        if sum(len(p) for p in pieces):
            out.append(np.concatenate(pieces, axis=0).astype(np.float64, copy=False))
        else:
            out.append(np.array([]))
    return out


def combine_templates(templates, n_temps, n_segments, seg_starts):
    """Mean and variance of every segment (kmeans.py:15-30)."""
    dim = templates[0].shape[1]
    res = np.zeros((n_segments, dim))
    vars = np.zeros((n_segments, dim))
    for s, seg in enumerate(segment_data(templates, n_temps, n_segments, seg_starts)):
        res[s] = seg.mean(axis=0)
        vars[s] = calc_variance(seg.T)
    return res, vars


def calc_transition_costs(n_temps, seg_lens, max_jump_dist=2):
    """Left-to-right transition costs from segment lengths (kmeans.py:53-95).

    Each template leaves segment i exactly once: p_jump = n_temps / (frames in segment i),
    cost[i+jump, i] = -log p_jump, cost[i, i] = -log(1 - p_jump); the jump skips segments
    that are empty in some template, by at most `max_jump_dist`; the last segment never
    jumps.  cost[i, j] is the cost of going from j to i."""
    n_segments = seg_lens.shape[1]
    has_empty = (seg_lens == 0).sum(axis=0) != 0
    res = np.full((n_segments, n_segments), np.inf)
    for i in range(n_segments):
        n_jump = 0 if i == n_segments - 1 else n_temps
        jump = 1
        s = i + 1
        while s < n_segments - 1 and has_empty[s + 1]:
            jump += 1
            if jump > max_jump_dist:
                break
            s += 1
        n_all = 0
        for t in range(n_temps):
            n_all += seg_lens[t, i]
        p_stay = (n_all - n_jump) / n_all
        p_jump = n_jump / n_all
        if n_jump:
            res[i + jump, i] = -np.log(p_jump)
        res[i, i] = -np.log(p_stay)
    return res


def get_segments_from_path(path, n_segments):
    """Segment start offsets from an alignment path: cumulative visit counts of rows
    0..n_segments-2 (kmeans.py:98-108)."""
    counts = np.zeros(n_segments, dtype=np.int64)
    rows, cnt = np.unique(np.asarray(path)[:, 0], return_counts=True)
    counts[rows] = cnt
    return np.add.accumulate(counts)[:-1]


def skmeans(templates, n_segments, dist_fun=euclidean, return_segmented_data=False, max_iteration=1000):
    """Segmental k-means (kmeans.py:111-155).

    :param templates: list of [T_r, D] arrays
    :param n_segments: number of segments (HMM states)
    :param dist_fun: frame distance for the alignment DP (default Euclidean)
    :return: (means [n,D], variances [n,D], transition costs [n,n][, segmented data])

    All templates are uploaded once; every iteration is ONE batched gh_dtw launch (one wave
    per template).  As in the reference the transition costs are those of the initial
    uniform segmentation for the whole loop (`seg_lens` is never refreshed, :139)."""
    assert max_iteration > 0
    n_temps = len(templates)
    seg_lens = np.zeros((n_temps, n_segments + 1), dtype=np.int64)
    for r in range(n_temps):
        seg_lens[r, 1:] = len(templates[r]) // n_segments
    seg_starts = np.add.accumulate(seg_lens, axis=1)[:, :-1]
    seg_lens = seg_lens[:, 1:]

    transition_costs = None
    res, vars = combine_templates(templates, n_temps, n_segments, seg_starts)
    frames = None
    try:
        for _ in range(max_iteration):
            seg_starts = np.zeros((n_temps, n_segments), dtype=np.int64)
            transition_costs = calc_transition_costs(n_temps, seg_lens)
            for r in range(n_temps):
                if templates[r].shape[0] < 5:
                    raise NameError('template is too small, cannot do dtw on it')
            if frames is None:
                frames = _hip.Batch(_ctx(), templates)
            _, paths = dtw_batch(templates, res, dist_fun, transition_costs, want_costs=False, batch=frames)
            for r in range(n_temps):
                seg_starts[r, 1:] = get_segments_from_path(paths[r], n_segments)
            new_res, vars = combine_templates(templates, n_temps, n_segments, seg_starts)
            if np.allclose(res, new_res):
                break
            res = new_res
    finally:
        if frames is not None:
            frames.close()
    if return_segmented_data:
        return res, vars, transition_costs, segment_data(templates, n_temps, n_segments, seg_starts)
    return res, vars, transition_costs


def cluster_centroids(data, clusters, k):
    """Mean of the frames assigned to each cluster (kmeans.py:158-164); an empty cluster
    gives a NaN row (and numpy's RuntimeWarning), as in the reference."""
    out = np.empty(shape=(k,) + data.shape[1:])
    for i in range(k):
        np.mean(data[clusters == i, :], axis=0, out=out[i])
    return out


def kmeans(data, k, centroids, dist_fun=euclidean, max_iteration=1000):
    """k-means as the reference runs it (kmeans.py:167-193).

    A random partition drawn from numpy's GLOBAL generator (seed it for reproducibility)
    only provides the per-cluster variances `cov`, which are returned untouched; every
    distance is taken under cov[0]; the loop ends when the centroids stop changing
    bit for bit.  The frames are uploaded once and each iteration's N x k distance +
    arg-min sweep is one gh_kmeans_assign launch.

    :return: (clusters [N] int64, centroids [k,D], cov [k,D])"""
    assert k == centroids.shape[0]
    data = np.ascontiguousarray(data, dtype=np.float64)
    n = data.shape[0]
    clusters = np.random.randint(0, k, n)
    cov = np.array([calc_variance(data[clusters == c].T) for c in range(k)])
    frames = None
    try:
        if dist_fun is euclidean or dist_fun is mahalanobis:
            frames = _hip.Batch(_ctx(), feats=data, offsets=[0, n])
        for _ in range(max(max_iteration, 1)):
            if dist_fun is mahalanobis:
                clusters = frames.kmeans_assign(centroids, var=cov[0])
            elif dist_fun is euclidean:
                clusters = frames.kmeans_assign(centroids)
            else:  # arbitrary callable: scored on the host, cell by cell, like the reference
                dists = np.zeros((n, k))
                for i in range(n):
                    for c in range(k):
                        dists[i, c] = dist_fun(centroids[c, :], data[i, :], cov[0])
                clusters = np.argmin(dists, axis=1)
            new_centroids = cluster_centroids(data, clusters, k)
            if np.array_equal(new_centroids, centroids):
                break
            centroids = new_centroids
    finally:
        if frames is not None:
            frames.close()
    return clusters, centroids, cov


def align_gmm_states(templates, gmm_states, transition_costs, n_segments):
    """Re-segment every template by Viterbi alignment against the trained mixtures
    (kmeans.py:196-205): one batched gh_loglik + gh_viterbi launch over all templates."""
    n_temps = len(templates)
    seg_starts = np.zeros((n_temps, n_segments), dtype=np.int64)
    res = decode_batch(templates, gmm_states, transition_costs)
    for r in range(n_temps):
        seg_starts[r, 1:] = get_segments_from_path(res["paths"][r], n_segments)
    return segment_data(templates, n_temps, n_segments, seg_starts)
