# -*- coding: utf-8 -*-
"""Segmental k-means, plain k-means and re-alignment on the MI355X.

Mirror of the reference's `sr/recognition/kmeans.py` (same public names and
return values).  The inner loops -- the N x k distance/arg-min sweep of `kmeans`
and the per-template DP of `skmeans` / `align_gmm_states` -- run in HIP
(gh_kmeans_assign, gh_dtw, gh_loglik + gh_viterbi) over frames that stay resident
on the GPU for the whole iteration loop.

Segmental k-means runs for MANY word models at once (`skmeans_multi`; the reference trains
word after word, sr/core.py:57-60): all templates of all words in one resident batch, per
iteration ONE alignment launch (every template against its own word's segment means,
gh_fit_dtw) and one device reduction keyed on (word, segment) for the new means and variances
(gh_fit_group_stats); the alignment never leaves the device -- what comes back per iteration
is [W, n, D] means, and a word drops out of the launches when its means stop moving.
`skmeans` is the one-word case of it.
"""
import numpy as np

from . import _hip
from .decode import dtw_batch, decode_batch
from .hmm_state import mahalanobis, euclidean

__all__ = ["calc_variance", "combine_templates", "segment_data", "segment_data_fast", "calc_transition_costs",
           "get_segments_from_path", "skmeans", "skmeans_multi", "cluster_centroids", "kmeans", "align_gmm_states"]


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


def segment_data_fast(templates, n_segments, seg_starts):
    """segment_data for MANY templates: the same arrays (frames of segment s over all templates, template after template),
    from one concatenation and n masks instead of n x n_temps slices (2 000 templates: 28 -> ~3 ms)."""
    R = len(templates)
    if R == 0:
        return [np.array([]) for _ in range(n_segments)]
    lengths = np.array([len(t) for t in templates], dtype=np.int64)
    X = np.concatenate([np.asarray(t) for t in templates], axis=0).astype(np.float64, copy=False)
    tpl = np.repeat(np.arange(R), lengths)
    t_in = np.arange(int(lengths.sum())) - np.repeat(np.cumsum(lengths) - lengths, lengths)
    starts = np.asarray(seg_starts, dtype=np.int64)
    # frame t of template r sits in segment s iff starts[r, s] <= t < starts[r, s + 1] (the last one runs to the end);
    # like the slices of segment_data, a frame may be claimed by no segment or -- starts not monotone -- by several
    out = []
    for s in range(n_segments):
        m = t_in >= starts[tpl, s]
        if s < n_segments - 1:
            m &= t_in < starts[tpl, s + 1]
        out.append(X[m] if m.any() else np.array([]))
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
    """Left-to-right transition costs from segment lengths (kmeans.py:53-95); cost[i, j] = cost of going from j to i.

    Every template leaves segment i exactly once, so with F_i frames in segment i over all templates
    p_jump = n_temps / F_i, cost[i + jump, i] = -log p_jump, cost[i, i] = -log(1 - p_jump); the last segment never
    jumps.  The jump is 1 unless segments behind i + 1 are empty in some template: it then grows by one per such
    segment (checked at i + 2, i + 3, ... while that index is not the last segment), and an attempt to go beyond
    `max_jump_dist` leaves the oversized value standing (the reference's loop increments before it tests)."""
    seg_lens = np.asarray(seg_lens)
    n = seg_lens.shape[1]
    frames = seg_lens.sum(axis=0)                               # F_i
    some_empty = (seg_lens == 0).any(axis=0)
    cost = np.full((n, n), np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        p_jump = np.where(np.arange(n) == n - 1, 0, n_temps) / frames
        stay = -np.log((frames - np.where(np.arange(n) == n - 1, 0, n_temps)) / frames)
        leave = -np.log(p_jump)
    for i in range(n):
        cost[i, i] = stay[i]
        if i == n - 1:
            continue
        jump, probe = 1, i + 1
        while probe < n - 1 and some_empty[probe + 1]:
            jump += 1
            if jump > max_jump_dist:
                break
            probe += 1
        cost[i + jump, i] = leave[i]
    return cost


def get_segments_from_path(path, n_segments):
    """Segment start offsets from an alignment path: cumulative visit counts of rows
    0..n_segments-2 (kmeans.py:98-108)."""
    counts = np.zeros(n_segments, dtype=np.int64)
    rows, cnt = np.unique(np.asarray(path)[:, 0], return_counts=True)
    counts[rows] = cnt
    return np.add.accumulate(counts)[:-1]


def _uniform_segments(lengths, n_segments):
    """The initial segmentation of kmeans.py:122-127: n equal pieces of T // n frames, the last one to the end.
    Returns (segment of every frame [sum T], seg_lens [n_temps, n] as the reference holds them: T // n each)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    q = lengths // n_segments
    # runs of q frames for the segments 0 .. n-2, the rest of the template for the last one (a template shorter than n: all last)
    run_len = np.repeat(q[:, None], n_segments, axis=1)
    run_len[:, -1] = lengths - q * (n_segments - 1)
    ids = np.repeat(np.tile(np.arange(n_segments, dtype=np.int32), len(lengths)), run_len.reshape(-1))
    return ids, np.repeat(q[:, None], n_segments, axis=1)


def _starts_from_ids(ids, lengths, n_segments):
    """seg_starts [n_temps, n]: frames of the template aligned to an earlier segment (what get_segments_from_path counts
    on the path, kmeans.py:98-108)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    tpl = np.repeat(np.arange(len(lengths)), lengths)
    counts = np.bincount(tpl * n_segments + ids, minlength=len(lengths) * n_segments).reshape(len(lengths), n_segments)
    # (the path holds the columns 0 .. T-2: the last frame is not counted -- it sits in the last segment, whose count is
    #  not part of any start)
    starts = np.zeros((len(lengths), n_segments), dtype=np.int64)
    np.cumsum(counts[:, :-1], axis=1, out=starts[:, 1:])
    return starts


def segment_order(lengths, n_temps, seg_starts, n_segments):
    """The frames of MANY words' templates (back to back, word after word, template after template) regrouped the way
    `segment_data` regroups them (kmeans.py:33-50): -> (order [N]: frame numbers grouped by (word, segment), inside a group
    template after template in time order; counts [W, n]: frames per group).  seg_starts [R, n] as `skmeans` holds them
    (cumulative visit counts: monotone), the last segment runs to the end of its template."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n_temps = np.asarray(n_temps, dtype=np.int64)
    R, N, W, n = len(lengths), int(np.sum(lengths)), len(n_temps), n_segments
    starts = np.asarray(seg_starts, dtype=np.int64).reshape(R, n)
    # a template's frames of one segment are ONE run [start of the segment, start of the next one): R x n runs, put in
    # (word, segment, template) order and expanded -- no per-frame key, no sort over the frames
    begin = np.minimum(starts, lengths[:, None])
    run_len = np.empty((R, n), dtype=np.int64)
    run_len[:, :-1] = begin[:, 1:] - begin[:, :-1]
    run_len[:, -1] = lengths - begin[:, -1]
    group = np.repeat(np.arange(W), n_temps)[:, None] * n + np.arange(n)[None, :]       # [R, n]: (word, segment) of a run
    perm = np.argsort(group.reshape(-1), kind="stable")                                  # templates stay in order inside a group
    rl = run_len.reshape(-1)[perm]
    rb = ((np.cumsum(lengths) - lengths)[:, None] + begin).reshape(-1)[perm]
    order = np.repeat(rb - (np.cumsum(rl) - rl), rl) + np.arange(N)
    counts = np.bincount(group.reshape(-1), weights=run_len.reshape(-1), minlength=W * n).astype(np.int64).reshape(W, n)
    return order, counts


_host_ws = {"buf": None, "busy": False}


def host_workspace(shape, dtype=np.float64):
    """-> (array of `shape`, release()): scratch for a call's own copy of the caller's frames.  The buffer is kept between
    calls (a fresh 62 MB array costs 5 ms of page faults when it is filled and 3 ms when it is freed -- a quarter of
    `train_words` on ten words x 200 templates); it grows to the largest request up to GMMHMM_HOST_WORKSPACE_MB
    (default 1024; 0: never kept).  A request while another call holds it, or beyond the cap, gets an ordinary array."""
    import os
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    cap = int(os.environ.get("GMMHMM_HOST_WORKSPACE_MB", "1024")) << 20
    if _host_ws["busy"] or nbytes > cap or nbytes == 0:
        return np.empty(shape, dtype=dtype), lambda: None
    _host_ws["busy"] = True                                         # (set under the GIL: no other thread sees it free)
    buf = _host_ws["buf"]
    if buf is None or buf.nbytes < nbytes:
        _host_ws["buf"] = buf = None                                # (the old one goes before the new one comes)
        _host_ws["buf"] = buf = np.empty(nbytes, dtype=np.uint8)

    def release():
        _host_ws["busy"] = False
    return buf[:nbytes].view(dtype).reshape(shape), release


concat_rows = _hip.concat_rows


def gather_rows(X, order):
    """X[order] for a big row-major matrix, the copy spread over a few threads (np.take releases the GIL;
    GMMHMM_HOST_THREADS, default 8): the 62 MB of the segments of ten words x 200 templates take 4 ms instead of 20 -- 2 ms
    of copying and 2-3.5 ms for the kernel to hand out 62 MB of zeroed pages, which threads do not speed up
    (`tools/host_copy_probe.py`)."""
    import os
    import threading
    N = len(order)
    out = np.empty((N,) + X.shape[1:], dtype=X.dtype)
    n_thr = min(int(os.environ.get("GMMHMM_HOST_THREADS", "8")), os.cpu_count() or 1) if N * X.shape[1] >= (1 << 20) else 1
    if n_thr <= 1:
        np.take(X, order, axis=0, out=out)
        return out
    cuts = np.linspace(0, N, n_thr + 1).astype(np.int64)
    thr = [threading.Thread(target=np.take, args=(X, order[cuts[i]:cuts[i + 1]]), kwargs=dict(axis=0, out=out[cuts[i]:cuts[i + 1]]))
           for i in range(n_thr)]
    for t in thr:
        t.start()
    for t in thr:
        t.join()
    return out


def split_segments(Xo, counts):
    """The regrouped frames as `segment_data` returns them: per word a list of n arrays (np.array([]) for an empty one)."""
    W, n = counts.shape
    cuts = np.concatenate([[0], np.cumsum(counts.reshape(-1))])
    return [[Xo[cuts[w * n + s]:cuts[w * n + s + 1]] if counts[w, s] else np.array([]) for s in range(n)] for w in range(W)]


def skmeans_multi(templates_by_word, n_segments, max_iteration=1000, frames=None):
    """Segmental k-means (kmeans.py:111-155, Euclidean frame distance) of every word model at once.

    templates_by_word: list over words of lists of [T_r, D] arrays.  Returns a list over words of
    (means [n,D], variances [n,D], transition costs [n,n], seg_starts [n_temps, n]) -- per word exactly what `skmeans`
    computes for it alone: the words only share the launches.  As in the reference the transition costs are those of
    the initial uniform segmentation for the whole loop (`seg_lens` is never refreshed, :139), the variances are the
    last ones computed, and on convergence the means of the PREVIOUS iteration are returned (:146-148).
    frames: the templates as a resident fp64 batch (word after word, template after template), when the caller holds one."""
    assert max_iteration > 0
    W = len(templates_by_word)
    n_temps = np.array([len(ts) for ts in templates_by_word], dtype=np.int64)
    lengths = np.array([len(t) for ts in templates_by_word for t in ts], dtype=np.int64)
    if np.any(lengths < 5):
        raise NameError('template is too small, cannot do dtw on it')
    utt_word = np.repeat(np.arange(W), n_temps).astype(np.int32)
    ctx = _ctx()
    own = frames is None
    if own:
        frames = _hip.Batch(ctx, [np.asarray(t, dtype=np.float64) for ts in templates_by_word for t in ts])
    tpl_off = np.concatenate([[0], np.cumsum(n_temps)])
    word_off = frames.offsets[tpl_off]                      # a word's templates sit back to back in the batch
    ids, seg_lens = _uniform_segments(lengths, n_segments)
    fit = _hip.FitSession(ctx, frames, word_off, max(n_segments, 2))
    try:
        fit.set_ids(ids)
        res, vars_, _ = fit.group_stats(n_segments)
        trans = np.array([calc_transition_costs(int(n_temps[w]), seg_lens[tpl_off[w]:tpl_off[w + 1]]) for w in range(W)])
        active = np.ones(W, dtype=np.uint8)
        for _ in range(max_iteration):
            fit.dtw(n_segments, res, trans, utt_word, active)
            new_res, new_vars, _cnt = fit.group_stats(n_segments, active)
            for w in np.flatnonzero(active):
                vars_[w] = new_vars[w]
                if np.allclose(res[w], new_res[w]):
                    active[w] = 0
                else:
                    res[w] = new_res[w]
            if not active.any():
                break
        ids = fit.clusters()
    finally:
        fit.close()
        if own:
            frames.close()
    starts = _starts_from_ids(ids, lengths, n_segments)
    return [(res[w], vars_[w], trans[w], starts[tpl_off[w]:tpl_off[w + 1]]) for w in range(W)]


def _device_skmeans_possible(templates):
    fs = getattr(_hip, "FitSession", None)
    return bool(getattr(fs, "available", False)) and len(templates) > 0 and 2 <= np.asarray(templates[0]).shape[1] <= 64


def skmeans(templates, n_segments, dist_fun=euclidean, return_segmented_data=False, max_iteration=1000):
    """Segmental k-means (kmeans.py:111-155).

    :param templates: list of [T_r, D] arrays
    :param n_segments: number of segments (HMM states)
    :param dist_fun: frame distance for the alignment DP (default Euclidean)
    :return: (means [n,D], variances [n,D], transition costs [n,n][, segmented data])

    With the default distance this is `skmeans_multi` for one word: alignment and segment statistics on the device.
    Any other `dist_fun` (and one-dimensional features) keeps the alignment launch per iteration with the segment
    bookkeeping in numpy."""
    n_temps = len(templates)
    if dist_fun is euclidean and 2 <= n_segments <= 32 and _device_skmeans_possible(templates):
        res, vars, transition_costs, seg_starts = skmeans_multi([templates], n_segments, max_iteration=max_iteration)[0]
    else:
        res, vars, transition_costs, seg_starts = _skmeans_host(templates, n_segments, dist_fun, max_iteration)
    if return_segmented_data:
        return res, vars, transition_costs, segment_data(templates, n_temps, n_segments, seg_starts)
    return res, vars, transition_costs


def _skmeans_host(templates, n_segments, dist_fun, max_iteration):
    """One batched alignment launch per iteration, means / variances / starts in numpy (any distance function)."""
    assert max_iteration > 0
    n_temps = len(templates)
    lengths = [len(t) for t in templates]
    if min(lengths) < 5:
        raise NameError('template is too small, cannot do dtw on it')
    _, seg_lens = _uniform_segments(lengths, n_segments)
    seg_starts = np.arange(n_segments)[None, :] * seg_lens[:, :1]
    transition_costs = calc_transition_costs(n_temps, seg_lens)
    res, vars = combine_templates(templates, n_temps, n_segments, seg_starts)
    frames = _hip.Batch(_ctx(), templates)
    try:
        for _ in range(max_iteration):
            _, paths = dtw_batch(templates, res, dist_fun, transition_costs, want_costs=False, batch=frames)
            seg_starts = np.zeros((n_temps, n_segments), dtype=np.int64)
            for r, path in enumerate(paths):
                seg_starts[r, 1:] = get_segments_from_path(path, n_segments)
            new_res, vars = combine_templates(templates, n_temps, n_segments, seg_starts)
            if np.allclose(res, new_res):
                break
            res = new_res
    finally:
        frames.close()
    return res, vars, transition_costs, seg_starts


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
