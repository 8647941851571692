# -*- coding: utf-8 -*-
"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A plain numpy (fp64 / int64) restatement of the GMM-HMM hot path of
tjysdsg/speech-recognition (`sr/recognition/*.py`), written on packed arrays
instead of the reference's object graph.  Every function cites the reference
file:line whose behaviour it follows (paths relative to /root/reference).

Who may import this module: `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` -- as the *checker* / reported CPU baseline
only.  Nothing under `speech-recognition_amd/` imports it; the product path has
no CPU fallback and fails loudly without the HIP library.

Parity pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, captured in this container by `tools/make_goldens.py` and committed as
`tests/golden/*.npz`; `tests/test_oracle_golden.py` checks every function below
against them.  A13 (forward-backward) does not exist in the reference: that one
function is "parity unpinned" and is pinned by brute-force enumeration instead.

Packed conventions used throughout
----------------------------------
* a mixture state  = (means[M,D], vars[M,D], w[M]); weights are NOT renormalised
* a state list     = E[R,T] emission-cost matrix + is_nes[R] bool
  (E[r,t] = cost of row r emitting frame t; 0 for non-emitting rows)
* transitions      = dense [R,R] cost matrix, trans[i,j] = cost of j -> i,
  +inf = no arc (decode.py:12-13)
* paths            = int64 [K,2] rows [row, col], END -> START, exactly as the
  reference returns them
"""
import warnings
from math import isinf

import numpy as np

TWO_PI = 2 * np.pi


# --------------------------------------------------------------------------- A1
def gauss_pdf(x, mean, var, dense_inv=None):
    """Diagonal Gaussian density in the LINEAR domain (hmm_state.py:36-45).

    `dense_inv` (optional [D,D]) reproduces the reference's cost structure (two
    dense dots against inv(diag(var)), hmm_state.py:17,42); without it the
    quadratic form uses the diagonal directly (same value up to summation order).
    """
    d = x.shape[0]
    if d != mean.shape[0]:
        raise NameError("The dimensions of the input don't match")  # hmm_state.py:45
    norm = 1.0 / (np.power(TWO_PI, float(d) / 2) * np.sqrt(np.prod(var)))
    dx = x - mean
    if dense_inv is not None:
        q = dx.dot(dense_inv).dot(dx.T)
    else:
        q = np.sum(dx * (1.0 / var) * dx)
    return norm * np.exp(-0.5 * q)


# --------------------------------------------------------------------------- A2
def mahalanobis(v1, v2, variance):
    """Diagonal-Gaussian negative log-likelihood (hmm_state.py:48-58)."""
    d = len(variance)
    m = v1 - v2
    return 0.5 * np.log(TWO_PI ** d * np.prod(variance)) + 0.5 * np.sum(m / variance * m)


def euclid(v1, v2, *_):
    """Default `dist_fun` of dtw/kmeans/skmeans (kmeans.py:111,167)."""
    return np.linalg.norm(v1 - v2)


# ------------------------------------------------------------------ N3 (front-end)
def delta_feature(feat):
    """core.py:13-22: central difference feat[i+1]-feat[i-1], one-sided at both ends
    (so T == 1 raises IndexError like the reference's feat[i + 1])."""
    T = len(feat)
    if T < 2:
        raise IndexError("index 1 is out of bounds for axis 0 with size %d" % T)
    d = np.zeros(feat.shape)
    d[0] = feat[1] - feat[0]
    d[-1] = feat[-1] - feat[-2]
    d[1:-1] = feat[2:] - feat[:-2]
    return d


def mfcc_features_signal(signal, sample_rate, frame_size=0.025, frame_stride=0.01, low_freq=80, high_freq=None):
    """mfcc_features (sr/feature/feature.py:43-82) on an in-memory signal (the reference reads a wav file, :44):
    pre-emphasis :45-46, segment :7-22, zero_padding :25-40, Hamming over the padded frame :52, power
    spectrum of a 512-point rfft :54-56, 40 mel filters :58-75, log10 with eps :76-78, ortho DCT-II 1..13 :80-81.
    Returns (filter_banks [T,40], mfcc [T,13])."""
    from scipy.fftpack import dct
    signal = np.asarray(signal)
    emphasized = np.append(signal[0], signal[1:] - 0.97 * signal[:-1])
    flen, fstep = int(frame_size * sample_rate), int(frame_stride * sample_rate)
    num_frames = int(np.ceil(emphasized.size / fstep))          # `slen > frame_len` (seconds!) is always true
    final_len = (num_frames - 1) * fstep + flen
    pad_sig = np.concatenate([emphasized, np.zeros(max(0, final_len - emphasized.size))])
    frames = np.stack([pad_sig[i * fstep:i * fstep + flen] for i in range(num_frames)])
    width = 1 << (flen - 1).bit_length()
    left = (width - flen) // 2
    padded = np.zeros((num_frames, width))
    padded[:, left:left + flen] = frames
    padded *= np.hamming(width)
    NFFT = 512
    pow_frames = (1.0 / NFFT) * np.absolute(np.fft.rfft(padded, NFFT)) ** 2
    nfilt = 40
    high_freq = high_freq or sample_rate / 2
    mel = np.linspace(2595 * np.log10(1 + low_freq / 700), 2595 * np.log10(1 + high_freq / 700), nfilt + 2)
    bins = np.floor((NFFT + 1) * (700 * (10 ** (mel / 2595) - 1)) / sample_rate)
    fbank = np.zeros((nfilt, NFFT // 2 + 1))
    for m in range(1, nfilt + 1):
        lo, ce, hi = int(bins[m - 1]), int(bins[m]), int(bins[m + 1])
        for k in range(lo, ce):
            fbank[m - 1, k] = (k - bins[m - 1]) / (bins[m] - bins[m - 1])
        for k in range(ce, hi):
            fbank[m - 1, k] = (bins[m + 1] - k) / (bins[m + 1] - bins[m])
    fb = np.dot(pow_frames, fbank.T)
    fb = np.log10(np.where(fb == 0, np.finfo(float).eps, fb))
    return fb, dct(fb, type=2, axis=1, norm="ortho")[:, 1:14]


def standardize(data):
    """feature.py:85-88: per-column (x - mean) / std over the frames of one utterance
    (population std, np.std's default ddof = 0)."""
    c = data - np.mean(data, axis=0)
    return c / np.std(c, axis=0)


def stack_features(ceps):
    """core.py:27-30 / :41-44: [ceps | delta | delta-delta] -> standardize."""
    df = delta_feature(ceps)
    ddf = delta_feature(df)
    return standardize(np.concatenate([ceps, df, ddf], axis=1))


# --------------------------------------------------------------------------- A3
def gmm_evaluate(x, means, vars_, w, neg_log=True, dense_inv=None):
    """GMM.evaluate (hmm_state.py:114-120): -log(sum_m w_m pdf_m(x)) in the
    linear domain (=> +inf once every component underflows), or the vector of
    weighted component densities when `neg_log` is False."""
    M = means.shape[0]
    res = np.array([gauss_pdf(x, means[m], vars_[m],
                              None if dense_inv is None else dense_inv[m]) * w[m]
                    for m in range(M)])
    if neg_log:
        with np.errstate(divide="ignore"):
            return -np.log(res.sum())
    return res


def gmm_neg_loglik_batch(X, means, vars_, w):
    """Vectorised LOG-domain form of A3 for a batch: X[N,D], means/vars[S,M,D],
    w[S,M] -> [N,S].  This is the algebra the HIP kernels use
    (logc_m - 0.5*sum((x-mu)^2/var), log-sum-exp over m); equal to
    `gmm_evaluate` to ~1e-15 rel wherever the linear domain does not underflow."""
    X = np.asarray(X, dtype=np.float64)
    S, M, D = means.shape
    logc = np.log(w) - 0.5 * (D * np.log(TWO_PI) + np.sum(np.log(vars_), axis=2))  # [S,M]
    out = np.empty((X.shape[0], S))
    for s in range(S):
        dx = X[:, None, :] - means[s][None, :, :]                   # [N,M,D]
        ll = logc[s][None, :] - 0.5 * np.sum(dx * dx / vars_[s][None], axis=2)
        mx = ll.max(axis=1, keepdims=True)
        out[:, s] = -(mx[:, 0] + np.log(np.exp(ll - mx).sum(axis=1)))
    return out


def emission_matrix(x, states, dense=False):
    """E[R,T] for a reference-style state list.  `states[r]` is None for a
    non-emitting row (NES.evaluate -> 0, hmm_state.py:90-91) or a
    (means[M,D], vars[M,D], w[M]) tuple (A3).  Distinct tuples are evaluated
    once and shared between rows (the reference recomputes per cell).
    `dense=True` evaluates through dense inv(diag(var)) matrices like the
    reference does (hmm_state.py:17,42) -- used when timing the CPU baseline."""
    T = len(x)
    E = np.zeros((len(states), T))
    cache = {}
    for r, st in enumerate(states):
        if st is None:
            continue
        key = id(st)
        if key not in cache:
            dinv = np.array([np.linalg.inv(np.diag(v)) for v in st[1]]) if dense else None
            cache[key] = np.array([gmm_evaluate(x[t], *st, dense_inv=dinv) for t in range(T)])
        E[r] = cache[key]
    return E


# --------------------------------------------------------------------------- A6
_NOPTR = np.iinfo(np.int64).min  # what np.full(.., np.inf, dtype=int) yields (decode.py:95)


def decode_fill(E, is_nes, trans, beam=None):
    """The forward sweep of decode_hmm_states (decode.py:94-124): cost matrix + back-pointers.

    beam (SURVEY.md 8(f) N4 -- NOT in the reference's decode_hmm_states, which has no pruning): the rank beam of
    `dtw` (decode.py:62-68) carried over to lattices.  After every column but the last, the cells of the finished
    column are ranked in ascending (cost, row) order (np.argsort of the column, ties by row) and every finite cell
    ranked >= beam is pruned: it reads +inf when the next column takes it as an origin, and the returned matrix shows
    it as +inf (what dtw's matrix shows for every pruned cell but those of its last column).  Reads inside the column
    (arcs touching a non-emitting row) happen before the pruning and are unaffected."""
    R, T = E.shape
    costs = np.full((R, T), np.inf)
    bp = np.full((R, T, 2), _NOPTR, dtype=np.int64)
    preds = [np.flatnonzero(~np.isinf(trans[r])) for r in range(R)]
    for c in range(T):
        for r in range(R):
            if r == 0 and c == 0:
                costs[0, 0] = E[0, 0]
                continue
            if len(preds[r]) == 0:
                continue
            best_v = None
            best_pt = None
            for o in preds[r]:
                cc = c if (is_nes[o] or is_nes[r]) else c - 1
                v = trans[r, o] + costs[o, cc]
                if best_v is None or v < best_v:
                    best_v, best_pt = v, (int(o), cc)
            if best_pt == (r, c):
                raise NameError("FUCKED")  # decode.py:120-121 (self-pointing cell)
            bp[r, c] = best_pt
            costs[r, c] = min(costs[r, c], best_v + E[r, c])
        if beam is not None and not isinf(beam) and c < T - 1:
            order = np.argsort(costs[:, c], kind="stable")          # (cost, row) order
            for r in order[int(beam):]:
                if not isinf(costs[r, c]):
                    costs[r, c] = np.inf
    return costs, bp


def decode_states(E, is_nes, trans, end_points=None, return_bp=False, beam=None):
    """decode_hmm_states (decode.py:80-146) on an emission matrix.

    Columns outer / rows inner (decode.py:97-98); (0,0) is the only start cell
    (:99-101); candidates are the finite arcs in ascending origin order
    (:105-107); an arc touching a non-emitting row reads the SAME column, else
    column c-1 -- which wraps to the last column at c == 0 (:109-114); first
    minimum wins (:118); the last of several equal end points wins (:129-134);
    the returned path excludes the end cell and stops on reaching column 0
    (:143-145).
    """
    R, T = E.shape
    costs, bp = decode_fill(E, is_nes, trans, beam=beam)
    if end_points is None:
        end_points = [[R - 1, T - 1]]
    best = np.inf
    end = []
    for e in end_points:
        if best >= costs[e[0], e[1]]:
            best = costs[e[0], e[1]]
            end = e
    i, j = end[0], end[1]
    if isinf(costs[i, j]):
        warnings.warn("decode_hmm_states: Cannot find a path when decoding sequence")
    path = []
    while j != 0:
        i, j = bp[i, j]
        path.append([i, j])
        if len(path) > R * T:
            # unreachable cells can point at each other inside one column; the reference
            # would loop forever here -- the oracle (test infrastructure) refuses instead
            raise RuntimeError("back-trace does not terminate")
    path = np.array(path)
    if return_bp:
        return costs, path, end, bp
    return costs, path


# --------------------------------------------------------------------------- A5
def dtw(E, trans, beam=np.inf):
    """dtw (decode.py:7-77) on a distance matrix E[n,T] = dist_fun(x[j], y[i]).

    Every origin of the previous column is a candidate, +inf arcs included
    (:44-51); a cell pruned by the beam is marked -1 (:62-68), is skipped as a
    candidate by the FIRST row that reads it and is turned back into +inf there
    (:46-48).  Path runs from (n-1, T-1) until (0,0) (:70-76).
    """
    n, T = E.shape
    assert T > 1 and n > 1  # decode.py:22
    costs = np.full((n, T), np.inf)
    bp = np.full((n, T, 2), _NOPTR, dtype=np.int64)
    for j in range(T):
        for i in range(n):
            if i == 0 and j == 0:
                costs[0, 0] = E[0, 0]
                continue
            cand, pts = [], []
            for o in range(trans.shape[1]):
                if costs[o, j - 1] == -1:
                    costs[o, j - 1] = np.inf
                else:
                    cand.append(trans[i, o] + costs[o, j - 1])
                    pts.append((o, j - 1))
            k = int(np.argmin(cand))
            bp[i, j] = pts[k]
            costs[i, j] = min(costs[i, j], cand[k] + E[i, j])
        if not isinf(beam):
            order = np.argsort(costs[:, j].flatten())
            for i in order[beam:]:
                if not isinf(costs[i, j]):
                    costs[i, j] = -1
    i, j = n - 1, T - 1
    path = []
    while i != 0 or j != 0:
        i, j = bp[i, j]
        path.append([i, j])
    return costs, np.array(path)


def distance_matrix(x, y, dist="euclid", variance=None):
    """E[i,j] = dist_fun(x[j], y[i][, variance[i]]) as dtw calls it (decode.py:35-38,56-59)."""
    n, T = len(y), len(x)
    E = np.empty((n, T))
    for i in range(n):
        for j in range(T):
            if dist == "euclid":
                E[i, j] = euclid(x[j], y[i])
            else:
                E[i, j] = mahalanobis(x[j], y[i], variance[i])
    return E


# --------------------------------------------------------------------------- A7
def gmm_em(data, means, vars_, w, k, max_iteration=10000, old=None):
    """GMM.em (hmm_state.py:122-159) for the first `k` of the M components.

    Arrays are updated in place like the reference mutates its object
    (hmm_state.py:161-165).  `old` = (mu_old[M,D], sigma_old[M,D], w_old[M]),
    the *_old members the convergence test compares against (:150-159); they
    start as tiles of the state's initial (mu, sigma) and 1/M (:110-112).
    Returns the number of iterations executed (index of the converged iteration
    + 1, or max_iteration).
    """
    N, D = data.shape
    mu_old, sigma_old, w_old = old
    it = 0
    for it in range(max_iteration):
        p = np.zeros((N, k))
        for i in range(N):
            p[i, :] = gmm_evaluate(data[i], means, vars_, w, neg_log=False)[:k]
        rs = np.sum(p, axis=1).reshape((N, 1))
        rs[rs == 0] = 10 ** (-5)
        p /= rs
        cs = np.sum(p, axis=0)
        cs[cs == 0] = 10 ** (-5)
        mu = np.zeros((D, k))
        sg = np.zeros((D, k))
        for c in range(k):
            mu[:, c] = np.sum(data * p[:, [c]], axis=0) / cs[c]
            sg[:, c] = np.sum((data - mu[:, c]) ** 2 * p[:, [c]], axis=0) / cs[c]
        mu, sg = mu.T, sg.T
        wn = p.mean(axis=0).T
        w[:k] = wn
        means[:k] = mu
        vars_[:k] = sg
        if np.any(sg == 0):
            # cov setter inverts diag(sigma) on every assignment (hmm_state.py:24-30)
            raise np.linalg.LinAlgError("Singular matrix")
        if np.allclose(mu, mu_old[:k]) and np.allclose(sg, sigma_old[:k]) and np.allclose(wn, w_old[:k]):
            break
        mu_old[:k] = mu
        sigma_old[:k] = sg
        w_old[:k] = wn
    return it + 1


# -------------------------------------------------------------------------- A14
def calc_variance(data):
    """kmeans.py:6-12 -- rows are variables: np.cov(...).diagonal(), ddof=1."""
    return np.cov(data).diagonal()


def cluster_centroids(data, clusters, k):
    """kmeans.py:158-164 (an empty cluster yields a NaN row + RuntimeWarning)."""
    out = np.empty((k,) + data.shape[1:])
    for i in range(k):
        np.mean(data[clusters == i, :], axis=0, out=out[i])
    return out


def kmeans(data, k, centroids, dist="euclid", max_iteration=1000):
    """kmeans (kmeans.py:167-193).  The random initial partition is drawn from
    the GLOBAL numpy RNG (:171) and only feeds the per-cluster variance `cov`
    (:173-177), which is returned unchanged (:193); every distance uses
    cov[0] (:183); the loop stops on exact equality of centroids (:190)."""
    assert k == centroids.shape[0]
    clusters = np.random.randint(0, k, data.shape[0])
    cov = np.array([calc_variance(data[clusters == c].T) for c in range(k)])
    N = data.shape[0]
    for _ in range(max(max_iteration, 1)):
        d = np.zeros((N, k))
        for i in range(N):
            for c in range(k):
                if dist == "euclid":
                    d[i, c] = euclid(centroids[c, :], data[i, :])
                else:
                    d[i, c] = mahalanobis(centroids[c, :], data[i, :], cov[0])
        clusters = np.argmin(d, axis=1)
        new_c = cluster_centroids(data, clusters, k)
        if np.array_equal(new_c, centroids):
            break
        centroids = new_c
    return clusters, centroids, cov


# -------------------------------------------------------------------------- A15
def segment_data(templates, n_segments, seg_starts):
    """kmeans.py:33-50: gather frames of segment s over all templates."""
    out = []
    for s in range(n_segments):
        rows = []
        for r, t in enumerate(templates):
            if s == n_segments - 1:
                rows += t[seg_starts[r, s]:].tolist()
            else:
                rows += t[seg_starts[r, s]:seg_starts[r, s + 1]].tolist()
        out.append(np.array(rows))
    return out


def combine_templates(templates, n_segments, seg_starts):
    """kmeans.py:15-30: per-segment mean and ddof=1 variance."""
    D = templates[0].shape[1]
    mu = np.zeros((n_segments, D))
    var = np.zeros((n_segments, D))
    segs = segment_data(templates, n_segments, seg_starts)
    for s in range(n_segments):
        mu[s] = segs[s].mean(axis=0)
        var[s] = calc_variance(segs[s].T)
    return mu, var


def calc_transition_costs(n_temps, seg_lens, max_jump_dist=2):
    """kmeans.py:53-95: left-to-right costs from segment lengths; a jump skips
    over (all-template-)empty segments up to `max_jump_dist`."""
    n = seg_lens.shape[1]
    empty = seg_lens == 0
    res = np.full((n, n), np.inf)
    for i in range(n):
        jump = 1
        n_jump = 0 if i == n - 1 else n_temps
        s = i + 1
        while s < n - 1:
            if np.sum(empty[:, s + 1]) == 0:
                break
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
    """kmeans.py:98-108: segment starts = running counts of each row in the path."""
    counts = np.zeros(n_segments, dtype=np.int64)
    u, c = np.unique(path[:, 0], return_counts=True)
    counts[u] = c
    return np.add.accumulate(counts)[:-1]


def skmeans(templates, n_segments, max_iteration=1000):
    """skmeans (kmeans.py:111-155) with its default Euclidean `dist_fun`.
    The transition costs are always those of the initial uniform segmentation
    (`seg_lens` is never updated inside the loop, :139)."""
    assert max_iteration > 0
    n_temps = len(templates)
    seg_lens = np.zeros((n_temps, n_segments + 1), dtype=np.int64)
    for r in range(n_temps):
        seg_lens[r, 1:] = len(templates[r]) // n_segments
    seg_starts = np.add.accumulate(seg_lens, axis=1)[:, :-1]
    seg_lens = seg_lens[:, 1:]
    trans = None
    res, var = combine_templates(templates, n_segments, seg_starts)
    for _ in range(max_iteration):
        seg_starts = np.zeros((n_temps, n_segments), dtype=np.int64)
        trans = calc_transition_costs(n_temps, seg_lens)
        for r in range(n_temps):
            if templates[r].shape[0] < 5:
                raise NameError("template is too small, cannot do dtw on it")
            _, path = dtw(distance_matrix(templates[r], res, "euclid"), trans)
            seg_starts[r, 1:] = get_segments_from_path(path, n_segments)
        new_res, var = combine_templates(templates, n_segments, seg_starts)
        if np.allclose(res, new_res):
            break
        res = new_res
    return res, var, trans, segment_data(templates, n_segments, seg_starts)


def align_gmm_states(templates, states, trans, n_segments):
    """kmeans.py:196-205: re-segment by A6 against the trained mixtures."""
    seg_starts = np.zeros((len(templates), n_segments), dtype=np.int64)
    nes = np.zeros(len(states), dtype=bool)
    for r, t in enumerate(templates):
        _, path = decode_states(emission_matrix(t, states), nes, trans)
        seg_starts[r, 1:] = get_segments_from_path(path, n_segments)
    return segment_data(templates, n_segments, seg_starts)


# --------------------------------------------------------------------------- A9
def split_fit_gmm(data, centroid0, state, n_gaussians, weight_den, weight_init, use_em=True):
    """Binary-split k-means + EM shared by HMM._fit_GMM (hmm.py:97-124) and the
    refit inside continuous_train (continuous_speech.py:122-142).

    `state` = dict(means[M,D], vars[M,D], w[M], mu_old, sigma_old, w_old),
    mutated in place.  n_splits = int(ln(n_gaussians)) (hmm.py:104): 4 -> 1
    split (2 trained components), 8 -> 2, 32 -> 3.  Cluster weights are
    `count / weight_den`, with the counts indexed by cluster ID (hmm.py:116-118).
    """
    n_splits = int(np.log(n_gaussians))
    assert n_splits > 0
    centroids = np.array(centroid0).reshape(1, -1)
    weights = np.full(n_gaussians, weight_init)
    iters = []
    for i in range(n_splits):
        k = 2 ** (i + 1)
        centroids = np.concatenate([centroids * 0.9, centroids * 1.1], axis=0)
        clusters, centroids, variance = kmeans(data, k, centroids, dist="mahalanobis")
        cs, cnt = np.unique(clusters, return_counts=True)
        for c in cs:
            weights[c] = cnt[c] / weight_den
        state["w"][:k] = weights[:k]
        state["means"][:k] = centroids
        state["vars"][:k] = variance
        if use_em:
            iters.append(gmm_em(data, state["means"], state["vars"], state["w"], k,
                                old=(state["mu_old"], state["sigma_old"], state["w_old"])))
    return iters


def new_gmm_state(mu, sigma, n_gaussians):
    """GMM.__init__ (hmm_state.py:105-112)."""
    return dict(means=np.tile(mu, (n_gaussians, 1)), vars=np.tile(sigma, (n_gaussians, 1)),
                w=np.full(n_gaussians, 1 / n_gaussians),
                mu_old=np.tile(mu, (n_gaussians, 1)), sigma_old=np.tile(sigma, (n_gaussians, 1)),
                w_old=np.full(n_gaussians, 1 / n_gaussians))


def state_tuple(st):
    return (st["means"], st["vars"], st["w"])


def hmm_fit(ys, n_segments, n_gaussians, use_gmm=True, use_em=True):
    """HMM.fit (hmm.py:57-95).  Returns dict(mu, sigma, transitions, segments[, gmm])."""
    mu, sigma, trans, segments = skmeans(ys, n_segments)
    model = dict(mu=mu, sigma=sigma, transitions=trans, segments=segments, n_segments=n_segments)
    if not use_gmm:
        return model
    gmm = [new_gmm_state(mu[i], sigma[i], n_gaussians) for i in range(n_segments)]
    for i in range(n_segments):
        split_fit_gmm(segments[i], mu[i, :], gmm[i], n_gaussians,
                      weight_den=segments[i].shape[0], weight_init=1 / segments[i].shape[0],
                      use_em=use_em)
    model["gmm"] = gmm
    model["segments"] = align_gmm_states(ys, [state_tuple(g) for g in gmm], trans, n_segments)
    return model


def hmm_evaluate(x, model, use_gmm=True):
    """HMM.evaluate (hmm.py:126-135): cost of the last state at the last frame."""
    n = model["n_segments"]
    if use_gmm:
        E = emission_matrix(x, [state_tuple(g) for g in model["gmm"]])
        costs, _ = decode_states(E, np.zeros(n, dtype=bool), model["transitions"])
    else:
        costs, _ = dtw(distance_matrix(x, model["mu"], "mahalanobis", model["sigma"]),
                       model["transitions"])
    return costs[-1, -1]


# -------------------------------------------------------------------------- A10
def build_state_sequences(n_states_per_word, word_trans, label_matrix):
    """build_state_sequences (continuous_speech.py:13-53) on packed words.

    `word_trans[l]` is the [n,n] transition block of word l.  Returns
    (row_word[R], row_state[R], is_nes[R], trans[R,R], end_rows) where
    row_word/row_state = -1 on non-emitting rows.  Row 0 is non-emitting, each
    layer is followed by one non-emitting row (:23,40-41); zero-cost arcs
    NES_k -> word start and word end -> NES_{k+1} (:46-49); the end rows are the
    last layer's last EMITTING rows (:53)."""
    n = n_states_per_word
    R = 1 + sum(len(lbls) * n + 1 for lbls in label_matrix)
    trans = np.full((R, R), np.inf)
    row_word = [-1]
    row_state = [-1]
    nes_rows = [0]
    starts, ends = [], []
    for labels in label_matrix:
        ls, le = [], []
        for l in labels:
            off = len(row_word)
            assert word_trans[l].shape[0] == n  # continuous_speech.py:37
            ls.append(off)
            le.append(off + n - 1)
            row_word += [l] * n
            row_state += list(range(n))
            trans[off:off + n, off:off + n] = word_trans[l]
        row_word.append(-1)
        row_state.append(-1)
        nes_rows.append(len(row_word) - 1)
        starts.append(ls)
        ends.append(le)
    for k in range(len(label_matrix)):
        for s in starts[k]:
            trans[s, nes_rows[k]] = 0
        for e in ends[k]:
            trans[nes_rows[k + 1], e] = 0
    row_word = np.array(row_word)
    return row_word, np.array(row_state), row_word < 0, trans, ends[-1]


# -------------------------------------------------------------------------- A12
def loop_grammar(word_transitions, n, word_penalty=0.0):
    """N4: word LOOP grammar -- NOT in the reference (which only has the
    exactly-K-words lattice of build_state_sequences); validated in tests as the minimum over K
    of the reference-style K-layer decodes.

    The reference's DP reads same-column origins only from rows already visited in this
    column (decode.py:97-98,109-111), so a loop is expressible by ROW ORDER alone:
        row 0                       non-emitting start (the reference's single start cell)
        rows of states 1..n-1       of every word, word-major
        one non-emitting LOOP row   collects every word's last state (same column, cost 0)
        rows of state 0             of every word: from the start row (0), from the loop row
                                    (word_penalty) -- both same-column hops -- and their self loop
    A word end at frame c therefore reaches the next word's first state in the same column, which
    scores frame c again: exactly the double emission of the K-layer lattice (SURVEY.md A6 quirk i).
    Requires n >= 2 (a one-state word would be its own same-column ancestor).

    Returns (is_nes [R], row_word [R] (-1 on non-emitting rows), row_local [R], trans [R,R], end_rows)."""
    W = len(word_transitions)
    assert n >= 2
    R = 2 + W * n
    loop_row = 1 + W * (n - 1)
    row_word = np.full(R, -1, dtype=np.int64)
    row_local = np.full(R, -1, dtype=np.int64)
    row_of = np.empty((W, n), dtype=np.int64)
    for w in range(W):
        for i in range(1, n):
            row_of[w, i] = 1 + w * (n - 1) + (i - 1)
        row_of[w, 0] = loop_row + 1 + w
    for w in range(W):
        for i in range(n):
            row_word[row_of[w, i]] = w
            row_local[row_of[w, i]] = i
    trans = np.full((R, R), np.inf)
    ends = []
    for w in range(W):
        wt = np.asarray(word_transitions[w], dtype=np.float64)
        for i in range(n):
            for j in range(n):
                if not np.isinf(wt[i, j]):
                    trans[row_of[w, i], row_of[w, j]] = wt[i, j]
        trans[row_of[w, 0], 0] = 0.0
        trans[row_of[w, 0], loop_row] = word_penalty
        trans[loop_row, row_of[w, n - 1]] = 0.0
        ends.append(int(row_of[w, n - 1]))
    return row_word < 0, row_word, row_local, trans, ends


def path_to_words(path, is_nes, row_word):
    """main.py:59-67 (+ split_result :39-52): reversed row sequence -> drop
    consecutive duplicates -> first emitting row of every run between
    non-emitting rows -> word index of that row."""
    rows = path[:, 0][::-1]
    rows = rows[np.insert(np.diff(rows).astype(bool), 0, True)]
    out = []
    cur = None
    for r in rows:
        if not is_nes[r]:
            if cur is None:
                cur = r
        elif cur is not None:
            out.append(int(row_word[cur]))
            cur = None
    if cur is not None:
        out.append(int(row_word[cur]))
    return out


# -------------------------------------------------------------------------- A11
def cut_segments(path, is_nes):
    """The regrouping loop of continuous_train (continuous_speech.py:90-106) on one alignment path (end -> start, as
    decode_states returns it): walking it start -> end, a run opens at the first cell of an emitting row seen while no
    run is open; a cell of a DIFFERENT row closes the open run as the frames [start, c), c = that cell's column,
    provided start < c, and does not itself open a run.  Yields (row, start, stop)."""
    start, cur = None, None
    for r, c in reversed(np.asarray(path).tolist()):
        if start is None and not is_nes[r]:
            start, cur = c, r
        if r != cur and start is not None and start < c:
            yield int(cur), int(start), int(c)
            start, cur = None, None


def continuous_train(data, models, label_seqs, n_gaussians=4, n_segments=5, max_iteration=1000,
                     on_iteration=None):
    """continuous_train (continuous_speech.py:56-179) on packed models.

    `models[i]` = dict(gmm=[state dicts], transitions[n,n]).  Per outer
    iteration: forced alignment of every utterance through its one-word-per-layer
    lattice (:80-89); frames regrouped per (word, state) -- a segment is cut when
    the row changes, the boundary frame goes to the next state and the final
    state's last segment is never flushed (:94-106); each visited state refit by
    split-k-means + EM in first-visit order (:114-142; weights count/n_segments);
    transitions from segment counts (:146-164); stop when every model's mixtures
    are allclose to the previous iteration's (:172-179; transitions not compared).
    Returns (models, n_outer_iterations, last gathered segments dict).
    """
    import copy
    old_models = models
    new_models = copy.deepcopy(models)
    # modelidx_state_map (continuous_speech.py:64-71) keeps the state OBJECTS of this
    # first copy for the whole run.  Iteration 0 looks them up by identity; later
    # iterations train fresh deep copies (same uuid => same hash), so the dict lookup
    # at :149 falls through to GMM.__eq__(current, first-copy) -- allclose on the
    # parameters -- and a state that moved since iteration 0 is reported as having
    # no data: its transitions are then NOT re-estimated.
    map_models = new_models
    gathered = None
    it = 0
    for it in range(max_iteration):
        gmm_data = {}  # (word, state) -> [segments], insertion-ordered like the reference dict
        word_trans = [m["transitions"] for m in new_models]
        for x, labels in zip(data, label_seqs):
            rw, rs, nes, trans, ends = build_state_sequences(n_segments_of(new_models), word_trans,
                                                             [[l] for l in labels])
            states = [None if nes[r] else state_tuple(new_models[rw[r]]["gmm"][rs[r]]) for r in range(len(rw))]
            # rows of a repeated word share one state object in the reference: share tuples too
            uniq = {}
            for r in range(len(rw)):
                if not nes[r]:
                    states[r] = uniq.setdefault((rw[r], rs[r]), states[r])
            _, path = decode_states(emission_matrix(x, states), nes, trans,
                                    end_points=[[e, -1] for e in ends])
            for cur, start, c in cut_segments(path, nes):
                gmm_data.setdefault((int(rw[cur]), int(rs[cur])), []).append(x[start:c])
        for (wi, si), segs in gmm_data.items():
            seg = np.vstack(segs)
            split_fit_gmm(seg, np.mean(seg, axis=0), new_models[wi]["gmm"][si], n_gaussians,
                          weight_den=n_segments, weight_init=1 / n_segments, use_em=True)
        for mi, m in enumerate(new_models):
            ns = len(m["gmm"])
            for si in range(ns):
                segs = gmm_data.get((mi, si))
                if segs is not None and new_models is not map_models and \
                        not gmm_equal(m["gmm"][si], map_models[mi]["gmm"][si]):
                    segs = None
                if segs is None:
                    warnings.warn("No MFCC data for state", UserWarning)
                    continue
                p_jump = len(segs) / sum(s.shape[0] for s in segs)
                if si < ns - 1:
                    m["transitions"][si + 1, si] = -np.log(p_jump)
                m["transitions"][si, si] = -np.log(1 - p_jump)
        gathered = gmm_data
        if on_iteration is not None:
            on_iteration(it, new_models)
        if all(models_equal(a, b) for a, b in zip(new_models, old_models)):
            break
        old_models = new_models
        new_models = copy.deepcopy(old_models)
    return new_models, it + 1, gathered


def n_segments_of(models):
    return len(models[0]["gmm"])  # continuous_speech.py:27-28 (all words same length)


def gmm_equal(a, b):
    """GMM.__eq__ (hmm_state.py:167-173)."""
    return (len(a["w"]) == len(b["w"]) and np.allclose(a["w"], b["w"])
            and all(np.allclose(a["means"][i], b["means"][i]) and np.allclose(a["vars"][i], b["vars"][i])
                    for i in range(len(a["w"]))))


def models_equal(a, b):
    """HMM.__eq__ for use_gmm models (hmm.py:30-39)."""
    return len(a["gmm"]) == len(b["gmm"]) and all(gmm_equal(x, y) for x, y in zip(a["gmm"], b["gmm"]))


# -------------------------------------------------------------------------- A13
def _lse(vals):
    vals = np.asarray(vals, dtype=np.float64)
    m = vals.max()
    if isinf(m):
        return m
    return m + np.log(np.exp(vals - m).sum())


def forward_backward(E, is_nes, trans, end_rows):
    """Sum-product twin of A6 -- NOT IN THE REFERENCE ("parity unpinned";
    SURVEY.md section 8(a) A13; pinned by brute-force path enumeration in tests).

    Same arcs, same same-column rule for arcs touching a non-emitting row, start
    only at (0,0), end set = `end_rows` in the last column; min -> -logsumexp.
    Works in the log-probability domain (logp = -cost).  Returns
    (log_alpha[R,T], log_beta[R,T], gamma[R,T], logP) where gamma is the
    posterior of passing through cell (r,t).  Requires every same-column arc to
    run from a lower to a higher row (true for A10 lattices)."""
    R, T = E.shape
    NEG = -np.inf
    la = np.full((R, T), NEG)
    preds = [np.flatnonzero(~np.isinf(trans[r])) for r in range(R)]
    succs = [np.flatnonzero(~np.isinf(trans[:, r])) for r in range(R)]
    for c in range(T):
        for r in range(R):
            if r == 0 and c == 0:
                la[0, 0] = -E[0, 0]
                continue
            terms = []
            for o in preds[r]:
                same = is_nes[o] or is_nes[r]
                if same:
                    if o >= r:
                        continue  # not yet computed in this column: +inf cost in A6
                    terms.append(la[o, c] - trans[r, o])
                elif c > 0:
                    terms.append(la[o, c - 1] - trans[r, o])
            if terms:
                la[r, c] = _lse(terms) - E[r, c]
    lb = np.full((R, T), NEG)
    end_set = set(int(e) for e in end_rows)
    for c in range(T - 1, -1, -1):
        for r in range(R - 1, -1, -1):
            terms = []
            if c == T - 1 and r in end_set:
                terms.append(0.0)
            for s in succs[r]:
                same = is_nes[s] or is_nes[r]
                if same:
                    if s <= r:
                        continue
                    terms.append(lb[s, c] - trans[s, r] - E[s, c])
                elif c + 1 < T:
                    terms.append(lb[s, c + 1] - trans[s, r] - E[s, c + 1])
            if terms:
                lb[r, c] = _lse(terms)
    logp = _lse([la[e, T - 1] for e in end_rows])
    with np.errstate(invalid="ignore"):
        gamma = np.exp(la + lb - logp)
    gamma[np.isnan(gamma)] = 0.0
    return la, lb, gamma, logp
