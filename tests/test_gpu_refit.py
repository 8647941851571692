# -*- coding: utf-8 -*-
"""The streaming matrix-core kernels of the device-resident refit (csrc/gh_refit_mfma.hip: refit_km_kernel / refit_em_kernel
behind gh_fit_kmeans / gh_fit_em): against the oracle's kmeans / gmm_em state by state (reference kmeans.py:167-193,
hmm_state.py:122-159), against the tile kernels they replace (GMMHMM_REFIT=tiles) over shapes that exercise every
instantiation and edge (D = 2 .. 64, k = 1 .. 8, empty / one-frame / slab-sized states), and on the cases the fast path
hands to the reference's own arithmetic: exact ties, NaN centroids, densities that underflow."""
import os

import numpy as np
import pytest

from oracle import ref_numpy as O

pytestmark = pytest.mark.gpu


def _session(segs, kmax, tiles=False):
    from sr.recognition import _hip
    ctx = _hip.default_context()
    off = np.concatenate([[0], np.cumsum([len(x) for x in segs])]).astype(np.int64)
    b = _hip.Batch(ctx, feats=np.concatenate(segs), offsets=[0, int(off[-1])])
    old = os.environ.pop("GMMHMM_REFIT", None)
    try:
        if tiles:
            os.environ["GMMHMM_REFIT"] = "tiles"
        fit = _hip.FitSession(ctx, b, off, kmax)
    finally:
        os.environ.pop("GMMHMM_REFIT", None)
        if old is not None:
            os.environ["GMMHMM_REFIT"] = old
    return b, fit, off


def _mixture_data(rng, lens, D, k, spread=3.0):
    segs = []
    for n in lens:
        c = rng.normal(size=(max(k, 1), D)) * spread + rng.normal(size=D) * 4
        segs.append(c[rng.integers(0, max(k, 1), n)] + rng.normal(size=(n, D)) * rng.uniform(0.5, 1.5, size=D))
    return segs


@pytest.mark.parametrize("D,k,lens", [(2, 2, (40, 90, 33)), (13, 4, (150, 64, 257, 16)), (13, 8, (300, 500)), (39, 4, (700, 333)),
                                      (39, 2, (90, 1000))])
def test_streaming_refit_equals_the_oracle_state_by_state(D, k, lens):
    rng = np.random.default_rng(100 * D + k)
    segs = _mixture_data(rng, lens, D, k)
    S = len(segs)
    c0 = np.stack([np.stack([x.mean(axis=0) * f for f in np.linspace(0.85, 1.15, k)]) for x in segs])
    # the oracle, state after state, drawing its partitions from the global generator; the same draws for the session
    np.random.seed(4)
    parts = [np.random.randint(0, k, len(x)) for x in segs]
    np.random.seed(4)
    want = []
    with np.errstate(all="ignore"):
        for s in range(S):
            want.append(O.kmeans(segs[s], k, c0[s].copy(), dist="mahalanobis", max_iteration=300))
    b, fit, off = _session(segs, k)
    cen, cov, cnt, its = fit.kmeans(k, c0, np.concatenate(parts).astype(np.uint8), max_iteration=300)
    ids = fit.clusters()
    for s in range(S):
        cl, ce, cv = want[s]
        np.testing.assert_array_equal(ids[off[s]:off[s + 1]], cl)
        np.testing.assert_array_equal(cen[s], ce)                        # sums in frame order: numpy's means, bit for bit
        np.testing.assert_allclose(cov[s], cv, rtol=1e-11)
        np.testing.assert_array_equal(cnt[s], np.bincount(cl, minlength=k))
    # EM from there (the first k components of a fresh state, hmm_state.py:104-112)
    w0 = cnt / np.diff(off)[:, None]
    mean, var, w = cen.copy(), cov.copy(), w0.copy()
    mu_old = np.array([np.tile(x.mean(axis=0), (k, 1)) for x in segs])
    sg_old = np.array([np.tile(x.var(axis=0), (k, 1)) for x in segs])
    w_old = np.full((S, k), 1.0 / k)
    want_it = []
    ref = []
    for s in range(S):
        m, v, ww = cen[s].copy(), cov[s].copy(), w0[s].copy()
        old = (mu_old[s].copy(), sg_old[s].copy(), w_old[s].copy())
        want_it.append(O.gmm_em(segs[s], m, v, ww, k, max_iteration=40, old=old))
        ref.append((m, v, ww, old))
    conv = fit.em(k, mean, var, w, mu_old, sg_old, w_old, np.diff(off).astype(np.float64), max_iteration=40)
    for s in range(S):
        m, v, ww, old = ref[s]
        assert (conv[s] + 1 if conv[s] >= 0 else 40) == want_it[s]
        np.testing.assert_allclose(mean[s], m, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(var[s], v, rtol=1e-8)
        np.testing.assert_allclose(w[s], ww, rtol=1e-9)
        np.testing.assert_allclose(mu_old[s], old[0], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(sg_old[s], old[1], rtol=1e-8)
    fit.close()
    b.close()


def _both(segs, k, c0, part, kmax=None, em_iters=25, km_iters=60):
    out = []
    for tiles in (False, True):
        b, fit, off = _session(segs, kmax or max(k, 2), tiles=tiles)
        cen, cov, cnt, its = fit.kmeans(k, c0, part, max_iteration=km_iters)
        ids = fit.clusters()
        n = np.diff(off).astype(np.float64)
        with np.errstate(all="ignore"):
            mean, var, w = cen.copy(), cov.copy(), cnt / np.maximum(n, 1.0)[:, None]
        ok = np.isfinite(var).all() and (var != 0).all() and np.isfinite(mean).all()
        res = dict(cen=cen, cov=cov, cnt=cnt, its=its, ids=ids)
        if ok:
            mu_old, sg_old, w_old = np.zeros_like(mean), np.ones_like(mean), np.zeros_like(w)
            conv = fit.em(k, mean, var, w, mu_old, sg_old, w_old, n, max_iteration=em_iters)
            res.update(mean=mean, var=var, w=w, conv=conv, mu_old=mu_old)
        out.append(res)
        fit.close()
        b.close()
    return out


@pytest.mark.parametrize("D", [2, 3, 12, 13, 15, 16, 17, 38, 39, 40, 47, 64])
@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 8])
def test_streaming_kernels_equal_the_tile_kernels(D, k):
    """Every instantiation (one or two component groups; 4, 10 or 17 accumulator columns, exact or not) and every edge of the
    slab walk: states with 0, 1, 15, 16, 17 frames, one exactly an item long, one a little longer."""
    rng = np.random.default_rng(7 * D + k)
    lens = (0, 1, 15, 16, 17, 512, 530, 200)
    segs = _mixture_data(rng, lens, D, k)
    N = sum(lens)
    part = rng.integers(0, k, size=N).astype(np.uint8)
    c0 = np.stack([np.stack([(x.mean(axis=0) if len(x) else np.zeros(D)) * f + 0.01 * j for j, f in enumerate(np.linspace(0.8, 1.2, k))])
                   for x in segs])
    new, old = _both(segs, k, c0, part)
    np.testing.assert_array_equal(new["ids"], old["ids"])
    np.testing.assert_array_equal(new["its"], old["its"])
    np.testing.assert_array_equal(new["cnt"], old["cnt"])
    np.testing.assert_array_equal(new["cen"], old["cen"])
    # the partition variances: ONE streaming pass around the state's shift point against two frame-order passes; the one-pass
    # form loses ~ (distance of the group from the shift point / its own spread)^2 ulps -- nothing for groups of real size
    # (test_device_session_equals_the_host_loop holds them to 1e-12 of np.cov), a few digits for the 2-5 frame groups here
    np.testing.assert_allclose(new["cov"], old["cov"], rtol=1e-9, equal_nan=True)
    assert ("mean" in new) == ("mean" in old)
    if "mean" in new:
        np.testing.assert_array_equal(new["conv"], old["conv"])
        for name in ("mean", "var", "w", "mu_old"):
            np.testing.assert_allclose(new[name], old[name], rtol=1e-9, atol=1e-11, equal_nan=True)


def test_ties_nan_centroids_and_far_frames_take_the_reference_arithmetic():
    """Two identical centroids (every frame ties exactly: np.argmin's first index), a NaN centroid (its distance is NaN for
    every frame: np.argmin returns it), data far from zero and from each other (the expanded form's terms cancel at
    1e8 and must not decide anything), an infinite feature value."""
    rng = np.random.default_rng(5)
    D, k = 13, 4
    segs = _mixture_data(rng, (300, 300, 400, 120), D, k)
    segs[2] = segs[2] + 1e4                                    # far from the origin, tight clusters
    segs[3][7, 3] = np.inf
    N = sum(len(x) for x in segs)
    part = rng.integers(0, k, size=N).astype(np.uint8)
    c0 = np.stack([np.stack([x[np.isfinite(x).all(axis=1)].mean(axis=0) * f for f in (0.9, 0.9, 1.1, 1.2)]) for x in segs])
    c0[1, 2] = np.nan
    with np.errstate(all="ignore"):
        new, old = _both(segs, k, c0, part, km_iters=3)
    np.testing.assert_array_equal(new["ids"], old["ids"])
    np.testing.assert_array_equal(new["cnt"], old["cnt"])
    np.testing.assert_array_equal(new["cen"], old["cen"])
    # and against numpy directly for the first sweep of state 0: duplicates go to the lower index
    b, fit, off = _session(segs, k)
    fit.kmeans(k, c0, part, max_iteration=1)
    ids = fit.clusters()
    x = segs[0]
    cl = part[:len(x)]
    var0 = np.cov(x[cl == 0].T).diagonal()
    d = np.array([[O.mahalanobis(c0[0, c], x[i], var0) for c in range(k)] for i in range(len(x))])
    np.testing.assert_array_equal(ids[:len(x)], np.argmin(d, axis=1))
    assert not np.any(ids[:len(x)] == 1)                       # (centroid 1 duplicates centroid 0)
    np.testing.assert_array_equal(ids[off[1]:off[2]], 2)       # the NaN centroid takes every frame of its state
    fit.close()
    b.close()


def test_em_densities_that_underflow_weigh_nothing():
    """A frame whose every component density underflows in the linear domain (hmm_state.py:114-120 multiplies
    exp(-q/2) by the normaliser) has no responsibilities at all; one whose exponent alone underflows drops that
    component.  Against the oracle's gmm_em (one iteration at a time) and the tile kernels."""
    rng = np.random.default_rng(9)
    D, k = 13, 4
    segs = _mixture_data(rng, (600, 500), D, k, spread=1.0)
    far = [np.arange(0, 600, 50), np.arange(0, 500, 70)]
    segs[0][far[0]] += 300.0                                   # ~300 standard deviations out in every dimension: q ~ 1e6
    segs[1][far[1]] -= 250.0
    S = len(segs)
    n = np.array([len(x) for x in segs], dtype=np.float64)
    mean0 = np.stack([np.stack([np.delete(x, f, axis=0).mean(axis=0) + 0.3 * j for j in range(k)]) for x, f in zip(segs, far)])
    var0 = np.stack([np.tile(np.delete(x, f, axis=0).var(axis=0), (k, 1)) for x, f in zip(segs, far)])
    w0 = np.full((S, k), 0.25)
    res = []
    for tiles in (False, True):
        b, fit, off = _session(segs, k, tiles=tiles)
        mean, var, w = mean0.copy(), var0.copy(), w0.copy()
        mu_old, sg_old, w_old = np.zeros_like(mean), np.ones_like(mean), np.zeros_like(w)
        conv = fit.em(k, mean, var, w, mu_old, sg_old, w_old, n, max_iteration=3)
        res.append((mean, var, w, conv))
        fit.close()
        b.close()
    for s in range(S):
        m, v, ww = mean0[s].copy(), var0[s].copy(), w0[s].copy()
        O.gmm_em(segs[s], m, v, ww, k, max_iteration=3, old=(np.zeros((k, D)), np.ones((k, D)), np.zeros(k)))
        for mean, var, w, conv in res:
            np.testing.assert_allclose(mean[s], m, rtol=1e-8, atol=1e-10)
            np.testing.assert_allclose(var[s], v, rtol=1e-8)
            np.testing.assert_allclose(w[s], ww, rtol=1e-9)
        # the far frames weigh nothing: the weights add up to the share of the others
        np.testing.assert_allclose(res[0][2][s].sum(), 1.0 - len(far[s]) / n[s], rtol=1e-12)


def test_more_than_eight_components_keep_the_tile_kernels():
    rng = np.random.default_rng(2)
    D, k = 13, 16
    segs = _mixture_data(rng, (900, 700), D, k)
    N = sum(len(x) for x in segs)
    part = rng.integers(0, k, size=N).astype(np.uint8)
    c0 = np.stack([np.stack([x.mean(axis=0) * f for f in np.linspace(0.7, 1.3, k)]) for x in segs])
    new, old = _both(segs, k, c0, part, kmax=16, em_iters=5, km_iters=10)
    np.testing.assert_array_equal(new["ids"], old["ids"])
    np.testing.assert_array_equal(new["cen"], old["cen"])
    assert ("mean" in new) == ("mean" in old)
    if "mean" in new:
        np.testing.assert_array_equal(new["mean"], old["mean"])       # (the same kernels ran: the same bits)


@pytest.mark.parametrize("D", [2, 13, 39, 63, 64])
def test_segment_means_are_numpys_row_after_row_sums(D):
    """gh_fit_segment_means (fit_segsum_wide_kernel: eight waves fetch, one adds): np.mean(segment, axis=0) bit for bit for
    states of 0, 1 and a few frames, lengths around the 32 rows of a wave's share and the 256 rows of a phase, an odd
    number of phases, and one long state; a NaN / inf frame lands where numpy puts it."""
    from sr.recognition import _hip
    ctx = _hip.default_context()
    rng = np.random.default_rng(100 + D)
    lens = [0, 1, 5, 31, 32, 33, 255, 256, 257, 511, 512, 513, 767, 768, 1025, 0, 20011]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    X = rng.normal(size=(int(off[-1]), D)) * 10.0 ** rng.integers(-3, 4, size=D)
    X[off[9] + 7, 0] = np.nan
    X[off[10] + 300, D - 1] = np.inf
    b = _hip.Batch(ctx, feats=X, offsets=[0, len(X)])
    fit = _hip.FitSession(ctx, b, off, 2)
    try:
        sums, counts = fit.segment_means()
    finally:
        fit.close()
        b.close()
    np.testing.assert_array_equal(counts, lens)
    for s, n in enumerate(lens):
        want = np.add.reduce(X[off[s]:off[s + 1]], axis=0) if n else np.zeros(D)
        np.testing.assert_array_equal(sums[s], want, err_msg="state %d (%d frames)" % (s, n))
        if n:
            with np.errstate(invalid="ignore"):
                np.testing.assert_array_equal(sums[s] / counts[s], np.mean(X[off[s]:off[s + 1]], axis=0))


@pytest.mark.parametrize("max_iteration", [1, 6])
def test_component_on_one_frame_raises_like_the_reference(max_iteration):
    """A component whose weight ends up on ONE frame: the reference's M-step gives mu = that frame and a variance of exactly 0,
    and `update_models` -> the covariance setter raises LinAlgError (hmm_state.py:24-30, 134-149) -- in the iteration it happens,
    also when that is the state's last one.  The device forms sum centred moments, which leave rounding noise there: the
    streaming update recognises the one-point component (sum r y^2 . sum r = (sum r y)^2 in every dimension)."""
    rng = np.random.default_rng(8)
    D, k = 5, 2
    body = rng.normal(size=(60, D))
    far = np.full((1, D), 60.0) + rng.normal(size=(1, D))       # one frame ~60 standard deviations out
    seg = np.concatenate([body[:30], far, body[30:]])
    other = rng.normal(size=(200, D)) * 2.0 + 3.0                # a healthy state beside it
    segs = [seg, other]
    n = np.array([len(x) for x in segs], dtype=np.float64)
    mean0 = np.stack([np.stack([body.mean(axis=0), far[0] - 0.5]), np.stack([other.mean(axis=0) - 1.0, other.mean(axis=0) + 1.0])])
    var0 = np.ones((2, k, D))
    w0 = np.full((2, k), 0.5)
    with pytest.raises(np.linalg.LinAlgError):                   # the oracle = the reference's arithmetic
        m, v, w = mean0[0].copy(), var0[0].copy(), w0[0].copy()
        O.gmm_em(seg, m, v, w, k, max_iteration=max_iteration, old=(np.zeros((k, D)), np.ones((k, D)), np.zeros(k)))
    for tiles in (False, True):
        b, fit, off = _session(segs, k, tiles=tiles)
        try:
            mean, var, w = mean0.copy(), var0.copy(), w0.copy()
            with pytest.raises(np.linalg.LinAlgError):
                fit.em(k, mean, var, w, np.zeros_like(mean), np.ones_like(mean), np.zeros_like(w), n, max_iteration=max_iteration)
        finally:
            fit.close()
            b.close()


def test_gmm_em_object_raises_for_a_component_on_one_frame():
    """The same collapse through the object API (`GMM.em`: E-step statistics from the device, M-step on the host): the
    covariance setter raises at the collapsed component, the component in front of it already updated."""
    import sr.recognition as R
    rng = np.random.default_rng(8)
    D, k = 5, 2
    body = rng.normal(size=(60, D))
    far = np.full((1, D), 60.0) + rng.normal(size=(1, D))
    seg = np.concatenate([body[:30], far, body[30:]])
    g = R.GMM(body.mean(axis=0), np.ones(D), k)
    g.update_models(np.stack([body.mean(axis=0), far[0] - 0.5]), np.ones((k, D)), np.full(k, 0.5))
    before = np.array(g.dists[0].mean)
    from test_gpu_api import quiet
    with quiet(), pytest.raises(np.linalg.LinAlgError):
        g.em(seg, k, max_iteration=4)
    assert not np.array_equal(np.asarray(g.dists[0].mean), before)          # component 0 was installed first
    np.testing.assert_array_equal(np.asarray(g.dists[1].cov), np.zeros(D))    # the value the setter refused to invert


def test_zero_variance_in_a_states_last_iteration_raises():
    """A feature that is 0 in every frame: the M-step's variance is exactly 0 in the FIRST update, and with max_iteration = 1
    that is also the last one -- the reference raises there (update_models runs before the convergence test)."""
    rng = np.random.default_rng(9)
    D, k = 4, 2
    seg = rng.normal(size=(80, D))
    seg[:, 2] = 0.0
    n = np.array([80.0])
    mean0 = np.stack([np.stack([seg.mean(axis=0) - 0.5, seg.mean(axis=0) + 0.5])])
    mean0[0, :, 2] = 0.0
    var0, w0 = np.ones((1, k, D)), np.full((1, k), 0.5)
    with pytest.raises(np.linalg.LinAlgError):
        O.gmm_em(seg, mean0[0].copy(), var0[0].copy(), w0[0].copy(), k, max_iteration=1, old=(np.zeros((k, D)), np.ones((k, D)), np.zeros(k)))
    for tiles in (False, True):
        b, fit, off = _session([seg], k, tiles=tiles)
        try:
            mean, var, w = mean0.copy(), var0.copy(), w0.copy()
            with pytest.raises(np.linalg.LinAlgError):
                fit.em(k, mean, var, w, np.zeros_like(mean), np.ones_like(mean), np.zeros_like(w), n, max_iteration=1)
        finally:
            fit.close()
            b.close()


def test_constant_feature_is_singular_whatever_the_rounding():
    """A feature with the same non-zero value in every frame, the start means beside it: the reference's variance in that
    dimension is exactly 0 or, when sum(p x) / sum(p) lands an ulp beside x, ~1e-31 (LinAlgError now or NaN later); the sums
    centred on the shift point give +-1e-16 y^2 -- within the noise, and the streaming update raises."""
    rng = np.random.default_rng(11)
    D, k = 4, 2
    seg = rng.normal(size=(120, D))
    seg[:, 1] = 3.7
    mean0 = np.stack([np.stack([seg.mean(axis=0) - 0.5, seg.mean(axis=0) + 0.5])])
    mean0[0, :, 1] = [3.4, 3.9]
    var0, w0 = np.ones((1, k, D)), np.full((1, k), 0.5)
    m, v, w = mean0[0].copy(), var0[0].copy(), w0[0].copy()
    try:
        O.gmm_em(seg, m, v, w, k, max_iteration=1, old=(np.zeros((k, D)), np.ones((k, D)), np.zeros(k)))
        assert v[:, 1].max() < 1e-25                              # (not exactly 0 this time: an ulp beside the constant)
    except np.linalg.LinAlgError:
        pass
    b, fit, off = _session([seg], k)
    try:
        mean, var, ww = mean0.copy(), var0.copy(), w0.copy()
        with pytest.raises(np.linalg.LinAlgError):
            fit.em(k, mean, var, ww, np.zeros_like(mean), np.ones_like(mean), np.zeros_like(ww), np.array([120.0]), max_iteration=3)
    finally:
        fit.close()
        b.close()


def test_partition_variances_of_tiny_groups_are_np_covs():
    """Random partitions of SMALL states: a group of two frames that lie close together has a variance of 1e-8 of the spread; the
    one-pass sums of the streaming pass keep 7-9 digits of it, so states of up to 64 frames take two passes (np.cov's numbers)."""
    rng = np.random.default_rng(10)
    D, k = 6, 3
    small = rng.normal(size=(7, D)) * 2.0 + 5.0
    small[4] = small[1] + rng.normal(size=D) * 1e-4             # frames 1 and 4: close together
    big = rng.normal(size=(500, D)) + 1.0
    part = np.concatenate([[0, 1, 2, 0, 1, 2, 0], rng.integers(0, k, size=500)]).astype(np.uint8)
    c0 = np.stack([np.stack([x.mean(axis=0) + 0.1 * j for j in range(k)]) for x in (small, big)])
    covs = []
    for tiles in (False, True):
        b, fit, off = _session([small, big], k, tiles=tiles)
        try:
            covs.append(fit.kmeans(k, c0, part, max_iteration=3)[1])
        finally:
            fit.close()
            b.close()
    want = np.array([np.cov(small[part[:7] == c].T).diagonal() for c in range(k)])
    np.testing.assert_allclose(covs[0][0], want, rtol=1e-14)
    np.testing.assert_allclose(covs[0][0], covs[1][0], rtol=5e-16)     # the tile path's two passes (an fma more or less)
    np.testing.assert_allclose(covs[0][1], covs[1][1], rtol=1e-12)
    assert covs[0][0][1].max() < 1e-7                            # (the group of the two close frames)


def test_tail_launches_change_nothing(capfd):
    """The multi-iteration (TAIL) launches -- up to 64 lock-step iterations inside one launch once the states still active fit
    the chip -- against one launch per iteration (GMMHMM_REFIT_TAIL=0): assignments, iteration counts, centroids, mixtures and
    converged_at are the same BITS; and the session's own count says the tail launches really ran (GMMHMM_REFIT_DEBUG)."""
    import re
    rng = np.random.default_rng(21)
    D, k = 13, 4
    # (clusters that overlap: k-means and EM take tens of iterations, the tail launches start behind the first poll at 8)
    segs = _mixture_data(rng, (3000, 2500, 1300, 900, 4000, 400), D, k, spread=0.6)
    N = sum(len(x) for x in segs)
    part = rng.integers(0, k, size=N).astype(np.uint8)
    c0 = np.stack([np.stack([x.mean(axis=0) * f for f in np.linspace(0.97, 1.03, k)]) for x in segs])
    res, counts = [], []
    for tail in ("1", "0"):
        os.environ["GMMHMM_REFIT_TAIL"] = tail
        os.environ["GMMHMM_REFIT_DEBUG"] = "1"
        try:
            b, fit, off = _session(segs, k)
            cen, cov, cnt, its = fit.kmeans(k, c0, part, max_iteration=200)
            ids = fit.clusters()
            n = np.diff(off).astype(np.float64)
            mean, var, w = cen.copy(), cov.copy(), cnt / n[:, None]
            conv = fit.em(k, mean, var, w, np.zeros_like(mean), np.ones_like(mean), np.zeros_like(w), n, max_iteration=300)
            capfd.readouterr()
            fit.close()
            b.close()
            err = capfd.readouterr().err
        finally:
            os.environ.pop("GMMHMM_REFIT_TAIL", None)
            os.environ.pop("GMMHMM_REFIT_DEBUG", None)
        m = re.search(r"(\d+) tail launches \((\d+) iterations\), (\d+) refused, (\d+) ordinary launches", err)
        assert m, err
        counts.append(tuple(int(v) for v in m.groups()))
        res.append(dict(cen=cen, cov=cov, cnt=cnt, its=its, ids=ids, mean=mean, var=var, w=w, conv=conv))
    assert counts[0][0] > 0 and counts[0][1] > counts[0][0]         # several iterations per tail launch
    assert counts[1][0] == 0 and counts[1][3] > counts[0][3]        # none when switched off: more ordinary launches instead
    # the same BITS: the sums run in a fixed order in both forms, and gh_refit_mfma.hip is built with -ffp-contract=on, so that
    # the two instantiations of a kernel contract the same multiply-adds (hipcc's default left the EM 1e-15 apart)
    for name in res[0]:
        np.testing.assert_array_equal(res[0][name], res[1][name], err_msg=name)
    assert res[0]["its"].max() > 8 and (res[0]["conv"] >= 0).any()
