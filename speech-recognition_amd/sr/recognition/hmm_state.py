# -*- coding: utf-8 -*-
"""Emission models of the GMM-HMM core, evaluated on the MI355X.

Mirror of the reference's `sr/recognition/hmm_state.py` (same class / function
names, signatures, attribute names -- so pickled models keep their layout -- and
exceptions).  Every density evaluation runs in HIP through libgmmhmm
(`_hip.py`); this file only holds parameters and the O(M) / O(k*D) host glue
(exp/sum of M component values, the M-step division, convergence tests).
"""
import copy
import uuid

import numpy as np

from . import _hip
from . import _pack

__all__ = ["MultivariateNormal", "mahalanobis", "HMMState", "NES", "GMM"]


def _ctx():
    return _hip.default_context()


def _deepcopy_arrays_directly(self, memo):
    """copy.deepcopy of a parameter object without the generic walk: numpy arrays are copied with .copy() (what
    ndarray.__deepcopy__ does, aliasing between objects kept through `memo` as deepcopy keeps it), everything else goes
    through copy.deepcopy with the same memo.  continuous_train deep-copies every model once per outer iteration
    (continuous_speech.py:62,101): 400 MultivariateNormal objects for ten 5-state 8-mixture words -- 10 ms of generic
    deepcopy, 2 ms like this; same result, same `id` (so the copies still hash like the originals)."""
    new = self.__class__.__new__(self.__class__)
    memo[id(self)] = new
    d = new.__dict__
    for k, v in self.__dict__.items():
        if type(v) is np.ndarray and v.dtype != object:
            y = memo.get(id(v))
            if y is None:
                y = v.copy()
                memo[id(v)] = y
            d[k] = y
        else:
            d[k] = copy.deepcopy(v, memo)
    return new


class MultivariateNormal:
    """Diagonal-covariance Gaussian (reference: hmm_state.py:5-45).

    Attributes `mean`, `_cov`, `inv_cov` as in the reference; assigning `cov`
    refreshes `inv_cov` and raises numpy.linalg.LinAlgError for a singular
    covariance, like the reference's np.linalg.inv does (hmm_state.py:17,30).
    """

    def __init__(self, mean, cov):
        self.mean = mean
        self.cov = cov

    __deepcopy__ = _deepcopy_arrays_directly

    @property
    def cov(self):
        return self._cov

    @cov.setter
    def cov(self, val):
        self._cov = val
        c = np.asarray(val)
        if c.ndim == 1:
            if np.any(c == 0):
                raise np.linalg.LinAlgError("Singular matrix")
            self.inv_cov = np.diag(1.0 / c)
        else:
            self.inv_cov = np.linalg.inv(c)

    @cov.deleter
    def cov(self):
        del self._cov

    def _diag(self):
        c = np.asarray(self._cov, dtype=np.float64)
        return c if c.ndim == 1 else np.diag(c)

    def pdf(self, x):
        """Density at one frame, linear domain (hmm_state.py:36-45)."""
        x = np.asarray(x, dtype=np.float64)
        mean = np.asarray(self.mean, dtype=np.float64)
        if x.shape[0] != mean.shape[0]:
            raise NameError("The dimensions of the input don't match")
        model = _pack.device_normal(_ctx(), mean, self._diag())      # cached on (mean, variance): pdf is called per frame
        return np.exp(model.component_loglik(0, x[None, :])[0, 0])


def mahalanobis(v1, v2, variance):
    """Diagonal-Gaussian negative log-likelihood of v1 under N(v2, diag(variance))
    (hmm_state.py:48-58).  Passing this function as `dist_fun` to dtw / kmeans /
    skmeans selects the fused HIP distance kernels."""
    v1 = np.asarray(v1, dtype=np.float64)
    v2 = np.asarray(v2, dtype=np.float64)
    var = np.asarray(variance, dtype=np.float64)
    return _hip.distance_matrix(_ctx(), v1[None, :], v2[None, :], var[None, :])[0, 0]


def euclidean(*args):
    """||args[0] - args[1]||: the default `dist_fun` of dtw / kmeans / skmeans in the
    reference (an anonymous lambda there, kmeans.py:111,167).  Recognised by identity and
    run in HIP inside the batched callers."""
    a = np.asarray(args[0], dtype=np.float64)
    b = np.asarray(args[1], dtype=np.float64)
    return _hip.distance_matrix(_ctx(), a[None, :], b[None, :])[0, 0]


def _allclose(a, b):
    """np.allclose(a, b) (the comparison of hmm_state.py:161-170) without its general-purpose set-up: for finite arrays
    |a - b| <= 1e-8 + 1e-5 |b| element by element IS numpy's test; anything else (nan, inf, shapes that do not
    broadcast the plain way) goes to np.allclose itself."""
    a, b = np.asarray(a), np.asarray(b)
    if a.shape == b.shape and a.dtype == np.float64 and b.dtype == np.float64:
        with np.errstate(invalid="ignore"):                         # (inf - inf: np.isclose is silent about it too)
            if (np.abs(a - b) <= 1e-8 + 1e-5 * np.abs(b)).all():
                return True
        if np.isfinite(a).all() and np.isfinite(b).all():
            return False
    return bool(np.allclose(a, b))


class HMMState:
    """Base class of all HMM states (hmm_state.py:61-78)."""

    def __init__(self):
        self.id = uuid.uuid4().int
        self.parent = None

    __deepcopy__ = _deepcopy_arrays_directly

    def evaluate(self, x):
        raise NotImplemented()

    def __eq__(self, other):
        raise NotImplemented()

    def __hash__(self):
        return hash(self.id)


class NES(HMMState):
    """Non-emitting state: costs nothing, consumes no frame (hmm_state.py:81-97)."""

    def __init__(self):
        super().__init__()

    def evaluate(self, x):
        return 0

    def __eq__(self, other):
        return self.id == other.id

    def __hash__(self):
        return hash(self.id)


class GMM(HMMState):
    """Gaussian mixture state (hmm_state.py:100-179).

    `n_gaussians` components start as copies of N(mu, diag(sigma)) with weight
    1/n; `update_models` / `em` overwrite the first k of them.  Weights are used
    as they are, never renormalised (hmm_state.py:115).
    """

    def __init__(self, mu, sigma, n_gaussians):
        super().__init__()
        self.n_gaussians = n_gaussians
        self.w = np.full(n_gaussians, 1 / n_gaussians)
        # n_gaussians copies of N(mu, diag(sigma)) (hmm_state.py:108): the singular-covariance check and the inverse are
        # worked out once, every copy gets an inverse of its own (400 np.diag calls for ten 5-state 8-mixture words were 2 ms)
        first = MultivariateNormal(mean=mu, cov=sigma)
        self.dists = [first]
        for _ in range(n_gaussians - 1):
            d = MultivariateNormal.__new__(MultivariateNormal)
            d.__dict__.update(mean=mu, _cov=sigma, inv_cov=first.inv_cov.copy())
            self.dists.append(d)
        self.mu_old = np.tile(mu, (n_gaussians, 1))
        self.sigma_old = np.tile(sigma, (n_gaussians, 1))
        self.w_old = np.full(n_gaussians, 1 / n_gaussians)

    # ---- device side -------------------------------------------------------
    def _device(self):
        return _pack.device_gmm(_ctx(), [self])

    def component_log_density(self, X):
        """log(w_m * pdf_m(x_n)) for every frame and component: [N, n_gaussians] (HIP)."""
        X = np.asarray(X, dtype=np.float64)
        if X.ndim != 2 or X.shape[1] != np.asarray(self.dists[0].mean).shape[0]:
            raise NameError("The dimensions of the input don't match")
        return self._device().component_loglik(0, X)

    def evaluate(self, x, return_neg_log_likelihood=True):
        """-log sum_m w_m pdf_m(x), or the vector of weighted densities (hmm_state.py:114-120).
        The linear-domain sum is kept so that a frame whose every component underflows
        costs +inf exactly like the reference."""
        res = np.exp(self.component_log_density(np.asarray(x, dtype=np.float64)[None, :])[0])
        if return_neg_log_likelihood:
            with np.errstate(divide="ignore"):
                return -np.log(res.sum())
        return res

    def evaluate_batch(self, X):
        """Negative log-likelihood of every row of X[N,D] (batched form of `evaluate`)."""
        with np.errstate(divide="ignore"):
            return -np.log(np.exp(self.component_log_density(X)).sum(axis=1))

    # ---- training ----------------------------------------------------------
    def em_update(self, stats, n, k, it):
        """M-step + convergence test of one EM iteration (hmm_state.py:134-159) from the E-step statistics
        `stats` [k, 1+2D] of `gh_em_accumulate` (occupancy, first and second moments around the CURRENT means) over
        the state's `n` frames.  Returns True when the reference's loop would `break` (parameters allclose to the
        previous iteration's); like there, the new parameters are installed before the test."""
        dim = (stats.shape[1] - 1) // 2
        means = np.array([np.asarray(d.mean, dtype=np.float64) for d in self.dists[:k]])
        occ = stats[:, 0].copy()
        weights = occ / n                      # p.mean(axis=0)            (:148)
        occ[occ == 0] = 10 ** (-5)             # avoid divide by 0        (:134-136)
        # statistics are centred on the current means m0:  S1 = sum r (x - m0), S2 = sum r (x - m0)^2
        mu = (means * stats[:, 0][:, None] + stats[:, 1:1 + dim]) / occ[:, None]   # sum r x / N
        # sum r (x - mu)^2 / N around the NEW mean (:141-143), from the centred sums
        delta = mu - means
        sigma = (stats[:, 1 + dim:] - delta * (2.0 * stats[:, 1:1 + dim] - delta * stats[:, 0][:, None])) / occ[:, None]
        # A component with all its weight on ONE frame: the reference's two-pass variance is exactly 0 there and the covariance
        # setter raises (hmm_state.py:24-30); the centred sums leave rounding noise (+-1e-16 of the spread).  One point <=>
        # sum r y^2 . sum r = (sum r y)^2 in every dimension (to the rounding of the sums): such a row gets its exact 0, and
        # update_models raises at that component with the ones in front installed, as in the reference.
        with np.errstate(all="ignore"):
            # (as weighted mean and mean square: for a component nobody is close to, s0 ~ 1e-200, products of the sums underflow)
            s0 = stats[:, 0][:, None]
            mq, qq = stats[:, 1:1 + dim] / s0, stats[:, 1 + dim:] / s0
            one_point = (stats[:, 0] >= 1e-290) & np.all(np.abs(mq * mq - qq) <= 3.6e-15 * np.abs(qq), axis=1)
        if one_point.any():
            sigma = np.where(one_point[:, None], 0.0, sigma)
        with np.errstate(invalid="ignore"):
            sigma = np.where(sigma < 0, 0.0, sigma)          # below the rounding noise of the centred sums: numerically singular
        self.update_models(mu, sigma, weights)
        if np.allclose(mu, self.mu_old[:k, :]) and np.allclose(sigma, self.sigma_old[:k, :]) \
                and np.allclose(weights, self.w_old[:k]):
            print("EM converged at iteration:", it)
            return True
        print("EM iteration:", str(it), end="\r", flush=True)
        self.mu_old[:k, :] = mu
        self.sigma_old[:k, :] = sigma
        self.w_old[:k] = weights
        return False

    def em(self, data, n_gaussians, max_iteration=10000):
        """EM on the first `n_gaussians` components over frames hard-assigned to this
        state (hmm_state.py:122-159).  E-step statistics come from the HIP kernel
        (`gh_em_accumulate`: N_c and the first / second moments around the current means, one pass
        over the resident frames); the M-step, `update_models` and the allclose convergence test stay
        on the host (`em_update`).  `lockstep.LockstepFitter` runs the same iteration for many states per launch."""
        data = np.ascontiguousarray(data, dtype=np.float64)
        k = n_gaussians
        n = data.shape[0]
        frames = _hip.Batch(_ctx(), feats=data, offsets=[0, n])
        try:
            for it in range(max_iteration):
                means, vars_, w = _pack.gmm_arrays(self)
                stats, _ = frames.em_accumulate(means[:k], vars_[:k], w[:k])
                if self.em_update(stats, n, k, it):
                    break
        finally:
            frames.close()

    def update_models(self, mus, sigmas, weights):
        k = mus.shape[0]
        self.w[:k] = weights
        sig = np.asarray(sigmas)
        if sig.ndim == 2 and k:
            # diagonal covariances (every caller of this package): what assigning `mean` / `cov` component after component
            # does (hmm_state.py:29-30), with the inverses filled in one step (np.diag per component was 1.2 ms per outer
            # iteration of continuous_train).  A zero variance raises at ITS component, the ones in front already updated,
            # as in the reference
            zero = np.flatnonzero((sig == 0).any(axis=1))
            upto = int(zero[0]) if len(zero) else k
            D = sig.shape[1]
            inv = np.zeros((upto, D, D))
            idx = np.arange(D)
            inv[:, idx, idx] = 1.0 / sig[:upto]
            for m in range(upto):
                self.dists[m].__dict__.update(mean=mus[m, :], _cov=sigmas[m, :], inv_cov=inv[m])
            if upto < k:
                self.dists[upto].mean = mus[upto, :]
                self.dists[upto].cov = sigmas[upto, :]            # raises numpy.linalg.LinAlgError
            return
        for m in range(k):
            self.dists[m].mean = mus[m, :]
            self.dists[m].cov = sigmas[m, :]

    def __eq__(self, other):
        if self.n_gaussians != other.n_gaussians or not _allclose(self.w, other.w):
            return False
        return all(_allclose(a.mean, b.mean) and _allclose(a.cov, b.cov)
                   for a, b in zip(self.dists, other.dists))

    def __len__(self):
        return self.n_gaussians

    def __hash__(self):
        return hash(self.id)
