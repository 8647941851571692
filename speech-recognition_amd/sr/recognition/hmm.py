# -*- coding: utf-8 -*-
"""Word model of the recogniser: `n_segments` left-to-right states with a cost matrix between them.

API and pickled layout are the reference's (`sr/recognition/hmm.py:8-135`: the instance dictionary holds exactly
gmm_states, mu, n_segments, segments, sigma, transitions, use_em, use_gmm, so models trained by either side load in
the other), the training path is not: after the segmental k-means of `skmeans` ALL states of the word are refit
together -- `lockstep.LockstepFitter` gathers their frames into one resident batch and advances every state's
split-k-means / EM with one launch per iteration -- where the reference fits state after state (hmm.py:97-124).
"""
import numpy as np

from . import decode as _decode
from . import kmeans as _km
from .hmm_state import GMM, mahalanobis
from .lockstep import LockstepFitter

__all__ = ["HMM"]

# what `reset` forgets: everything learnt from data (n_segments and the two switches stay)
_LEARNT = {"mu": None, "sigma": None, "transitions": None, "segments": list, "gmm_states": None}


class HMM:
    """
    n_segments   number of states
    mu, sigma    [n_segments, D] state means / variances of the single-Gaussian model (also the seeds of the mixtures)
    transitions  [n_segments, n_segments] costs, entry [i, j] = cost of moving j -> i, +inf = no arc
    segments     per state, the training frames aligned to it (list of [N_s, D])
    gmm_states   list of `GMM` states when `use_gmm`
    use_gmm / use_em   what `fit` was asked to train
    """

    def __init__(self, n_segments):
        self.n_segments = n_segments
        self.reset()
        self.use_gmm = self.use_em = True

    def reset(self):
        for name, blank in _LEARNT.items():
            setattr(self, name, blank() if callable(blank) else blank)

    # ---- comparison / access ------------------------------------------------------------------------------
    def __eq__(self, other):
        """hmm.py:30-41 -- what `continuous_train` calls convergence: every state's mixture allclose to the other
        model's; the transition costs play no part."""
        if not self.use_gmm:
            return bool(np.allclose(self.mu, other.mu) and np.allclose(self.sigma, other.sigma))
        if not (other.use_gmm and self.n_segments == other.n_segments):
            return False
        mine, theirs = self.gmm_states, other.gmm_states
        return not any(mine[i] != theirs[i] for i in range(self.n_segments))

    def __getitem__(self, item):
        assert self.use_gmm == True  # noqa: E712 (the reference's own comparison: a non-bool truthy flag fails it too)
        if type(item) not in (int, slice):
            raise TypeError('The type of index is not supported')
        return self.gmm_states[item]

    # ---- training -----------------------------------------------------------------------------------------
    def fit(self, ys, n_gaussians, use_gmm=True, use_em=True):
        """Train on the templates `ys` (list of [T_u, D]; hmm.py:57-76).  use_gmm=False stops after the segmental
        k-means (one Gaussian per state, scored by `dtw`); otherwise every state gets a mixture of `n_gaussians`
        components of which int(ln n_gaussians) binary splits are trained (hmm.py:104) -- by k-means alone or, with
        use_em, k-means followed by EM."""
        self.use_gmm, self.use_em = use_gmm, use_em
        if use_gmm:
            self.fit_GMM(ys, n_gaussians)
        else:
            self._segmental_kmeans(ys)
        return self

    def _segmental_kmeans(self, ys):
        self.mu, self.sigma, self.transitions, self.segments = _km.skmeans(ys, self.n_segments, return_segmented_data=True)

    def fit_GMM(self, ys, n_gaussians):
        """Segmental k-means, then the mixtures of all states (hmm.py:81-95), then a re-alignment of the templates
        against the trained mixtures."""
        print('Doing segmental k-means')
        self._segmental_kmeans(ys)
        self.gmm_states = [GMM(m, s, n_gaussians) for m, s in zip(self.mu, self.sigma)]
        fitter = LockstepFitter(self.segments)
        try:
            fitter.split_and_fit(self.gmm_states, start_centroids=self.mu,
                                 weight_divisor=[len(seg) for seg in self.segments],     # hmm.py:108,118
                                 n_gaussians=n_gaussians, use_em=self.use_em)
        finally:
            fitter.close()
        self.segments = _km.align_gmm_states(ys, self.gmm_states, self.transitions, self.n_segments)

    def _fit_GMM(self, data, n_gaussians, seg_i):
        """The refit of ONE state on `data` (hmm.py:97-124) -- the lock-step fitter with a single segment."""
        assert int(np.log(n_gaussians)) > 0
        fitter = LockstepFitter([data])
        try:
            fitter.split_and_fit([self.gmm_states[seg_i]], start_centroids=[self.mu[seg_i]], weight_divisor=[len(data)],
                                 n_gaussians=n_gaussians, use_em=self.use_em)
        finally:
            fitter.close()

    # ---- scoring ------------------------------------------------------------------------------------------
    def evaluate(self, x):
        """Cost of the cheapest alignment of `x` that ends in the last state on the last frame (hmm.py:126-135)."""
        if self.use_gmm:
            costs = _decode.decode_hmm_states(x, self.gmm_states, self.transitions)[0]
        else:
            costs = _decode.dtw(x, self.mu, mahalanobis, self.transitions, self.sigma)[0]
        return costs[-1, -1]
