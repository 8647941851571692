# -*- coding: utf-8 -*-
"""Word model: a left-to-right chain of mixture states (mirror of the reference's
`sr/recognition/hmm.py`; attribute names kept so pickled models keep their layout:
gmm_states, mu, n_segments, segments, sigma, transitions, use_em, use_gmm)."""
import numpy as np

from .decode import dtw, decode_hmm_states
from .kmeans import kmeans, skmeans, align_gmm_states
from .hmm_state import GMM, mahalanobis

__all__ = ["HMM"]


class HMM:
    """
    Attributes
    ----------
    mu, sigma: per-state mean / variance [n_segments, D] (single-Gaussian model)
    transitions: cost matrix [n_segments, n_segments], [i, j] = cost of j -> i
    segments: training frames of every state (list of [N_s, D])
    gmm_states: list of GMM states when `use_gmm`
    """

    def __init__(self, n_segments):
        self.n_segments = n_segments
        self.mu = None
        self.sigma = None
        self.transitions = None
        self.segments = []
        self.gmm_states = None
        self.use_gmm = True
        self.use_em = True

    def __eq__(self, other):
        """Convergence test of continuous training (hmm.py:30-41): mixtures allclose;
        transitions are not compared."""
        if not self.use_gmm:
            return np.allclose(self.mu, other.mu) and np.allclose(self.sigma, other.sigma)
        if not other.use_gmm or self.n_segments != other.n_segments:
            return False
        return all(not (a != b) for a, b in zip(self.gmm_states[:self.n_segments], other.gmm_states))

    def reset(self):
        self.mu = None
        self.sigma = None
        self.transitions = None
        self.segments = []
        self.gmm_states = None

    def __getitem__(self, item):
        assert self.use_gmm == True
        if type(item) is int or type(item) is slice:
            return self.gmm_states[item]
        raise TypeError('The type of index is not supported')

    def fit(self, ys, n_gaussians, use_gmm=True, use_em=True):
        """Train on a list of [T_u, D] templates (hmm.py:57-76).

        use_gmm: mixture per state (segmental k-means, then split-k-means [+ EM] per state,
        then re-alignment); otherwise one Gaussian per state from segmental k-means alone.
        n_gaussians: size of every mixture; int(ln(n_gaussians)) binary splits are trained
        (hmm.py:104), the remaining components keep the state's initial Gaussian."""
        self.use_em = use_em
        self.use_gmm = use_gmm
        if use_gmm:
            self.fit_GMM(ys, n_gaussians)
        else:
            self.mu, self.sigma, self.transitions, self.segments = skmeans(ys, self.n_segments,
                                                                           return_segmented_data=True)
        return self

    def _init_gmm(self, n_gaussians):
        self.gmm_states = [GMM(self.mu[i, :], self.sigma[i, :], n_gaussians) for i in range(self.n_segments)]

    def fit_GMM(self, ys, n_gaussians):
        print('Doing segmental k-means')
        self.mu, self.sigma, self.transitions, self.segments = skmeans(ys, self.n_segments,
                                                                       return_segmented_data=True)
        self._init_gmm(n_gaussians)
        for i, seg in enumerate(self.segments):
            self._fit_GMM(seg, n_gaussians, i)
        self.segments = align_gmm_states(ys, self.gmm_states, self.transitions, self.n_segments)

    def _fit_GMM(self, data, n_gaussians, seg_i):
        """Binary-split k-means (+ EM) of one state (hmm.py:97-124)."""
        n_splits = int(np.log(n_gaussians))
        assert n_splits > 0
        state = self.gmm_states[seg_i]
        n = data.shape[0]
        centroids = np.array([self.mu[seg_i, :]])
        weights = np.full(n_gaussians, 1 / n)
        for i in range(n_splits):
            k = 2 ** (i + 1)
            centroids = np.concatenate([centroids * 0.9, centroids * 1.1], axis=0)
            clusters, centroids, variance = kmeans(data, k, centroids, dist_fun=mahalanobis)
            ids, counts = np.unique(clusters, return_counts=True)
            for c in ids:
                weights[c] = counts[c] / n  # counts looked up by cluster ID, as in hmm.py:116-118
            state.update_models(centroids, variance, weights[:k])
            if self.use_em:
                state.em(data, k)

    def evaluate(self, x):
        """Cost of the best alignment of `x` ending in the last state at the last frame
        (hmm.py:126-135)."""
        if self.use_gmm:
            costs, _ = decode_hmm_states(x, self.gmm_states, self.transitions)
        else:
            costs, _ = dtw(x, self.mu, mahalanobis, self.transitions, self.sigma)
        return costs[-1, -1]
