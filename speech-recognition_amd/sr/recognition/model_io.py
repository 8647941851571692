# -*- coding: utf-8 -*-
"""Model wire formats.

* The reference persists models with `pickle` (continuous_speech.py:167-170, core.py:57): because this package
  keeps the reference's module paths, class names and attribute names, its pickles load here and the pickles
  written here (`continuous_train`, `save_models_pickle`) load in the reference
  (tests/test_pickle_reference_roundtrip.py runs that round trip against the reference itself).
* `save_models_npz` / `load_models_npz`: the packed arrays the GPU path consumes -- means / variances / weights
  [W, n, M, D], transition costs [W, n, n], single-Gaussian mu / sigma [W, n, D] -- one file for a whole
  vocabulary, no code objects inside.
"""
import pickle

import numpy as np

from .hmm import HMM
from .hmm_state import GMM

__all__ = ["save_models_pickle", "load_models_pickle", "save_models_npz", "load_models_npz", "models_from_arrays"]


def save_models_pickle(path, models, protocol=2):
    """One pickle holding the list of HMM objects (protocol 2 by default: readable by every Python 3)."""
    with open(path, "wb") as f:
        pickle.dump(list(models), f, protocol=protocol)


def load_models_pickle(path):
    with open(path, "rb") as f:
        return pickle.load(f)


def save_models_npz(path, models):
    """Pack a vocabulary of equally shaped GMM word models into one .npz."""
    models = list(models)
    n = models[0].n_segments
    assert all(m.use_gmm and m.n_segments == n and len(m.gmm_states) == n for m in models), \
        "npz packing needs GMM models with the same number of states"
    M = models[0].gmm_states[0].n_gaussians
    assert all(g.n_gaussians == M for m in models for g in m.gmm_states), "all mixtures must have the same size"
    means = np.array([[[d.mean for d in g.dists] for g in m.gmm_states] for m in models], dtype=np.float64)
    vars_ = np.array([[[d.cov for d in g.dists] for g in m.gmm_states] for m in models], dtype=np.float64)
    w = np.array([[g.w for g in m.gmm_states] for m in models], dtype=np.float64)
    trans = np.array([m.transitions for m in models], dtype=np.float64)
    extra = {}
    if all(m.mu is not None and m.sigma is not None for m in models):
        extra = dict(mu=np.array([m.mu for m in models]), sigma=np.array([m.sigma for m in models]))
    np.savez_compressed(path, format=np.array("gmmhmm-npz-1"), means=means, vars=vars_, weights=w, transitions=trans,
                        use_em=np.array([bool(m.use_em) for m in models]), **extra)


def models_from_arrays(means, vars_, weights, transitions, mu=None, sigma=None, use_em=True):
    """HMM objects (fresh state ids) from packed arrays: means / vars_ [W, n, M, D], weights [W, n, M],
    transitions [W, n, n] (or a list of W matrices)."""
    means, vars_, weights = (np.asarray(a, dtype=np.float64) for a in (means, vars_, weights))
    W, n, M, D = means.shape
    out = []
    for i in range(W):
        h = HMM(n)
        h.use_gmm = True
        h.use_em = bool(use_em if np.isscalar(use_em) else use_em[i])
        h.gmm_states = []
        for s in range(n):
            g = GMM(means[i, s, 0].copy(), vars_[i, s, 0].copy(), M)
            g.update_models(means[i, s].copy(), vars_[i, s].copy(), weights[i, s].copy())
            g.parent = h
            h.gmm_states.append(g)
        h.transitions = np.array(transitions[i], dtype=np.float64)
        if mu is not None and sigma is not None:
            h.mu, h.sigma = np.array(mu[i], dtype=np.float64), np.array(sigma[i], dtype=np.float64)
        out.append(h)
    return out


def load_models_npz(path):
    """Inverse of `save_models_npz`: list of HMM objects (fresh state ids)."""
    z = np.load(path, allow_pickle=False)
    assert str(z["format"]) == "gmmhmm-npz-1", "not a gmmhmm npz model file"
    has_mu = "mu" in z.files
    return models_from_arrays(z["means"], z["vars"], z["weights"], z["transitions"], z["mu"] if has_mu else None,
                              z["sigma"] if has_mu else None, use_em=z["use_em"])
