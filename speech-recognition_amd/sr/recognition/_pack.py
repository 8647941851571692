# -*- coding: utf-8 -*-
"""Object graph -> packed device structures, with small content-addressed caches.

The reference API hands over Python objects (lists of GMM / NES states, dense
transition matrices with +inf holes).  The kernels want structure-of-arrays
models and arc lists.  Everything here is host-side bookkeeping; no likelihood
or DP arithmetic happens in this file.
"""
import hashlib
import threading
from collections import OrderedDict

import numpy as np

from . import _hip


_NES = None


def is_nes(state):
    global _NES
    if _NES is None:          # (hmm_state imports this module: resolved on first use, not once per row)
        from .hmm_state import NES
        _NES = NES
    return type(state) == _NES  # the reference tests the exact type (decode.py:109)


def gmm_arrays(g):
    """(means[M,D], vars[M,D], w[M]) of one GMM object, as stored in its dists."""
    means = np.array([np.asarray(d.mean, dtype=np.float64) for d in g.dists])
    vars_ = np.array([np.asarray(d.cov, dtype=np.float64) for d in g.dists])
    if vars_.ndim == 3:  # full matrices handed to MultivariateNormal: keep their diagonals
        vars_ = np.array([np.diag(v) for v in vars_])
    return means, vars_, np.asarray(g.w, dtype=np.float64).copy()


def pack_states(states):
    """row_state[R] (-1 for non-emitting rows) + stacked parameters of the DISTINCT
    emitting state objects (rows sharing one object share one model state, as the K-layer
    lattice does, continuous_speech.py:38).  Mixtures of different sizes are padded with
    zero-weight unit-variance components."""
    row_state = np.empty(len(states), dtype=np.int32)
    uniq, index = [], {}
    for r, s in enumerate(states):
        if is_nes(s):
            row_state[r] = -1
            continue
        k = id(s)
        if k not in index:
            index[k] = len(uniq)
            uniq.append(s)
        row_state[r] = index[k]
    return row_state, uniq


def stack_gmms(gmms):
    arrs = [gmm_arrays(g) for g in gmms]
    M = max(a[0].shape[0] for a in arrs)
    D = arrs[0][0].shape[1]
    S = len(arrs)
    means = np.zeros((S, M, D))
    vars_ = np.ones((S, M, D))
    w = np.zeros((S, M))
    for s, (m, v, ww) in enumerate(arrs):
        if m.shape[1] != D:
            raise NameError("The dimensions of the input don't match")  # hmm_state.py:45
        k = m.shape[0]
        means[s, :k], vars_[s, :k], w[s, :k] = m, v, ww
    return means, vars_, w


try:                      # content keys of the handle caches: a 128-bit non-cryptographic hash is all that is needed
    import xxhash as _xx     # (blake2b took 2 ms for the 1 MB transition matrix of a K = 7 lattice: twice the decode itself)
except ImportError:          # pragma: no cover
    _xx = None


def _digest(*arrays):
    h = _xx.xxh3_128() if _xx is not None else hashlib.blake2b(digest_size=16)
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(("%s%s" % (a.dtype.str, a.shape)).encode())
        h.update(memoryview(a).cast("B") if a.size else b"")
    return h.digest()


class _LRU(OrderedDict):
    """Small content-addressed cache of device handles; keys start with id(ctx): a handle is never shared between
    contexts (it is bound to the context it was created with).  Thread-safe.  The cache never destroys a handle that
    somebody else may still hold: eviction forgets it, the handle's own finaliser frees it."""

    def __init__(self, cap):
        super().__init__()
        self.cap = cap
        self.lock = threading.RLock()

    def lookup(self, key, make):
        with self.lock:
            if key in self:
                self.move_to_end(key)
                return self[key]
            val = make()
            self[key] = val
            while len(self) > self.cap:
                # Evicting only drops the cache's reference.  Long-lived holders (a recogniser keeps the handle of its
                # stacked models) and threads that are inside a library call with the GIL released still use the
                # object; the device memory goes when the last of them lets go (PackedGMM / Lattices.__del__).
                self.popitem(last=False)
            return val

    def purge(self, ctx):
        """Close and drop every handle created with `ctx` (before the context itself is closed)."""
        with self.lock:
            for key in [k for k in self if k[0] == id(ctx)]:
                self.pop(key).close()


_gmm_cache = _LRU(8)
_normal_cache = _LRU(1024)   # single Gaussians behind MultivariateNormal.pdf (a few hundred bytes of HBM each)
_lat_cache = _LRU(16)


def device_gmm(ctx, gmms):
    """PackedGMM for a list of GMM objects (cached on parameter content)."""
    means, vars_, w = stack_gmms(gmms)
    key = (id(ctx), _digest(means, vars_, w))
    return _gmm_cache.lookup(key, lambda: _hip.PackedGMM(ctx, means, vars_, w))


def device_normal(ctx, mean, var):
    """PackedGMM of ONE diagonal Gaussian (cached on parameter content): what `MultivariateNormal.pdf` evaluates -- the
    reference calls it frame by frame (hmm_state.py:36-45), so the handle outlives the call."""
    mean = np.asarray(mean, dtype=np.float64)
    var = np.asarray(var, dtype=np.float64)
    key = (id(ctx), _digest(mean, var))
    return _normal_cache.lookup(key, lambda: _hip.PackedGMM(ctx, mean[None, None, :], var[None, None, :], np.ones((1, 1))))


def graph_from_dense(row_state, transitions, start_rows, end_rows):
    transitions = np.asarray(transitions, dtype=np.float64)
    to, frm = np.nonzero(~np.isinf(transitions))
    return dict(row_state=np.asarray(row_state, dtype=np.int32), arc_to=to.astype(np.int32),
                arc_from=frm.astype(np.int32), arc_cost=transitions[to, frm],
                start_rows=np.asarray(start_rows, dtype=np.int32), end_rows=np.asarray(end_rows, dtype=np.int32))


def device_lattices(ctx, graphs):
    """Lattices for a list of graph dicts (cached on content)."""
    parts = []
    for g in graphs:
        parts += [g["row_state"], g["arc_to"], g["arc_from"], g["arc_cost"], g["start_rows"], g["end_rows"]]
    key = (id(ctx), _digest(*[np.asarray(p) for p in parts]))
    return _lat_cache.lookup(key, lambda: _hip.Lattices(ctx, graphs))


def clear_caches():
    for c in (_gmm_cache, _normal_cache, _lat_cache):
        while c:
            _, v = c.popitem()
            v.close()
