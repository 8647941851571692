# -*- coding: utf-8 -*-
"""Every state's mixture refit advanced in LOCK-STEP on the GPU.

The reference trains one state after the other -- binary-split k-means, then mixture EM, on the frames the alignment
gave that state (hmm.py:97-124 inside `HMM.fit`, continuous_speech.py:114-142 inside `continuous_train`) -- thousands
of short passes over small arrays.  `LockstepFitter` gathers the frames of ALL states into one resident batch and
advances every state that has not converged yet with ONE launch per iteration (`gh_kmeans_assign_multi`,
`gh_em_accumulate_multi`); states drop out through a converged mask, exactly where the reference's per-state loops
would `break`.  The results are the reference's: the random partitions behind the k-means variances are drawn from
numpy's global generator in the order the sequential algorithm consumes it (they only depend on the frame counts), the
centroid means are numpy's, the M-step / convergence test is `GMM.em_update`.

DEVICE-RESIDENT form (round 3, `_hip.FitSession` = gh_fit_*): centroids, variances, mixture parameters, the allclose
test's "old" parameters, the active mask and the iteration counters live in HBM; a k-means / EM iteration is a handful
of kernel launches with no host synchronisation, the M-step + convergence test of `GMM.em_update` and the centroid
update + `np.array_equal` stop rule are kernels, and the host reads one counter every 8 iterations.  The partition
variances behind the k-means distance (kmeans.py:171-177) are two passes over each (state, cluster)'s frames on the
device (ddof 1) instead of `np.cov(...).diagonal()`, which forms the D x D matrix by a BLAS product to read its diagonal:
same value up to the library's summation order (`compat_cov=True` keeps np.cov and the host loop).  Taken whenever the
library is there, D >= 2, and the reducer is absent or the library's own communicator (`parallel.NativeReducer`: the
collectives then run on the device buffers); anything else keeps the host loop below.

Sharded over ranks (`reducer` = `parallel.StatsAllReducer` of a process group with more than one rank) every rank holds
part of each state's frames; cluster sums / counts and the EM statistics are all-reduced -- ONE collective per
lock-step iteration.  Summation order then differs from the single-process run, and the random
partitions are per rank: parity is statistical there, as the survey says.
"""
import threading

import numpy as np

from . import _hip
from .hmm_state import GMM

__all__ = ["LockstepFitter", "PartitionStream"]


def _variance_rows(x):
    """calc_variance(x.T) of kmeans.py:6-12: np.cov(...).diagonal() -- ddof = 1, per dimension."""
    return np.cov(x.T).diagonal()


def _fast_partition_ok():
    """np.random.randint(0, k, n) for a power of two k draws one 32-bit number per value and masks it (numpy's legacy
    RandomState, `_rand_int64` with rng < 2^32 and mask == rng: no rejection) -- the SAME draws as a full-range uint32
    randint.  Checked once against the installed numpy (state restored); False keeps the plain call."""
    st = np.random.get_state()
    try:
        np.random.seed(12345)
        a = [np.random.randint(0, k, 257) for k in (2, 4, 8, 16, 32)]
        tail_a = np.random.randint(0, 1 << 30, 4)
        np.random.seed(12345)
        b = [np.random.randint(0, 1 << 32, 257, dtype=np.uint32) & np.uint32(k - 1) for k in (2, 4, 8, 16, 32)]
        tail_b = np.random.randint(0, 1 << 30, 4)
        return all(np.array_equal(x, y) for x, y in zip(a, b)) and np.array_equal(tail_a, tail_b)
    except Exception:
        return False
    finally:
        np.random.set_state(st)


_FAST_PARTITION = None


def random_partition(k, n):
    """The random partition of kmeans.py:171 -- np.random.randint(0, k, n), the same values from the same draws of numpy's
    global generator -- as uint8 from 32-bit draws when k is a power of two (half the time of the int64 path: 31 -> 14 ms
    per outer iteration of continuous_train on 1.3 M frames x three binary splits)."""
    global _FAST_PARTITION
    if _FAST_PARTITION is None:
        _FAST_PARTITION = _fast_partition_ok()
    if _FAST_PARTITION and 2 <= k <= 256 and (k & (k - 1)) == 0:
        return (np.random.randint(0, 1 << 32, int(n), dtype=np.uint32) & np.uint32(k - 1)).astype(np.uint8)
    return np.random.randint(0, k, int(n))


def fast_partition_ok():
    global _FAST_PARTITION
    if _FAST_PARTITION is None:
        _FAST_PARTITION = _fast_partition_ok()
    return _FAST_PARTITION


class PartitionStream:
    """The random partitions of a whole refit (kmeans.py:171, one np.random.randint(0, k, n_s) per state and split) drawn
    AHEAD on a worker thread -- while the device aligns the utterances -- instead of between alignment and refit
    (4.5 ms of every continuous_train outer iteration on 1.4 M frames x two splits).

    For a power of two k every value is ONE 32-bit output of numpy's global MT19937, masked (`fast_partition_ok`
    checks that against the installed numpy), and the sequential algorithm consumes them state after state, split after
    split.  How many it consumes is only known after the alignment (frames on state entries are dropped), so the worker
    draws `n_max` outputs from a COPY of the global generator's state in chunks, remembering the state in front of
    every chunk; `take(lengths, n_splits)` cuts the words the refit really uses and leaves the global generator exactly
    where those draws leave it (the snapshot in front of the last chunk + the remainder drawn again).  Nothing else may
    draw from np.random between construction and take()."""

    CHUNK = 1 << 16
    _kept = None                # (buf, low) of the stream used last

    def __init__(self, n_max):
        self.n_max = int(n_max)
        self.state0 = np.random.get_state()
        # (the two buffers are kept between streams: continuous_train makes one per outer iteration, and 14 MB of fresh pages
        #  touched by the worker while the main thread drives the device cost the 2nd and 3rd iteration of a run 25-35 ms each)
        kept = PartitionStream._kept
        PartitionStream._kept = None
        if kept is not None and len(kept[0]) >= self.n_max:
            self.buf, self.low = kept
        else:
            self.buf = np.empty(self.n_max, dtype=np.uint32)
            self.low = np.empty(self.n_max, dtype=np.uint8)    # the low byte of every draw (k <= 256: all a partition needs)
        self.snaps = []
        self.error = None
        self.thread = threading.Thread(target=self._run, name="gmmhmm-partitions", daemon=True)
        self.thread.start()

    def _run(self):
        try:
            rs = np.random.RandomState()
            rs.set_state(self.state0)
            for a in range(0, self.n_max, self.CHUNK):
                self.snaps.append(rs.get_state())
                b = min(self.n_max, a + self.CHUNK)
                self.buf[a:b] = rs.randint(0, 1 << 32, b - a, dtype=np.uint32)
                self.low[a:b] = self.buf[a:b]                     # (same_kind cast, modulo 256: on the worker, not in take())
        except BaseException as e:        # (re-raised by take)
            self.error = e

    def take(self, lengths, n_splits):
        """-> parts[s][i]: uint8 partition of state s's n_s frames for split i (k = 2^(i+1))."""
        self.thread.join()
        if self.error is not None:
            np.random.set_state(self.state0)
            raise self.error
        lengths = [int(n) for n in lengths]
        assert n_splits <= 8, "PartitionStream: more than 256 groups"
        used = n_splits * sum(lengths)
        assert used <= self.n_max, "PartitionStream: the refit draws more values than were made"
        rs = np.random.RandomState()
        j = used // self.CHUNK
        if j < len(self.snaps):
            rs.set_state(self.snaps[j])
            rem = used - j * self.CHUNK
        else:                              # (exactly n_max, a whole number of chunks)
            rs.set_state(self.snaps[-1] if self.snaps else self.state0)
            rem = used - (len(self.snaps) - 1) * self.CHUNK if self.snaps else 0
        if rem:
            rs.randint(0, 1 << 32, rem, dtype=np.uint32)
        np.random.set_state(rs.get_state())
        parts, at = [], 0
        for n in lengths:
            row = []
            for i in range(n_splits):
                row.append(self.low[at:at + n] & np.uint8(2 ** (i + 1) - 1))
                at += n
            parts.append(row)
        PartitionStream._kept = (self.buf, self.low)          # (the partitions above are arrays of their own)
        return parts

    def cancel(self):
        """Nothing was used: the global generator stays where it was."""
        self.thread.join()
        np.random.set_state(self.state0)
        PartitionStream._kept = (self.buf, self.low)


class LockstepFitter:
    """segments: list of [N_s, D] arrays, one per state, in the order the reference would train them.
    source: (resident batch, row indices) -- the rows of an fp64 batch that, in this order, ARE the concatenated
    segments; the fitter's batch is then gathered on the device instead of uploaded again."""

    def __init__(self, segments, ctx=None, reducer=None, source=None, lengths=None, dim=None, kmax=None, compat_cov=False):
        """segments: the states' frames as host arrays -- or None with `lengths` [S] and `dim` when `source` names them
        as rows of a resident batch (nothing is copied to or from the host then; needs the device-resident form)."""
        self.ctx = ctx if ctx is not None else _hip.default_context()
        if segments is not None:
            self.segs = [np.ascontiguousarray(s, dtype=np.float64) for s in segments]
            lengths = [len(s) for s in self.segs]
            dim = self.segs[0].shape[1] if self.segs else 0
        else:
            assert source is not None and lengths is not None and dim is not None
            # (no state at all -- empty data, or every alignment failed: the empty fitter must still build, the callers'
            #  S == 0 early returns then do what the reference does: warn 'No MFCC data', keep the models)
            self.segs = None if len(lengths) else []
        self.S = len(lengths)
        self.D = int(dim) if self.S else 0
        self.seg_off = np.zeros(self.S + 1, dtype=np.int64)
        np.cumsum(lengths, out=self.seg_off[1:])
        self.reducer = reducer
        self.sharded = bool(reducer is not None and getattr(reducer, "enabled", False) and reducer.world_size > 1)
        if self.S and source is not None and hasattr(source[0], "gather") and source[0].np_dtype == np.float64:
            if isinstance(source[1], tuple) and source[1][0] == "runs":     # (start, length, dest, rows): contiguous runs of rows
                _, r_start, r_len, r_dest, r_n = source[1]
                assert r_n == int(self.seg_off[-1])
                self.batch = source[0].gather_runs(r_start, r_len, r_dest, r_n)
            else:
                assert len(source[1]) == int(self.seg_off[-1])
                self.batch = source[0].gather(source[1])
        else:
            assert self.segs is not None, "LockstepFitter: frames neither on the host nor in a resident fp64 batch"
            self.batch = _hip.Batch(self.ctx, feats=np.concatenate(self.segs) if self.S else np.zeros((0, 1)),
                                    offsets=[0, int(self.seg_off[-1])]) if self.S else None
        # the device-resident session (see the module docstring)
        self.fit = None
        self.comm = reducer.comm if (self.sharded and getattr(reducer, "native", False)) else None
        if self.S and not compat_cov and self.D >= 2 and (not self.sharded or self.comm is not None) and hasattr(_hip, "FitSession"):
            try:
                self.fit = _hip.FitSession(self.ctx, self.batch, self.seg_off, kmax if kmax else 8)
            except _hip.Unsupported:
                self.fit = None
        assert self.fit is not None or self.segs is not None, "LockstepFitter: this shape needs the frames on the host"
        self.collectives = 0
        # frames per state over all ranks (weights are counts / n)
        self.n_global = (self._reduce(np.diff(self.seg_off).astype(np.float64)) if self.sharded and self.S
                         else np.diff(self.seg_off))

    def _reduce(self, a):
        self.collectives += 1
        return self.reducer(np.ascontiguousarray(a, dtype=np.float64))

    def close(self):
        if self.fit is not None:
            self.fit.close()
            self.fit = None
        if self.batch is not None:
            self.batch.close()
            self.batch = None

    def segment_means(self):
        """np.mean(segment, axis=0) of every state (the start centroids of continuous_train, continuous_speech.py:116)."""
        if self.fit is not None:
            sums, counts = self.fit.segment_means()
            with np.errstate(all="ignore"):
                return sums / counts[:, None]
        return np.array([np.mean(seg, axis=0) for seg in self.segs])

    def _ensure_session(self, k):
        """The session is sized for kmax components at creation; a larger k builds a new one."""
        if self.fit is not None and k > self.fit.kmax:
            self.fit.close()
            try:
                self.fit = _hip.FitSession(self.ctx, self.batch, self.seg_off, k)
            except _hip.Unsupported:
                self.fit = None
                assert self.segs is not None, "LockstepFitter: this shape needs the frames on the host"

    # ------------------------------------------------------------------ k-means (kmeans.py:167-193), all states
    def kmeans(self, k, centroids, max_iteration=1000, partitions=None, want_clusters=True):
        """centroids [S, k, D] -> (clusters: list of int64 [N_s], centroids [S, k, D], cov [S, k, D]).
        partitions: the random partition of every state's frames (kmeans.py:171), drawn by the caller when the order
        of draws matters; default: drawn here, state after state."""
        S, D, off = self.S, self.D, self.seg_off
        if S == 0:        # no state was visited (continuous_train: "No MFCC data" for every state)
            return [], np.zeros((0, k, D)), np.zeros((0, k, D))
        self._ensure_session(k)
        if self.fit is not None:
            lens = np.diff(off)
            if partitions is None:
                partitions = [random_partition(k, n) for n in lens]
            part = np.concatenate([np.asarray(p, dtype=np.uint8) for p in partitions]) if S else np.zeros(0, np.uint8)
            cen, cov, cnt, its = self.fit.kmeans(k, centroids, part, max_iteration=max_iteration, comm=self.comm)
            if self.comm is not None:
                self.collectives += 3 + int(its.max(initial=0))   # partition sums (2), per iteration one, cluster sizes
            self.last_counts = cnt
            if not want_clusters:
                return None, cen, cov
            cl = self.fit.clusters()
            return [cl[off[s]:off[s + 1]].astype(np.int64) for s in range(S)], cen, cov
        cov = np.empty((S, k, D))
        part_stats = np.zeros((S, 2 * D + 1))
        for s, x in enumerate(self.segs):
            part = random_partition(k, x.shape[0]) if partitions is None else partitions[s]
            if self.sharded:      # variance of cluster 0 from sums over all ranks (ddof = 1), the only row that is used
                x0 = x[part == 0]
                part_stats[s, 0] = len(x0)
                part_stats[s, 1:1 + D] = x0.sum(axis=0)
                part_stats[s, 1 + D:] = (x0 * x0).sum(axis=0)
            else:
                with np.errstate(all="ignore"):
                    cov[s] = np.array([_variance_rows(x[part == c]) for c in range(k)])
        if self.sharded:
            ps = self._reduce(part_stats)
            n0 = ps[:, [0]]
            with np.errstate(all="ignore"):
                v0 = (ps[:, 1 + D:] - ps[:, 1:1 + D] ** 2 / n0) / (n0 - 1.0)
            cov[:] = v0[:, None, :]
        centroids = np.array(centroids, dtype=np.float64)
        # the assignments stay on the device between iterations when nothing on the host needs them (D >= 2)
        resident = D >= 2 and hasattr(self.batch, "resident_clusters")
        if resident:
            self.batch.resident_clusters(reset=True, fetch=False)
        clusters = _hip.RESIDENT if resident else np.full(int(off[-1]), -1, dtype=np.int32)
        active = np.ones(S, dtype=np.uint8)
        # cluster sums come back accumulated in frame order -- the order np.mean(x[cl == c], axis=0) adds the rows of a
        # C-contiguous array -- so sums / count IS cluster_centroids' mean, bit for bit (kmeans.py:158-164); with one
        # feature dimension numpy sums pairwise instead, and the means are taken on the host
        dev_means = D >= 2
        for _ in range(max(max_iteration, 1)):
            got, changed, sums = self.batch.kmeans_assign_multi(off, centroids, var=cov[:, 0, :], clusters=clusters,
                                                               active=active, want_sums=self.sharded or dev_means)
            if not resident:
                clusters = got
            if self.sharded:
                red = self._reduce(np.concatenate([sums.reshape(S, -1), changed.reshape(S, 1).astype(np.float64)], axis=1))
                sums, changed = red[:, :-1].reshape(S, k, D + 1), red[:, -1]
                with np.errstate(all="ignore"):
                    new = sums[:, :, :D] / sums[:, :, [D]]
                for s in np.nonzero(active)[0]:
                    if changed[s] == 0:
                        active[s] = 0                 # assignments stable on every rank: centroids are stable
                    centroids[s] = new[s]
            else:
                if dev_means:
                    with np.errstate(all="ignore"):
                        means = sums[:, :, :D] / sums[:, :, [D]]       # an empty cluster: 0 / 0 = nan, like np.mean of nothing
                for s in np.nonzero(active)[0]:
                    if dev_means:
                        new = means[s]
                    else:
                        cl = clusters[off[s]:off[s + 1]]
                        new = np.empty((k, D))
                        with np.errstate(all="ignore"):
                            for c in range(k):
                                np.mean(self.segs[s][cl == c, :], axis=0, out=new[c])   # cluster_centroids, kmeans.py:158-164
                    if np.array_equal(new, centroids[s]):
                        active[s] = 0                 # kmeans.py:190-191
                    else:
                        centroids[s] = new
            if not active.any():
                break
        if resident:
            clusters = self.batch.resident_clusters()
        return [clusters[off[s]:off[s + 1]].astype(np.int64) for s in range(S)], centroids, cov

    # ------------------------------------------------------------------ mixture EM (hmm_state.py:122-159), all states
    def em(self, states, k, max_iteration=10000):
        """Run GMM.em(data_s, k) for every state object in `states` (one per segment), in lock-step."""
        S, off = self.S, self.seg_off
        if S == 0:
            return
        self._ensure_session(k)
        if self.fit is not None:
            D = self.D
            mean = np.array([[np.asarray(d.mean, dtype=np.float64) for d in g.dists[:k]] for g in states]).reshape(S, k, D)
            var = np.array([[np.asarray(d.cov, dtype=np.float64) for d in g.dists[:k]] for g in states]).reshape(S, k, D)
            w = np.array([np.asarray(g.w[:k], dtype=np.float64) for g in states]).reshape(S, k)
            mu_old = np.ascontiguousarray(np.array([g.mu_old[:k, :] for g in states], dtype=np.float64))
            sg_old = np.ascontiguousarray(np.array([g.sigma_old[:k, :] for g in states], dtype=np.float64))
            w_old = np.ascontiguousarray(np.array([g.w_old[:k] for g in states], dtype=np.float64))
            conv = self.fit.em(k, mean, var, w, mu_old, sg_old, w_old, self.n_global, max_iteration=max_iteration, comm=self.comm)
            if self.comm is not None:
                self.collectives += int(np.where(conv >= 0, conv + 1, max_iteration).max(initial=0))
            for s, g in enumerate(states):
                g.update_models(mean[s], var[s], w[s])
                g.mu_old[:k, :], g.sigma_old[:k, :], g.w_old[:k] = mu_old[s], sg_old[s], w_old[s]
                if conv[s] >= 0:
                    print("EM converged at iteration:", int(conv[s]))
            return
        active = np.ones(S, dtype=np.uint8)
        for it in range(max_iteration):
            means = np.array([[np.asarray(d.mean, dtype=np.float64) for d in g.dists[:k]] for g in states])
            vars_ = np.array([[np.asarray(d.cov, dtype=np.float64) for d in g.dists[:k]] for g in states])
            w = np.array([np.asarray(g.w[:k], dtype=np.float64) for g in states])
            stats, _ = self.batch.em_accumulate_multi(off, means, vars_, w, active=active)
            if self.sharded:
                stats = self._reduce(stats)
            for s in np.nonzero(active)[0]:
                if states[s].em_update(stats[s], self.n_global[s], k, it):
                    active[s] = 0
            if not active.any():
                break

    # ------------------------------------------------------------------ the whole refit of hmm.py:97-124
    def split_and_fit(self, states, start_centroids, n_gaussians, weight_divisor, use_em=True, parts=None):
        """Binary-split k-means + EM of every state: for i in range(int(ln n_gaussians)): centroids x 0.9 / x 1.1,
        k-means under the mahalanobis distance, weights = cluster counts / weight_divisor[s] (looked up by cluster id,
        hmm.py:116-118), update_models, EM on the first 2^(i+1) components.
        start_centroids [S, D]; weight_divisor [S] (the state's frame count in HMM.fit, `n_segments` in
        continuous_train, continuous_speech.py:127,137)."""
        n_splits = int(np.log(n_gaussians))
        assert n_splits > 0
        S = self.S
        if S == 0:
            return
        centroids = np.asarray(start_centroids, dtype=np.float64).reshape(S, 1, self.D)
        div = np.asarray(weight_divisor, dtype=np.float64)
        weights = np.repeat((1.0 / div)[:, None], n_gaussians, axis=1)
        # numpy's global generator is consumed the way the sequential algorithm consumes it: state after state, and
        # inside a state split after split (one draw of N_s cluster ids per kmeans call, kmeans.py:171; nothing else
        # in the refit draws) -- the draws only depend on the frame counts, so they can all be made up front
        # (`parts`: the same draws made ahead by a PartitionStream)
        if parts is None:
            parts = [[random_partition(2 ** (i + 1), n_s) for i in range(n_splits)] for n_s in np.diff(self.seg_off)]
        for i in range(n_splits):
            k = 2 ** (i + 1)
            centroids = np.concatenate([centroids * 0.9, centroids * 1.1], axis=1)
            clusters, centroids, cov = self.kmeans(k, centroids, partitions=[p[i] for p in parts], want_clusters=False)
            if self.fit is not None:
                # device-resident session: the mixtures travel as arrays; the state objects are written ONCE, after the last
                # split (every split overwrites the first 2^(i+1) components the previous one wrote, nothing reads them in
                # between -- 400 MultivariateNormal updates per outer iteration of continuous_train were 3.6 ms of host time)
                counts = self.last_counts                      # cluster sizes (over all ranks when sharded), from the device
                for s in range(S):
                    ids = np.nonzero(counts[s])[0]             # np.unique(clusters): the ids that occur, ascending ...
                    cnt = counts[s] if self.sharded else counts[s][ids]     # ... and their sizes
                    for c in ids:
                        # looked up by cluster ID (hmm.py:116-118) -- the reference's indexing; an IndexError there too
                        weights[s, c] = cnt[c] / div[s]
                if np.any(cov == 0):
                    raise np.linalg.LinAlgError("Singular matrix")   # MultivariateNormal.cov's np.linalg.inv (hmm_state.py:30)
                mean_a, var_a, w_a = np.ascontiguousarray(centroids), np.ascontiguousarray(cov), np.ascontiguousarray(weights[:, :k])
                if use_em:
                    mean_a, var_a, w_a = mean_a.copy(), var_a.copy(), w_a.copy()
                    mu_old = np.ascontiguousarray(np.array([g.mu_old[:k, :] for g in states], dtype=np.float64))
                    sg_old = np.ascontiguousarray(np.array([g.sigma_old[:k, :] for g in states], dtype=np.float64))
                    w_old = np.ascontiguousarray(np.array([g.w_old[:k] for g in states], dtype=np.float64))
                    conv = self.fit.em(k, mean_a, var_a, w_a, mu_old, sg_old, w_old, self.n_global, max_iteration=10000, comm=self.comm)
                    if self.comm is not None:
                        self.collectives += int(np.where(conv >= 0, conv + 1, 10000).max(initial=0))
                    for s, g in enumerate(states):
                        g.mu_old[:k, :], g.sigma_old[:k, :], g.w_old[:k] = mu_old[s], sg_old[s], w_old[s]
                        if conv[s] >= 0:
                            print("EM converged at iteration:", int(conv[s]))
                if i == n_splits - 1 or not use_em:
                    for s, g in enumerate(states):
                        g.update_models(mean_a[s], var_a[s], w_a[s])
                continue
            if self.fit is not None:
                counts = self.last_counts                      # cluster sizes (over all ranks when sharded), from the device
            elif self.sharded:
                counts = np.zeros((S, k))
                for s in range(S):
                    counts[s] = np.bincount(clusters[s], minlength=k)[:k]
                counts = self._reduce(counts)
            for s in range(S):
                if self.sharded:
                    for c in np.nonzero(counts[s])[0]:
                        weights[s, c] = counts[s, c] / div[s]
                elif self.fit is not None:
                    ids = np.nonzero(counts[s])[0]             # np.unique(clusters): the ids that occur, ascending ...
                    cnt = counts[s][ids]                       # ... and their sizes
                    for c in ids:
                        weights[s, c] = cnt[c] / div[s]        # looked up by cluster ID (hmm.py:116-118): the reference's indexing
                else:
                    ids, cnt = np.unique(clusters[s], return_counts=True)
                    for c in ids:
                        weights[s, c] = cnt[c] / div[s]       # counts looked up by cluster ID, as in hmm.py:116-118
                states[s].update_models(centroids[s], cov[s], weights[s, :k])
            if use_em:
                self.em(states, k)
