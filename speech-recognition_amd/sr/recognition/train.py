# -*- coding: utf-8 -*-
"""Batched EM (Baum-Welch) training of the mixture parameters on one or more GPUs.

The reference trains by Viterbi alignment + per-state EM (`continuous_train`,
continuous_speech.py:56-179 -- mirrored in `continuous_speech.py` here).  The north-star
additionally asks for the soft version: E-step = forward-backward over every utterance's
forced-alignment lattice, M-step from sufficient statistics that are all-reduced over the
GPUs of a node.  This module is that loop; it is NEW functionality, built
from the same graphs (`packed_lattice`) and the same centred-statistics layout as `GMM.em`.

One EM iteration, per rank (utterances are sharded, models replicated):
    gh_loglik            frame x state likelihoods of the rank's frames        (HIP, MFMA)
    gh_forward_backward  log P(u), per-frame state occupancies                 (HIP)
    gh_bw_accumulate     [S, M, 1+2D] centred statistics of the rank           (HIP)
    all-reduce           ONE packed fp64 buffer (stats + counts + log P)       (RCCL / gloo)
    m_step               new means / variances / weights                       (host, tiny)
"""
import numpy as np

from . import _hip
from .continuous_speech import packed_lattice
from .parallel import m_step, StatsAllReducer

__all__ = ["BaumWelchTrainer"]


class BaumWelchTrainer:
    """Soft EM over word-level transcriptions.

    means / vars_ / weights: [W, n, M, D] / [W, n, M, D] / [W, n, M] word-state mixtures;
    transitions: list of W [n, n] cost matrices (kept fixed);
    data: list of [T_u, D] utterances of THIS rank; label_seqs: word indices per utterance.
    var_floor: lower bound of the re-estimated variances; None (default) = 1e-6 x the mean variance of the
    initial model -- the centred single-pass variance can cancel to 0 (or slightly below) for a component
    that holds on to a single frame, and a non-positive variance is a LinAlgError in the next E-step."""

    def __init__(self, means, vars_, weights, transitions, data, label_seqs, device=None, reducer=None,
                 var_floor=None, occ_floor=0.0, min_occupancy=1e-8):
        self.ctx = _hip.default_context(device)
        self.W, self.n, self.M, self.D = means.shape
        self.S = self.W * self.n
        self.means = np.array(means, dtype=np.float64).reshape(self.S, self.M, self.D)
        self.vars = np.array(vars_, dtype=np.float64).reshape(self.S, self.M, self.D)
        self.weights = np.array(weights, dtype=np.float64).reshape(self.S, self.M)
        self.var_floor = 1e-6 * float(np.mean(self.vars)) if var_floor is None else float(var_floor)
        self.occ_floor, self.min_occupancy = occ_floor, float(min_occupancy)
        self.reducer = reducer if reducer is not None else StatsAllReducer(gpu_index=self.ctx.device)
        self.batch = _hip.Batch(self.ctx, data)
        keys, graphs = {}, []
        self.utt_graph = np.empty(len(label_seqs), dtype=np.int32)
        for u, labels in enumerate(label_seqs):
            key = tuple(int(l) for l in labels)
            if key not in keys:
                keys[key] = len(graphs)
                graphs.append(packed_lattice(transitions, self.n, [[l] for l in key])[0])
            self.utt_graph[u] = keys[key]
        self.lat = _hip.Lattices(self.ctx, graphs) if graphs else None
        # an utterance's alignment only involves the states of its own words: likelihoods for that range only
        lo = np.array([min(int(l) for l in labels) * self.n if len(labels) else 0 for labels in label_seqs], dtype=np.int32)
        hi = np.array([(max(int(l) for l in labels) + 1) * self.n if len(labels) else self.S for labels in label_seqs],
                      dtype=np.int32)
        self.state_ranges = (lo, hi)
        self.history = []
        self.n_stats = self.S * self.M * (1 + 2 * self.D)
        self.last_timing = {}

    # layout of the ONE buffer that crosses ranks: [statistics S*M*(1+2D) | total log-likelihood | utterances]
    def _packed_len(self):
        return self.n_stats + 2

    def e_step(self, stats_dev=None):
        """Returns (stats [S,M,1+2D] or None when they were left in `stats_dev`, total log-likelihood) of this rank."""
        gmm = _hip.PackedGMM(self.ctx, self.means, self.vars, self.weights)
        try:
            if self.batch.U == 0:
                return (None if stats_dev else np.zeros((self.S, self.M, 1 + 2 * self.D))), 0.0
            self.batch.loglik(gmm, fetch=False, state_ranges=self.state_ranges)
            r = self.lat.forward_backward(self.batch, utt_lattice=self.utt_graph, want_occ=True, fetch_occ=False)
            stats = self.batch.bw_accumulate(gmm, occ_floor=self.occ_floor, stats_dev=stats_dev, fetch=stats_dev is None)
            logp = r["logp"]
            return stats, float(np.sum(logp[np.isfinite(logp)]))
        finally:
            gmm.close()

    def iteration(self):
        """One EM iteration over all ranks; returns the total log-likelihood BEFORE the update."""
        import time
        t0 = time.perf_counter()
        red = self.reducer
        if red.on_gpu:
            # statistics go from the kernel's slabs straight into the tensor RCCL reduces: no host bounce
            buf, ptr = red.device_buffer(self._packed_len())
            if self.batch.U == 0:
                buf.zero_()
            _, ll = self.e_step(stats_dev=ptr)
            buf[self.n_stats:] = red.torch.tensor([ll, float(self.batch.U)], dtype=red.torch.float64)
            t1 = time.perf_counter()
            packed = red.reduce_device()                     # the ONE collective of the iteration
        else:
            stats, ll = self.e_step()
            packed = np.concatenate([stats.reshape(-1), [ll, float(self.batch.U)]])
            t1 = time.perf_counter()
            packed = red(packed)                            # the ONE collective of the iteration (gloo / single rank)
        t2 = time.perf_counter()
        stats = packed[:self.n_stats].reshape(self.S, self.M, 1 + 2 * self.D)
        ll = float(packed[self.n_stats])
        counts = stats[:, :, 0].sum(axis=1)                  # responsibilities of a state add up to its occupancy
        seen = counts > 0
        mu, sigma, w = m_step(stats[seen], counts[seen], self.means[seen])
        sigma = np.maximum(sigma, self.var_floor)
        ok = stats[seen][:, :, 0] > self.min_occupancy       # components that received mass
        self.means[seen] = np.where(ok[:, :, None], mu, self.means[seen])
        self.vars[seen] = np.where(ok[:, :, None], sigma, self.vars[seen])
        self.weights[seen] = np.where(ok, w, self.weights[seen])
        self.history.append(ll)
        self.last_timing = dict(e_step_s=t1 - t0, allreduce_s=t2 - t1, m_step_s=time.perf_counter() - t2)
        return ll

    def fit(self, n_iterations=5):
        for _ in range(n_iterations):
            self.iteration()
        return self.history

    def close(self):
        self.batch.close()
        if self.lat is not None:
            self.lat.close()
