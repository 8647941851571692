# -*- coding: utf-8 -*-
"""Batched EM (Baum-Welch) training of the mixture parameters on one or more GPUs.

The reference trains by Viterbi alignment + per-state EM (`continuous_train`,
continuous_speech.py:56-179 -- mirrored in `continuous_speech.py` here).  The north-star
additionally asks for the soft version: E-step = forward-backward over every utterance's
forced-alignment lattice, M-step from sufficient statistics that are all-reduced over the
GPUs of a node.  This module is that loop; it is NEW functionality, built
from the same graphs (`packed_lattice`) and the same centred-statistics layout as `GMM.em`.

One EM iteration, per rank (utterances are sharded, models replicated):
    gh_loglik_subset     likelihoods of every utterance's own states                            (HIP, MFMA)
    gh_forward_backward  log P(u), per-frame state occupancies, expected self transitions       (HIP)
    gh_bw_accumulate     [S, M, 1+2D] centred statistics of the rank                            (HIP)
    all-reduce           ONE packed fp64 buffer (statistics + self transitions + log P + count) (RCCL / gloo)
    m_step               new means / variances / weights, new transition costs                  (host, tiny)
With one-word transcripts (isolated-word training, BASELINE configs[2]) the whole iteration -- including the all-reduce
(the library's own RCCL communicator, `parallel.NativeReducer`), the M-step, the transition update, the model re-pack
and the stop rule -- is ONE call that enqueues kernels on the context's stream (`_hip.EMSession`, gh_em_iteration):
no host arithmetic, no synchronisation between the steps; the call-by-call form above remains for multi-word
transcripts and for process groups that are not RCCL (gloo in the CPU tests).
The loop around it is the reference's training loop in its soft form (continuous_speech.py:144-179): transition
costs re-estimated every iteration (-log p_jump / -log(1 - p_jump), :146-164, with expected instead of counted
segments), one pickle per word model and iteration (:167-170), stop when every mixture is `allclose` to the previous
iteration's (:172-179; transitions are not compared there either).
"""
import numpy as np

from . import _hip
from .parallel import m_step, StatsAllReducer

__all__ = ["BaumWelchTrainer"]


class BaumWelchTrainer:
    """Soft EM over word-level transcriptions.

    means / vars_ / weights: [W, n, M, D] / [W, n, M, D] / [W, n, M] word-state mixtures;
    transitions: list of W [n, n] cost matrices (re-estimated every iteration unless update_transitions=False);
    data: list of [T_u, D] utterances of THIS rank; label_seqs: word indices per utterance.
    var_floor: lower bound of the re-estimated variances; None (default) = 1e-6 x the mean variance of the
    initial model -- the centred single-pass variance can cancel to 0 (or slightly below) for a component
    that holds on to a single frame, and a non-positive variance is a LinAlgError in the next E-step.
    occ_floor: state posteriors gamma_t(s) <= occ_floor are dropped from the statistics.  Default 1e-30: such a term cannot
    move a sum the M-step uses (a component needs min_occupancy = 1e-8 of summed responsibility to be re-estimated) in
    its 16 digits, and the statistics kernel skips 16-frame blocks in which a state pair has no occupancy at all --
    posteriors are sharp, so ~45 % of the (block, pair) work is left (0.0 keeps every non-zero posterior: ~75 %).
    output_path: directory that receives `<word index>.pkl` (reference-compatible HMM pickles) after every iteration,
    written by rank 0 only."""

    def __init__(self, means, vars_, weights, transitions, data, label_seqs, device=None, reducer=None,
                 var_floor=None, occ_floor=1e-30, min_occupancy=1e-8, update_transitions=True, output_path=None,
                 device_resident=True):
        self.ctx = _hip.default_context(device)
        self.W, self.n, self.M, self.D = means.shape
        self.S = self.W * self.n
        self._means = np.array(means, dtype=np.float64).reshape(self.S, self.M, self.D)
        self._vars = np.array(vars_, dtype=np.float64).reshape(self.S, self.M, self.D)
        self._weights = np.array(weights, dtype=np.float64).reshape(self.S, self.M)
        self._transitions = [np.array(t, dtype=np.float64) for t in transitions]
        self._stale = False           # the device-resident session holds a newer model than the host copies
        self.session = None
        self._gmm = None
        self.update_transitions = bool(update_transitions)
        self.output_path = output_path
        self.var_floor = 1e-6 * float(np.mean(self.vars)) if var_floor is None else float(var_floor)
        self.occ_floor, self.min_occupancy = occ_floor, float(min_occupancy)
        self.reducer = reducer if reducer is not None else StatsAllReducer(gpu_index=self.ctx.device)
        # (a rank may hold no utterances: its batch still has the model's feature dimension)
        self.batch = _hip.Batch(self.ctx, data) if len(data) else _hip.Batch(self.ctx, feats=np.zeros((0, self.D)), offsets=[0])
        keys = {}
        self.graph_labels = []
        self.utt_graph = np.empty(len(label_seqs), dtype=np.int32)
        for u, labels in enumerate(label_seqs):
            key = tuple(int(l) for l in labels)
            if key not in keys:
                keys[key] = len(self.graph_labels)
                self.graph_labels.append(key)
            self.utt_graph[u] = keys[key]
        self.lat = None
        self.history = []
        self.converged = False
        self.n_stats = self.S * self.M * (1 + 2 * self.D)
        self.last_timing = {}
        # one-word transcripts + a reducer that is the library's own communicator (or none): the device-resident session
        red = self.reducer
        native = getattr(red, "native", False)
        alone = not getattr(red, "enabled", False) or red.world_size == 1
        single = all(len(l) == 1 for l in label_seqs)
        possible = bool(device_resident and (native or alone) and self.batch.np_dtype == np.float64
                        and all(len(l) >= 1 for l in label_seqs))
        if not alone and native:      # (another kind of reducer rules the session out on every rank alike)
            # the choice is COLLECTIVE: a rank with a multi-word transcript (or another dtype) takes the
            # call-by-call path, and then every rank must -- M-step and stop test on the device here and in numpy there
            # would let the replicated models drift apart bit by bit, and with them the number of collectives per fit()
            n_no = red(np.array([0.0 if possible else 1.0]))
            possible = possible and float(np.asarray(n_no).ravel()[0]) == 0.0
        if possible:
            # one-word transcripts: the chain-form session; word strings: the sequence-form one (either may decline a
            # shape -- the call-by-call path stays; ranks of the two kinds share the packed buffer and the M-step kernel)
            try:
                if single:
                    self.session = _hip.EMSession(self.ctx, self.batch, self._means, self._vars, self._weights,
                                                  np.asarray(self._transitions), [int(l[0]) for l in label_seqs], self.var_floor,
                                                  occ_floor=self.occ_floor, min_occupancy=self.min_occupancy,
                                                  update_transitions=self.update_transitions)
                else:
                    self.session = _hip.EMSession(self.ctx, self.batch, self._means, self._vars, self._weights,
                                                  np.asarray(self._transitions), self.utt_graph, self.var_floor,
                                                  occ_floor=self.occ_floor, min_occupancy=self.min_occupancy,
                                                  update_transitions=self.update_transitions, transcripts=self.graph_labels)
            except _hip.Unsupported:
                self.session = None
        if not alone and native:
            # ... and so is the outcome: a rank whose shapes the session declined takes every rank to the call-by-call path
            n_no = red(np.array([0.0 if (self.session is not None or not possible) else 1.0]))
            if possible and float(np.asarray(n_no).ravel()[0]) != 0.0 and self.session is not None:
                self.session.close()
                self.session = None
        self._label_seqs = label_seqs
        self.state_sets = None
        if self.session is None:
            self._ensure_call_path()

    def _ensure_call_path(self):
        """Graphs and state sets of the call-by-call E-step (built on first use when the session runs the iterations)."""
        if self.state_sets is None:
            self._build_lattices()
            # an utterance's alignment only involves the states of its own words: likelihoods for those states only
            from .continuous_speech import transcript_state_sets
            self.state_sets = transcript_state_sets(self._label_seqs, self.n, self.W)

    # the model on the host; with a device-resident session the copies are refreshed when somebody looks
    def _pull(self):
        if self._stale and self.session is not None:
            m, v, w, t = self.session.model()
            self._means, self._vars, self._weights = m, v, w
            self._transitions = [t[i] for i in range(self.W)]
            self._stale = False

    means = property(lambda self: (self._pull(), self._means)[1], lambda self, v: setattr(self, "_means", v))
    vars = property(lambda self: (self._pull(), self._vars)[1], lambda self, v: setattr(self, "_vars", v))
    weights = property(lambda self: (self._pull(), self._weights)[1], lambda self, v: setattr(self, "_weights", v))
    transitions = property(lambda self: (self._pull(), self._transitions)[1], lambda self, v: setattr(self, "_transitions", v))

    def _build_lattices(self):
        """One forced-alignment graph per distinct label sequence, from the current transition costs."""
        if self.lat is not None:
            self.lat.close()
        if not self.graph_labels:
            self.lat = None
            return
        if getattr(self, "_flat_labels", None) is None:      # the label strings never change: flattened once
            self._flat_labels = _hip.Lattices.flatten_transcripts(self.graph_labels)
        self.lat = _hip.Lattices.from_transcripts(self.ctx, self.transitions, self.n, self.graph_labels, flat=self._flat_labels)

    # layout of the ONE buffer that crosses ranks:
    #   [statistics S*M*(1+2D) | expected self transitions S | total log-likelihood | utterances]
    def _packed_len(self):
        return self.n_stats + self.S + 2

    def e_step(self, stats_dev=None):
        """Returns (stats [S,M,1+2D] or None when they were left in `stats_dev`, expected self transitions [S],
        total log-likelihood) of this rank."""
        self._ensure_call_path()
        if self.session is not None and self.lat is not None:
            self._build_lattices()      # (the session may have moved the transition costs since the graphs were built)
        if self._gmm is None:           # packed once; later E-steps re-pack in place on the device (gh_gmm_update)
            self._gmm = _hip.PackedGMM(self.ctx, self.means, self.vars, self.weights)
        else:
            self._gmm.update(self.means, self.vars, self.weights)
        gmm = self._gmm
        if self.batch.U == 0:
            return (None if stats_dev else np.zeros((self.S, self.M, 1 + 2 * self.D))), np.zeros(self.S), 0.0
        self.batch.loglik(gmm, fetch=False, state_sets=self.state_sets)
        r = self.lat.forward_backward(self.batch, utt_lattice=self.utt_graph, want_occ=True, fetch_occ=False,
                                      want_self_xi=True)
        stats = self.batch.bw_accumulate(gmm, occ_floor=self.occ_floor, stats_dev=stats_dev, fetch=stats_dev is None)
        logp = r["logp"]
        return stats, r["self_xi"], float(np.sum(logp[np.isfinite(logp)]))

    def _comm(self):
        red = self.reducer
        return red.comm if getattr(red, "native", False) else None

    def iteration(self, sync=True):
        """One EM iteration over all ranks; returns the total log-likelihood BEFORE the update.
        sync=False (device-resident session only): the iteration is enqueued and None is returned; `drain()` collects
        the log-likelihoods of all iterations enqueued so far."""
        import time
        t0 = time.perf_counter()
        if self.session is not None:
            out = self.session.iteration(comm=self._comm(), sync=sync)
            self._stale = True
            if not sync:
                # (the session keeps the last 4096 history rows: unread ones are collected before the ring wraps)
                if self.session.iterations_done - len(self.history) >= 4000:
                    self._drain_to(self.session.iterations_done)
                return None
            self._drain_to(self.session.iterations_done - 1)
            ll, _, self.converged = out
            self.history.append(ll)
            if self.output_path is not None and self._is_writer():
                self.save(self.output_path)
            dt = time.perf_counter() - t0
            self.last_timing = dict(e_step_s=dt, allreduce_s=0.0, m_step_s=0.0)
            return ll
        assert sync, "sync=False needs the device-resident session"
        red = self.reducer
        if red.on_gpu:
            # statistics go from the kernel's slabs straight into the tensor RCCL reduces: no host bounce
            buf, ptr = red.device_buffer(self._packed_len())
            if self.batch.U == 0:
                buf.zero_()
            _, xi, ll = self.e_step(stats_dev=ptr)
            buf[self.n_stats:] = red.torch.from_numpy(np.concatenate([xi, [ll, float(self.batch.U)]]))
            t1 = time.perf_counter()
            packed = red.reduce_device()                     # the ONE collective of the iteration
        else:
            stats, xi, ll = self.e_step()
            packed = np.concatenate([stats.reshape(-1), xi, [ll, float(self.batch.U)]])
            t1 = time.perf_counter()
            packed = red(packed)                            # the ONE collective of the iteration (gloo / single rank)
        t2 = time.perf_counter()
        stats = packed[:self.n_stats].reshape(self.S, self.M, 1 + 2 * self.D)
        xi = packed[self.n_stats:self.n_stats + self.S]
        ll = float(packed[self.n_stats + self.S])
        old = (self.means.copy(), self.vars.copy(), self.weights.copy())
        counts = stats[:, :, 0].sum(axis=1)                  # responsibilities of a state add up to its occupancy
        seen = counts > 0
        mu, sigma, w = m_step(stats[seen], counts[seen], self.means[seen])
        sigma = np.maximum(sigma, self.var_floor)
        ok = stats[seen][:, :, 0] > self.min_occupancy       # components that received mass
        self.means[seen] = np.where(ok[:, :, None], mu, self.means[seen])
        self.vars[seen] = np.where(ok[:, :, None], sigma, self.vars[seen])
        self.weights[seen] = np.where(ok, w, self.weights[seen])
        if self.update_transitions:
            self._update_transitions(counts, xi)
        self.history.append(ll)
        # the reference's stop rule (continuous_speech.py:172-179 -> GMM.__eq__): every mixture allclose to the last one's
        self.converged = bool(np.allclose(self.weights, old[2]) and np.allclose(self.means, old[0]) and
                              np.allclose(self.vars, old[1]))
        if self.output_path is not None and self._is_writer():
            self.save(self.output_path)
        self.last_timing = dict(e_step_s=t1 - t0, allreduce_s=t2 - t1, m_step_s=time.perf_counter() - t2)
        return ll

    def _update_transitions(self, counts, self_xi):
        """continuous_speech.py:146-164 with expected counts: p_jump = segments / frames of the state, where the
        expected number of segments (visits) is frames - self transitions; cost(s -> s+1) = -log p_jump (not for
        the last state), cost(s -> s) = -log(1 - p_jump).  States without frames keep their costs (:149-153)."""
        changed = False
        for wi in range(self.W):
            t = self.transitions[wi]
            for si in range(self.n):
                s = wi * self.n + si
                if not counts[s] > 0:
                    continue
                p_stay = min(max(self_xi[s] / counts[s], 0.0), 1.0)
                with np.errstate(divide="ignore"):
                    if si < self.n - 1:
                        t[si + 1, si] = -np.log(1.0 - p_stay)
                    t[si, si] = -np.log(p_stay)
                changed = True
        if changed:
            self._build_lattices()

    def _drain_to(self, upto):
        """history rows of enqueued-but-unread iterations [len(history), upto) (device-resident session)."""
        have = len(self.history)
        if upto > have:
            rows = self.session.history(have, upto - have)
            self.history.extend(float(x) for x in rows[:, 0])
            self.converged = bool(rows[-1, 2])
            if np.any(rows[:, 3].astype(np.int64) & 16):
                raise np.linalg.LinAlgError("Singular matrix")

    def drain(self):
        """Wait for every enqueued iteration; returns the history (total log-likelihood before each update)."""
        if self.session is not None:
            self._drain_to(self.session.iterations_done)
        return self.history

    def _is_writer(self):
        red = self.reducer
        if getattr(red, "native", False):
            return red.rank == 0
        return not (red.enabled and red.dist.get_rank() != 0)

    def models(self):
        """The current parameters as reference-compatible HMM objects (one per word)."""
        from .model_io import models_from_arrays
        shp = (self.W, self.n, self.M, self.D)
        return models_from_arrays(self.means.reshape(shp), self.vars.reshape(shp), self.weights.reshape(shp[:3]),
                                  self.transitions, mu=self.means.reshape(shp)[:, :, 0], sigma=self.vars.reshape(shp)[:, :, 0])

    def save(self, output_path):
        """`<word index>.pkl` per model, like continuous_speech.py:167-170."""
        import os
        import pickle
        os.makedirs(output_path, exist_ok=True)
        for i, m in enumerate(self.models()):
            with open(os.path.join(output_path, str(i) + '.pkl'), 'wb') as f:
                pickle.dump(m, f)

    def fit(self, n_iterations=5, until_converged=False):
        """n_iterations EM iterations; with until_converged the loop ends early once an iteration leaves every mixture
        allclose to the previous one (continuous_speech.py:172-179), the reference's max_iteration otherwise."""
        stream_all = self.session is not None and not until_converged and self.output_path is None
        for _ in range(n_iterations):
            self.iteration(sync=not stream_all)      # nothing on the host needs the result in between: enqueue them all
            if until_converged and self.converged:
                break
        return self.drain()

    def close(self):
        if self.session is not None:
            self._pull()
            self.session.close()
            self.session = None
        if self._gmm is not None:
            self._gmm.close()
            self._gmm = None
        self.batch.close()
        if self.lat is not None:
            self.lat.close()
            self.lat = None
