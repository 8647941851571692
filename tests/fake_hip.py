# -*- coding: utf-8 -*-
"""TEST DOUBLE of `sr.recognition._hip` for the GPU-less tier.

Implements the binding's surface (Context, PackedGMM, Batch, Lattices, distance_matrix) on top
of the CPU oracle so that the HOST LOGIC of the mirror package (object packing, lattice building,
segment bookkeeping, training loops, error mapping) can run under `pytest -m "not gpu"`.  It lives
in tests/ and is installed only by a test fixture (monkeypatching `_hip`'s attributes); the product
never imports it, and the GPU tier runs the same API tests against the real library.
"""
import warnings

import numpy as np

from oracle import ref_numpy as O


class Context:
    def __init__(self, device=0):
        self.device = device
        self.h = object()

    def sync(self):
        pass

    def close(self):
        self.h = None


_ctx = Context()


def default_context(device=None):
    return _ctx


class PackedGMM:
    def __init__(self, ctx, mean, var, weight):
        self.mean, self.var, self.w = (np.array(a, dtype=np.float64) for a in (mean, var, weight))
        self.S, self.M, self.D = self.mean.shape
        if np.any(self.var == 0):
            raise np.linalg.LinAlgError("Singular matrix")

    def update(self, mean, var, weight):
        self.__init__(None, mean, var, weight)

    def component_loglik(self, state, x):
        x = np.asarray(x, dtype=np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.log(np.array([O.gmm_evaluate(f, self.mean[state], self.var[state], self.w[state], neg_log=False)
                                    for f in x]))

    def close(self):
        pass


class Batch:
    def __init__(self, ctx, utterances=None, dtype=np.float64, feats=None, offsets=None, cepstra=None, frontend_mode=0,
                 pcm=None, sample_rate=16000, mfcc_params=None):
        self.ctx = ctx
        self.np_dtype = np.dtype(dtype)
        if pcm is not None:
            cepstra = mfcc(ctx, pcm, sample_rate, mfcc_params)[1]
        if cepstra is not None:  # N3 front-end through the oracle
            outs = []
            for c in cepstra:
                c = np.asarray(c, dtype=np.float64)
                if frontend_mode == 2:
                    with np.errstate(divide="ignore", invalid="ignore"):
                        outs.append(O.standardize(c))
                    continue
                df = O.delta_feature(c)
                raw = np.concatenate([c, df, O.delta_feature(df)], axis=1)
                with np.errstate(divide="ignore", invalid="ignore"):
                    outs.append(raw if frontend_mode == 1 else O.standardize(raw))
            utterances = outs
        if utterances is not None:
            lens = [len(u) for u in utterances]
            offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            D = np.asarray(utterances[0]).shape[1] if lens else 1
            feats = np.concatenate([np.asarray(u, dtype=np.float64).reshape(-1, D) for u in utterances]) if lens \
                else np.zeros((0, D))
        self.feats = np.asarray(feats, dtype=np.float64)
        self.offsets = np.asarray(offsets, dtype=np.int64)
        self.N, self.D = self.feats.shape
        self.U = len(self.offsets) - 1
        self.S = None
        self.nll = None
        self.occ = None

    @property
    def lengths(self):
        return np.diff(self.offsets)

    def utt(self, u):
        return self.feats[self.offsets[u]:self.offsets[u + 1]]

    def features(self):
        return [self.utt(u).astype(self.np_dtype) for u in range(self.U)]

    def loglik(self, gmm, fetch=True, state_ranges=None, state_sets=None):
        with np.errstate(divide="ignore", invalid="ignore"):
            self.nll = np.array([[O.gmm_evaluate(x, gmm.mean[s], gmm.var[s], gmm.w[s]) for s in range(gmm.S)]
                                 for x in self.feats]).reshape(self.N, gmm.S)
        self.S = gmm.S
        return self.nll.copy() if fetch else None

    def dtw(self, trans, y=None, var=None, beam=0, dist=None, want_costs=True):
        costs, paths = [], []
        for u in range(self.U):
            if dist is not None:
                E = np.asarray(dist[u], dtype=np.float64)
            else:
                E = O.distance_matrix(self.utt(u), y, "euclid" if var is None else "mahalanobis", var)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                c, p = O.dtw(E, np.asarray(trans), beam=np.inf if beam <= 0 else beam)
            costs.append(c)
            paths.append(p.reshape(-1, 2).astype(np.int64))
        return (costs if want_costs else None), paths

    def kmeans_assign(self, centroids, var=None, first=0, count=None):
        x = self.feats[first:first + (self.N - first if count is None else count)]
        if len(x) == 0:
            return np.zeros(0, dtype=np.int64)
        d = np.array([[O.euclid(c, f) if var is None else O.mahalanobis(c, f, var) for c in centroids] for f in x])
        return np.argmin(d, axis=1).astype(np.int64)

    def em_accumulate(self, mean, var, weight, first=0, count=None):
        x = self.feats[first:first + (self.N - first if count is None else count)]
        mean, var, weight = (np.asarray(a, dtype=np.float64) for a in (mean, var, weight))
        if np.any(var == 0):
            raise np.linalg.LinAlgError("Singular matrix")
        k, D = mean.shape
        with np.errstate(divide="ignore", invalid="ignore"):
            p = np.array([O.gmm_evaluate(f, mean, var, weight, neg_log=False) for f in x]).reshape(len(x), k)
            rs = p.sum(axis=1, keepdims=True)
            r = np.where(rs == 0, 0.0, p / np.where(rs == 0, 1.0, rs))
            stats = np.zeros((k, 1 + 2 * D))
            for c in range(k):
                d = x - mean[c]
                stats[c, 0] = r[:, c].sum()
                stats[c, 1:1 + D] = (r[:, [c]] * d).sum(axis=0)
                stats[c, 1 + D:] = (r[:, [c]] * d * d).sum(axis=0)
            ll = float(np.log(rs[rs > 0]).sum())
        return stats, ll

    def kmeans_assign_multi(self, seg_off, centroids, var=None, clusters=None, active=None, want_sums=False):
        centroids = np.asarray(centroids, dtype=np.float64)
        S, k, D = centroids.shape
        from sr.recognition import _hip as real
        resident = clusters is real.RESIDENT
        if resident:
            if getattr(self, "_clusters", None) is None:
                self._clusters = np.full(self.N, -1, dtype=np.int32)
            clusters = self._clusters
        if clusters is None:
            clusters = np.full(self.N, -1, dtype=np.int32)
        changed = np.zeros(S, dtype=np.int32)
        sums = np.zeros((S, k, D + 1)) if want_sums else None
        for s in range(S):
            if active is not None and not active[s]:
                continue
            a, b = int(seg_off[s]), int(seg_off[s + 1])
            new = self.kmeans_assign(centroids[s], None if var is None else np.asarray(var)[s], first=a, count=b - a).astype(np.int32)
            changed[s] = int(np.sum(new != clusters[a:b]))
            clusters[a:b] = new
            if want_sums:
                for c in range(k):
                    x = self.feats[a:b][new == c]
                    sums[s, c, :D] = x.sum(axis=0)
                    sums[s, c, D] = len(x)
        return (None if resident else clusters), changed, sums

    def bw_accumulate(self, gmm, occ_floor=0.0, stats_dev=None, fetch=True):
        """gh_bw_accumulate: r_nsm = occ[n,s] w_sm pdf_sm(x_n) / sum_m' ...; [S, M, 1 + 2D] centred statistics."""
        S, M, D = gmm.mean.shape
        stats = np.zeros((S, M, 1 + 2 * D))
        for s in range(S):
            fr = np.flatnonzero(self.occ[:, s] > occ_floor)
            if len(fr) == 0:
                continue
            x = self.feats[fr]
            with np.errstate(all="ignore"):
                ll = np.stack([np.log(gmm.w[s, m]) - 0.5 * (D * np.log(2 * np.pi) + np.log(gmm.var[s, m]).sum()
                                                                + ((x - gmm.mean[s, m]) ** 2 / gmm.var[s, m]).sum(axis=1)) for m in range(M)], axis=1)
                ll -= ll.max(axis=1, keepdims=True)
                r = np.exp(ll)
                r = r / r.sum(axis=1, keepdims=True) * self.occ[fr, s][:, None]
            for m in range(M):
                d = x - gmm.mean[s, m]
                stats[s, m, 0] = r[:, m].sum()
                stats[s, m, 1:1 + D] = (r[:, [m]] * d).sum(axis=0)
                stats[s, m, 1 + D:] = (r[:, [m]] * d * d).sum(axis=0)
        return stats

    def gather(self, rows, offsets=None):
        rows = np.asarray(rows, dtype=np.int64)
        return Batch(self.ctx, feats=self.feats[rows], offsets=[0, len(rows)] if offsets is None else offsets)

    def resident_clusters(self, reset=False, fetch=True):
        if reset or getattr(self, "_clusters", None) is None:
            self._clusters = np.full(self.N, -1, dtype=np.int32)
        return self._clusters.copy() if fetch else None

    def em_accumulate_multi(self, seg_off, mean, var, weight, active=None, stats_dev=None):
        mean = np.asarray(mean, dtype=np.float64)
        S, k, D = mean.shape
        stats, ll = np.zeros((S, k, 1 + 2 * D)), np.zeros(S)
        for s in range(S):
            if active is not None and not active[s]:
                continue
            a, b = int(seg_off[s]), int(seg_off[s + 1])
            stats[s], ll[s] = self.em_accumulate(mean[s], np.asarray(var)[s], np.asarray(weight)[s], first=a, count=b - a)
        return stats, ll

    def close(self):
        pass


class Lattices:
    def __init__(self, ctx, graphs):
        self.ctx = ctx
        self.graphs = graphs
        self.L = len(graphs)
        self.R = [len(g["row_state"]) for g in graphs]
        self.n_end = [len(g["end_rows"]) for g in graphs]
        self.beam = None

    @staticmethod
    def flatten_transcripts(label_seqs):
        return None

    @classmethod
    def from_transcripts(cls, ctx, word_transitions, n, label_seqs=None, state_base=None, flat=None):
        from sr.recognition.continuous_speech import packed_lattice
        return cls(ctx, [packed_lattice(word_transitions, n, [[int(l)] for l in labels], state_base=state_base)[0]
                         for labels in label_seqs])

    def set_beam(self, beam):
        self.beam = None if (beam is None or beam <= 0 or beam == float("inf")) else int(beam)

    def _dense(self, g):
        R = len(g["row_state"])
        t = np.full((R, R), np.inf)
        t[np.asarray(g["arc_to"], dtype=int), np.asarray(g["arc_from"], dtype=int)] = g["arc_cost"]
        return t

    def _emissions(self, batch, u, g):
        rs = np.asarray(g["row_state"], dtype=int)
        nll = batch.nll[batch.offsets[u]:batch.offsets[u + 1]]
        return np.where(rs[:, None] >= 0, nll[:, np.maximum(rs, 0)].T, 0.0), rs < 0

    def path_cap(self, l, T):
        return 3 * T

    def viterbi(self, batch, utt_lattice=None, want_path=True, want_costs=False, want_end_cost=True, fused_gmm=None,
                log_domain=False):
        """Each start row is decoded as its own reference problem (the reference has ONE start cell,
        row 0; a stacked graph is W independent chains), then merged.  fused_gmm: the double of gh_viterbi_fused --
        the oracle scores the cells first (mahalanobis for log_domain, the linear-domain GMM.evaluate otherwise)."""
        if fused_gmm is not None:
            g = fused_gmm
            if log_domain:
                batch.nll = np.array([[O.mahalanobis(x, g.mean[s, 0], g.var[s, 0]) - np.log(g.w[s, 0]) for s in range(g.S)]
                                      for x in batch.feats]).reshape(batch.N, g.S)
                batch.S = g.S
            else:
                batch.loglik(g, fetch=False)
        U = batch.U
        lidx = np.zeros(U, dtype=int) if utt_lattice is None else np.asarray(utt_lattice, dtype=int)
        out = dict(best_end=np.zeros(U, dtype=np.int32), end_cost=[], paths=[], costs=[])
        for u in range(U):
            g = self.graphs[lidx[u]]
            E, nes = self._emissions(batch, u, g)
            trans = self._dense(g)
            R, T = E.shape
            starts = [int(s) for s in g["start_rows"]]
            ends = [int(e) for e in g["end_rows"]]
            if T == 0:
                out["best_end"][u] = -1
                out["end_cost"].append(np.full(len(ends), np.inf))
                out["paths"].append(np.zeros((0, 2), dtype=np.int64))
                out["costs"].append(np.zeros((R, 0)))
                continue
            # the reference has ONE start cell, row 0; a graph with several start rows is that many
            # independent sub-graphs [start, next start): fill each with the reference sweep
            costs = np.full((R, T), np.inf)
            bp = np.full((R, T, 2), O._NOPTR, dtype=np.int64)
            ss = sorted(starts)
            for k, s0 in enumerate(ss):
                s1 = ss[k + 1] if k + 1 < len(ss) else R
                lo = 0 if k == 0 else s0
                sl = slice(lo, s1)
                cs, bps = O.decode_fill(E[sl], nes[sl], trans[sl, sl], beam=self.beam)
                costs[sl] = cs
                bps = bps.copy()
                bps[:, :, 0] = np.where(bps[:, :, 0] == O._NOPTR, O._NOPTR, bps[:, :, 0] + lo)
                bp[sl] = bps
            ec = np.array([costs[e, T - 1] for e in ends])
            best, bi = np.inf, -1
            for k, cst in enumerate(ec):
                if best >= cst:
                    best, bi = cst, k
            out["best_end"][u] = bi
            out["end_cost"].append(ec)
            out["costs"].append(costs)
            path = []
            if want_path and T > 1 and bi >= 0:
                i, j = ends[bi], T - 1
                while j != 0:
                    i, j = bp[i, j]
                    path.append([i, j])
                    if len(path) > R * T:
                        raise RuntimeError("back-trace does not terminate")
            out["paths"].append(np.array(path, dtype=np.int64).reshape(-1, 2))
        out["end_cost_flat"] = np.concatenate(out["end_cost"]) if out["end_cost"] else np.zeros(0)
        return out

    def viterbi_labels(self, batch, row_label, utt_lattice=None, max_labels=None, as_lists=True, want_end_cost=True):
        if isinstance(row_label, np.ndarray) and row_label.ndim == 1 and self.L == 1:
            row_label = [row_label]
        r = self.viterbi(batch, utt_lattice=utt_lattice, want_path=True)
        lidx = np.zeros(batch.U, dtype=int) if utt_lattice is None else np.asarray(utt_lattice, dtype=int)
        labels = []
        for u, p in enumerate(r["paths"]):
            rl = np.asarray(row_label[lidx[u]])
            labels.append(np.array(O.path_to_words(p, rl < 0, rl), dtype=np.int32) if len(p) else np.zeros(0, dtype=np.int32))
        return dict(labels=labels, best_end=r["best_end"], end_cost_flat=r["end_cost_flat"])

    def forward_backward(self, batch, utt_lattice=None, want_matrices=False, want_occ=False, fetch_occ=True, want_self_xi=False):
        """gh_forward_backward through the oracle (O.forward_backward): log P, occupancies [N,S], expected self
        transitions xi_t(r -> r) = alpha_t(r) a_rr b_r(x_{t+1}) beta_{t+1}(r) / P summed per state."""
        U = batch.U
        lidx = np.zeros(U, dtype=int) if utt_lattice is None else np.asarray(utt_lattice, dtype=int)
        S = batch.nll.shape[1]
        logp = np.empty(U)
        occ = np.zeros((batch.N, S))
        xi = np.zeros(S)
        mats = dict(alpha=[], beta=[], gamma=[])
        for u in range(U):
            g = self.graphs[lidx[u]]
            E, nes = self._emissions(batch, u, g)
            trans = self._dense(g)
            rs = np.asarray(g["row_state"], dtype=int)
            if E.shape[1] == 0:
                logp[u] = -np.inf
                continue
            with np.errstate(all="ignore"):
                la, lb, gamma, lp = O.forward_backward(E, nes, trans, [int(e) for e in g["end_rows"]])
            logp[u] = lp
            f0 = int(batch.offsets[u])
            for r in np.flatnonzero(rs >= 0):
                occ[f0:f0 + E.shape[1], rs[r]] += gamma[r]
                if np.isfinite(trans[r, r]) and np.isfinite(lp):
                    with np.errstate(all="ignore"):
                        x = np.exp(la[r, :-1] - trans[r, r] - E[r, 1:] + lb[r, 1:] - lp)
                    xi[rs[r]] += np.nansum(x)
            mats["alpha"].append(la); mats["beta"].append(lb); mats["gamma"].append(gamma)
        batch.occ = occ
        out = dict(logp=logp)
        if want_self_xi:
            out["self_xi"] = xi
        if want_matrices:
            out.update(mats)
        if want_occ and fetch_occ:
            out["occ"] = occ
        return out

    def align_segments(self, batch, utt_lattice=None):
        """gh_align_segments: the reference's own regrouping loop (continuous_speech.py:90-106) over the oracle's paths."""
        r = self.viterbi(batch, utt_lattice=utt_lattice, want_path=True)
        lidx = np.zeros(batch.U, dtype=int) if utt_lattice is None else np.asarray(utt_lattice, dtype=int)
        fs = np.full(batch.N, -1, dtype=np.int32)
        start = np.zeros(batch.N, dtype=bool)
        for u, path in enumerate(r["paths"]):
            rs = np.asarray(self.graphs[lidx[u]]["row_state"])
            f0 = int(batch.offsets[u])
            open_row, open_at = None, None
            for row, c in path[::-1]:
                if open_at is None and rs[row] >= 0:
                    open_row, open_at = row, c
                if row != open_row and open_at is not None and open_at < c:
                    fs[f0 + open_at:f0 + c] = rs[open_row]
                    start[f0 + open_at] = True
                    open_row, open_at = None, None
        return dict(frame_state=fs, segment_start=start, end_cost_flat=r["end_cost_flat"], best_end=r["best_end"])

    def close(self):
        pass


def mfcc(ctx, signals, sample_rate=16000, mfcc_params=None):
    fs, st, lo, hi = mfcc_params if mfcc_params is not None else (0.025, 0.01, 80, None)
    fbs, mfs = [], []
    for x in signals:
        x = np.asarray(x).reshape(-1)
        if len(x) == 0:
            raise IndexError("index 0 is out of bounds for axis 0 with size 0")
        fb, mf = O.mfcc_features_signal(x, sample_rate, fs, st, lo, hi)
        fbs.append(fb)
        mfs.append(mf)
    return fbs, mfs


def distance_matrix(ctx, x, y, var=None):
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    v = None if var is None else np.asarray(var, dtype=np.float64).reshape(-1, y.shape[1])
    out = np.empty((len(y), len(x)))
    for i in range(len(y)):
        for n in range(len(x)):
            out[i, n] = O.euclid(x[n], y[i]) if v is None else O.mahalanobis(x[n], y[i], v[0] if len(v) == 1 else v[i])
    return out


class EMSession:
    """The device-resident EM session has no test double: callers keep the call-by-call path."""

    def __init__(self, *a, **k):
        from sr.recognition import _hip
        raise _hip.Unsupported("no device-resident session on the test double")


class FitSession:
    """No test double of the device-resident refit: callers keep the host loop (and the frames on the host)."""
    available = False

    def __init__(self, *a, **k):
        from sr.recognition import _hip
        raise _hip.Unsupported("no device-resident refit on the test double")


NAMES = ("Context", "PackedGMM", "Batch", "Lattices", "distance_matrix", "default_context", "mfcc", "EMSession", "FitSession")


def install(monkeypatch, hip_module):
    for name in NAMES:
        monkeypatch.setattr(hip_module, name, globals()[name])
