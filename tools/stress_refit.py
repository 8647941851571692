# -*- coding: utf-8 -*-
"""Randomised sweep of the device-resident refit: the streaming kernels (ordinary + TAIL launches, the eight-wave row sums) against
the tile kernels (GMMHMM_REFIT=tiles, GMMHMM_SEGSUM=narrow) on random shapes -- states of 0 .. 40 000 frames, D = 2 .. 64,
k = 1 .. 8 -- same assignments, iteration counts, centroid bits and converged_at, mixtures to 1e-8.  Where the two forms' EM differs the
oracle decides, state by state: a difference of the TILE kernels is a note (known: a component collapsed onto one frame -- the
reference's linear-domain densities overflow to NaN, the streaming kernels follow it, the tile kernels keep finite weights), a
difference of the streaming kernels fails the run.

    python tools/stress_refit.py [trials] [seed]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-recognition_amd"))
from sr.recognition import _hip

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = _hip.default_context()


def run(mode, b, off, k, c0, part, n_frames, km_it, em_it):
    for name, val in (("GMMHMM_REFIT", "tiles"), ("GMMHMM_SEGSUM", "narrow")):
        if mode == "tiles":
            os.environ[name] = val
        else:
            os.environ.pop(name, None)
    fit = _hip.FitSession(ctx, b, off, max(k, 2))
    try:
        sums, cnts = fit.segment_means()
        cen, cov, cnt, its = fit.kmeans(k, c0, part, max_iteration=km_it)
        ids = fit.clusters()
        with np.errstate(all="ignore"):
            w = cnt / n_frames[:, None]
        mean, var, ww = cen.copy(), cov.copy(), w.copy()
        mu_old, sg_old, w_old = np.zeros_like(mean), np.ones_like(mean), np.zeros_like(ww)
        ok = np.isfinite(mean).all() and np.isfinite(var).all() and (var > 0).all() and np.isfinite(ww).all()
        conv = fit.em(k, mean, var, ww, mu_old, sg_old, w_old, n_frames, max_iteration=em_it) if ok else None
    finally:
        fit.close()
    return dict(sums=sums, cen=cen, cov=cov, cnt=cnt, its=its, ids=ids, mean=mean, var=var, w=ww, conv=conv)


bad = 0
t_start = time.time()
for trial in range(trials):
    S = int(rng.integers(1, 40))
    D = int(rng.choice([2, 3, 7, 13, 16, 39, 40, 47, 63, 64]))
    k = int(rng.choice([1, 2, 3, 4, 5, 8]))
    pool = [0, 1, 2, 15, 16, 17, 64, 511, 512, 513]
    lens = np.array([int(rng.choice(pool)) if rng.random() < 0.3 else int(rng.integers(20, 4000)) for _ in range(S)])
    if rng.random() < 0.15:
        lens[int(rng.integers(0, S))] = int(rng.integers(20000, 40000))
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    N = int(off[-1])
    if N == 0:
        continue
    X = np.empty((N, D))
    c0 = np.zeros((S, k, D))
    for s in range(S):
        cl = rng.normal(size=(k, D)) * rng.uniform(0.5, 3.0) + rng.normal(size=D) * rng.uniform(0, 8.0)
        which = rng.integers(0, k, size=lens[s])
        X[off[s]:off[s + 1]] = cl[which] + rng.normal(size=(lens[s], D)) * rng.uniform(0.3, 1.5, size=D)
        c0[s] = cl + rng.normal(size=(k, D)) * 0.3
    part = rng.integers(0, k, size=N).astype(np.uint8)
    n_frames = np.diff(off).astype(np.float64)
    km_it, em_it = int(rng.choice([1, 3, 40, 200])), int(rng.choice([1, 5, 30, 100]))
    b = _hip.Batch(ctx, feats=X, offsets=[0, N])
    def guarded(mode):
        try:
            return run(mode, b, off, k, c0, part, n_frames, km_it, em_it)
        except np.linalg.LinAlgError:
            return "LinAlgError"
    try:
        new, old = guarded("mfma"), guarded("tiles")
    finally:
        b.close()
    if isinstance(new, str) or isinstance(old, str):
        if new != old:
            # the oracle decides: does the reference raise for one of the states (hmm_state.py:24-30, 149)?
            from oracle import ref_numpy as O
            ok_run = old if isinstance(new, str) else new
            raises, nans, unchecked = [], [], 0
            for s in range(S):
                if lens[s] > 3500 or lens[s] == 0:
                    unchecked += lens[s] > 3500
                    continue
                m, v = ok_run["cen"][s].copy(), ok_run["cov"][s].copy()
                with np.errstate(all="ignore"):
                    w = (ok_run["cnt"][s] / n_frames[s]).copy()
                    try:
                        O.gmm_em(X[off[s]:off[s + 1]], m, v, w, k, max_iteration=em_it, old=(np.zeros((k, D)), np.ones((k, D)), np.zeros(k)))
                        if np.isnan(v).any():
                            nans.append(s)
                    except np.linalg.LinAlgError:
                        raises.append(s)
            who = "the streaming kernels" if isinstance(new, str) else "the tile kernels"
            # a component on ONE frame: the reference's variance is exactly 0 (LinAlgError) or, with denormal-small weights left
            # on other frames, ~1e-74 and NaN one iteration later; the streaming update raises for both
            right = (len(raises) + len(nans) > 0) if isinstance(new, str) else len(raises) == 0
            print("trial %d: S=%d D=%d k=%d km_it=%d em_it=%d: singular covariance raised by %s only; the oracle raises for states %s, ends in NaN for %s (%d states too long to check) -> streaming %s" % (
                trial, S, D, k, km_it, em_it, who, raises, nans, unchecked, "as the reference (fails there too)" if right else "DIFFERS from the reference"), flush=True)
            if not right and not unchecked:
                bad += 1
        continue
    problems, notes = [], []
    if not np.array_equal(new["sums"], old["sums"], equal_nan=True):
        problems.append("segment sums")
    if not np.array_equal(new["ids"], old["ids"]):
        problems.append("ids (%d differ)" % int((new["ids"] != old["ids"]).sum()))
    if not np.array_equal(new["its"], old["its"]):
        problems.append("k-means iterations")
    if not np.array_equal(new["cen"], old["cen"], equal_nan=True):
        problems.append("centroid bits")
    if not np.array_equal(new["cnt"], old["cnt"]):
        problems.append("cluster sizes")
    with np.errstate(all="ignore"):
        if not np.allclose(new["cov"], old["cov"], rtol=1e-9, atol=0, equal_nan=True):
            rel = np.abs(new["cov"] - old["cov"]) / np.abs(old["cov"])
            rel[np.isnan(new["cov"]) & np.isnan(old["cov"])] = 0
            rel[np.isnan(rel)] = np.inf
            s_, c_, d_ = np.unravel_index(np.argmax(rel), rel.shape)
            grp = part[off[s_]:off[s_ + 1]] == c_
            ref = np.cov(X[off[s_]:off[s_ + 1]][grp].T).diagonal()[d_] if grp.sum() > 1 else np.nan
            problems.append("partition variances: worst state %d (%d frames) group %d (%d frames) dim %d: streaming %r tiles %r np.cov %r" % (
                s_, lens[s_], c_, int(grp.sum()), d_, new["cov"][s_, c_, d_], old["cov"][s_, c_, d_], ref))
        if (new["conv"] is None) != (old["conv"] is None):
            problems.append("EM ran in one form only")
        elif new["conv"] is not None:
            differ = [s for s in range(S) if new["conv"][s] != old["conv"][s]
                      or not all(np.allclose(new[nm][s], old[nm][s], rtol=1e-8, atol=1e-300, equal_nan=True) for nm in ("mean", "var", "w"))]
            for s in differ:
                # which of the two is the reference's arithmetic?  (the oracle, state by state; small states only: it loops in Python)
                verdict = "not checked against the oracle (%d frames)" % lens[s]
                if lens[s] <= 3000:
                    from oracle import ref_numpy as O
                    seg = X[off[s]:off[s + 1]]
                    m, v = new["cen"][s].copy(), new["cov"][s].copy()
                    w = (new["cnt"][s] / n_frames[s]).copy()
                    oit = O.gmm_em(seg, m, v, w, k, max_iteration=em_it, old=(np.zeros((k, D)), np.ones((k, D)), np.zeros(k)))
                    oconv = oit - 1 if oit < em_it else -1
                    same = lambda r: (r["conv"][s] == oconv or (oconv == -1 and r["conv"][s] == -1)) and all(
                        np.allclose(r[nm][s], ref, rtol=1e-8, atol=1e-300, equal_nan=True) for nm, ref in (("mean", m), ("var", v), ("w", w)))
                    rel = lambda r: max(float(np.nanmax(np.abs(r[nm][s] - ref) / (np.abs(ref) + 1e-300))) if np.isfinite(ref).any() else 0.0
                                        for nm, ref in (("mean", m), ("var", v), ("w", w)))
                    verdict = "%d frames, cluster sizes %s; oracle: converged_at %d%s; streaming %s (max rel %.3g), tiles %s (max rel %.3g); min var oracle %.3g" % (
                        lens[s], new["cnt"][s].astype(int).tolist(), oconv, " (NaN parameters)" if np.isnan(v).any() else "",
                        "== oracle" if same(new) else "DIFFERS", rel(new), "== oracle" if same(old) else "differs", rel(old), float(np.nanmin(v)) if np.isfinite(v).any() else float("nan"))
                    if same(new):
                        notes.append("state %d (%d frames): %s" % (s, lens[s], verdict))
                        continue
                    # a component about to collapse: its variance is 1e-7 or less of the state's -- the centred sums of both
                    # device forms keep eps * spread^2 / var digits of it (8e-6 relative at 1.5e-11), the reference's two passes all
                    with np.errstate(all="ignore"):
                        ratio = np.nanmin(v / np.var(seg, axis=0)[None, :]) if np.isfinite(v).any() else np.nan
                    if ratio < 1e-6 and rel(new) < 64 * 2.2e-16 / ratio:
                        notes.append("state %d (%d frames): a component with %.1e of the state's variance; %s" % (s, lens[s], ratio, verdict))
                        continue
                problems.append("EM state %d: converged_at %d vs %d; %s" % (s, new["conv"][s], old["conv"][s], verdict))
    for nt in notes:
        print("trial %d (S=%d D=%d k=%d km_it=%d em_it=%d): note -- %s" % (trial, S, D, k, km_it, em_it, nt), flush=True)
    if problems:
        bad += 1
        print("trial %d: S=%d D=%d k=%d N=%d km_it=%d em_it=%d lens=%s...: %s" % (trial, S, D, k, N, km_it, em_it, lens[:8].tolist(), "; ".join(problems)), flush=True)
    elif trial % 10 == 0:
        print("trial %d ok (S=%d D=%d k=%d N=%d), %.0f s" % (trial, S, D, k, N, time.time() - t_start), flush=True)
print("%d trials, %d with differences" % (trials, bad))
sys.exit(1 if bad else 0)
