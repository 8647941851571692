# -*- coding: utf-8 -*-
"""Randomised sweep of `continuous_train` (embedded Viterbi training, continuous_speech.py:56-179): the device-resident path
(alignment as runs, frames gathered on the device, streaming refit with TAIL launches, partitions drawn ahead, pickles on a
worker) against `compat_cov=True` -- np.cov partition variances, frames and refit loop on the host, the path of round 2 --
under the same numpy seed, on random vocabularies and transcripts.  The pickled models of every outer iteration: transition
costs and mixture weights to 1e-9, means / variances to 1e-7 (np.cov's BLAS summation order against the device's; cluster ids are
expected identical), the same exception where one is raised.

    python tools/stress_ctrain.py [trials] [seed]
"""
import contextlib
import io
import os
import pickle
import shutil
import sys
import tempfile
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-recognition_amd"))
import sr.recognition as R
from sr.recognition import _pack

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        yield


def utt(means, vars_, words, n, M, tmin, tmax):
    segs = []
    for wd in words:
        T = int(rng.integers(tmin, tmax))
        st = np.minimum(np.arange(T) * n // T, n - 1)
        comp = rng.integers(0, M, size=T)
        segs.append(means[wd, st, comp] + np.sqrt(vars_[wd, st, comp]) * rng.normal(size=(T, means.shape[-1])))
    return np.concatenate(segs)


def run(models, data, strings, ng, n, iters, seed, compat):
    out = tempfile.mkdtemp(prefix="gmmhmm_stress_ctrain_")
    try:
        np.random.seed(seed)
        try:
            with quiet():
                R.continuous_train(data, pickle.loads(pickle.dumps(models)), strings, out, n_gaussians=ng, n_segments=n, max_iteration=iters, compat_cov=compat)
        except (np.linalg.LinAlgError, IndexError, AssertionError, ValueError) as e:
            if os.environ.get("STRESS_TRACE"):
                import traceback
                traceback.print_exc()
            return type(e).__name__
        return [pickle.load(open(os.path.join(out, "%d.pkl" % w), "rb")) for w in range(len(models))]
    finally:
        shutil.rmtree(out, ignore_errors=True)


bad = 0
t0 = time.time()
for trial in range(trials):
    W, n, M, D = int(rng.integers(2, 6)), int(rng.integers(2, 6)), 2, int(rng.choice([4, 10, 13]))
    K = int(rng.integers(1, 5))
    ng = int(rng.choice([4, 4, 8]))
    means = rng.normal(size=(W, n, M, D)) * 2.5
    vars_ = rng.uniform(0.5, 1.2, size=(W, n, M, D))
    np.random.seed(int(rng.integers(0, 1 << 30)))
    try:
        with quiet():
            models = [R.HMM(n).fit([utt(means, vars_, [w], n, M, 24, 40) for _ in range(int(rng.integers(6, 14)))], ng) for w in range(W)]
    except (np.linalg.LinAlgError, IndexError):
        continue
    for h in models:
        for s in h.gmm_states:
            s.parent = h
    strings = [[int(v) for v in rng.integers(0, W, size=int(rng.integers(1, K + 1)))] for _ in range(int(rng.integers(15, 60)))]
    data = [utt(means, vars_, s, n, M, 24, 40) for s in strings]
    iters, seed = int(rng.integers(1, 4)), int(rng.integers(0, 1 << 30))
    if os.environ.get("STRESS_ONLY") and trial != int(os.environ["STRESS_ONLY"]):
        continue
    dev = run(models, data, strings, ng, n, iters, seed, False)
    host = run(models, data, strings, ng, n, iters, seed, True)
    problems = []
    if isinstance(dev, str) or isinstance(host, str):
        if dev != host:
            problems.append("device path: %s, host path: %s" % (dev if isinstance(dev, str) else "ok", host if isinstance(host, str) else "ok"))
    else:
        for w, (a, b) in enumerate(zip(dev, host)):
            with np.errstate(all="ignore"):
                if not np.allclose(a.transitions, b.transitions, rtol=1e-9, equal_nan=True):
                    problems.append("word %d: transition costs" % w)
                (ma, va, wa), (mb, vb, wb) = _pack.stack_gmms(a.gmm_states), _pack.stack_gmms(b.gmm_states)
                if not np.allclose(wa, wb, rtol=1e-9, equal_nan=True):
                    problems.append("word %d: weights (max diff %.3g)" % (w, float(np.nanmax(np.abs(wa - wb)))))
                elif not (np.allclose(ma, mb, rtol=1e-7, atol=1e-9, equal_nan=True) and np.allclose(va, vb, rtol=1e-7, equal_nan=True)):
                    problems.append("word %d: means / variances (max rel %.3g)" % (w, float(np.nanmax(np.abs(va - vb) / np.abs(vb)))))
    if problems:
        bad += 1
        print("trial %d: W=%d n=%d D=%d K<=%d ng=%d utterances=%d iterations=%d seed=%d: %s" % (trial, W, n, D, K, ng, len(data), iters, seed, "; ".join(problems[:4])), flush=True)
    elif trial % 5 == 0:
        print("trial %d ok (W=%d n=%d D=%d ng=%d, %d utterances, %d iterations%s), %.0f s" % (
            trial, W, n, D, ng, len(data), iters, ", both raise " + dev if isinstance(dev, str) else "", time.time() - t0), flush=True)
print("%d trials, %d with differences" % (trials, bad))
sys.exit(1 if bad else 0)
