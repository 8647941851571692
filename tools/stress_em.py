# -*- coding: utf-8 -*-
"""Random shapes through the device-resident EM iteration (own-state likelihoods -> chain forward-backward, one- or two-way,
8 or 16 lanes -> statistics kernel, pairs or 16-component pieces, block lists from occupancy ranges -> M-step) against the
call-by-call trainer forced onto the any-graph forward-backward and the generic statistics kernel.  Utterances shorter
than the chain (no path), one-frame utterances, skip arcs, switched-off components, words nobody uttered.
usage: stress_em.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-recognition_amd")):
    sys.path.insert(0, p)
from sr.recognition.train import BaumWelchTrainer

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
DIMS = [1, 3, 4, 5, 8, 13, 16, 21, 24, 37, 39, 40]
worst = 0.0
for case in range(cases):
    W = int(rng.integers(1, 9)); n = int(rng.integers(2, 17)); M = int(rng.choice([1, 2, 3, 5, 8, 9, 12, 16, 17, 31, 32, 33, 48, 64]))
    D = int(rng.choice(DIMS)); U = int(rng.choice([1, 3, 63, 64, 65, 130, 300]))
    skip = bool(rng.integers(0, 2)) and n >= 3
    if M * (D + 1) > 2048:            # (the generic statistics kernel -- the checker here -- stops at M (D + 1) <= 2048)
        M = 2048 // (D + 1)
    means = rng.normal(size=(W, n, M, D)) * 2.0
    vars_ = rng.uniform(0.5, 1.5, size=(W, n, M, D))
    w = rng.dirichlet(np.ones(M), size=(W, n))
    if M > 1 and rng.integers(0, 2):
        w[0, 0, 0] = 0.0
        w[0, 0] /= w[0, 0].sum()
    trans = np.full((n, n), np.inf)
    for i in range(n):
        trans[i, i] = -np.log(0.7) if i < n - 1 else 0.0
        if i < n - 1:
            trans[i + 1, i] = -np.log(0.25 if skip else 0.3)
        if skip and i < n - 2:
            trans[i + 2, i] = -np.log(0.05)
    words = rng.integers(0, W, size=U)
    data, labels = [], []
    for u in range(U):
        kind = rng.integers(0, 10)
        T = 1 if kind == 0 else (int(rng.integers(1, n + 1)) if kind == 1 else int(rng.integers(n, 6 * n + 2)))
        st = np.minimum(np.arange(T) * n // max(T, 1), n - 1)
        comp = rng.integers(0, M, size=T)
        data.append(means[words[u], st, comp] + np.sqrt(vars_[words[u], st, comp]) * rng.normal(size=(T, D)))
        labels.append([int(words[u])])
    for form in ("1", "2"):
        os.environ["GMMHMM_FBCHAIN"] = form
        a = BaumWelchTrainer(means + 0.2, vars_, w, [trans] * W, data, labels)
        assert a.session is not None, (W, n, M, D)
        la = a.iteration()
        packed = a.session.packed()
        del os.environ["GMMHMM_FBCHAIN"]
        b = BaumWelchTrainer(means + 0.2, vars_, w, [trans] * W, data, labels, device_resident=False)
        os.environ["GMMHMM_FB"] = "generic"; os.environ["GMMHMM_BW"] = "generic"
        try:
            stats, xi, ll = b.e_step()
            lb = b.iteration()
        finally:
            del os.environ["GMMHMM_FB"], os.environ["GMMHMM_BW"]
        tag = "case %d form %s: W=%d n=%d M=%d D=%d U=%d skip=%s" % (case, form, W, n, M, D, U, skip)
        got = packed[:a.n_stats].reshape(stats.shape)
        scale = np.maximum(np.abs(stats), 1e-6)
        err = float(np.max(np.abs(got - stats) / scale))
        worst = max(worst, err)
        assert err < 1e-6, (tag, err)
        np.testing.assert_allclose(packed[a.n_stats:a.n_stats + a.S], xi, rtol=1e-7, atol=1e-9, err_msg=tag)
        if np.isfinite(ll) and ll != 0.0:
            assert abs(packed[a.n_stats + a.S] - ll) <= 1e-10 * abs(ll), (tag, packed[a.n_stats + a.S], ll)
            assert abs(la - lb) <= 1e-10 * abs(lb), (tag, la, lb)
        np.testing.assert_allclose(a.means, b.means, rtol=1e-6, atol=1e-8, err_msg=tag)
        np.testing.assert_allclose(a.vars, b.vars, rtol=1e-5, atol=1e-12, err_msg=tag)
        np.testing.assert_allclose(a.weights, b.weights, rtol=1e-6, atol=1e-10, err_msg=tag)
        a.close(); b.close()
    print(tag, "ok", flush=True)
print("stress_em ok: %d cases, worst relative statistics error %.2e" % (cases, worst))
