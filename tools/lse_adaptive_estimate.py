# -*- coding: utf-8 -*-
"""Would an ADAPTIVE log-sum-exp epilogue pay in the headline kernel?  fp32 exponentials first, a
wave-uniform redo in fp64 wherever the fp32 error could exceed 1e-10 |nll| (the bound the repo's parity tests hold
likelihoods to).  The error of log(s), s = sum exp(a - max), from fp32 terms is <= 6e-8 (s - 1) / s (the maximum's term is
exactly 1): it vanishes only where ONE component carries the mixture.  This script counts, on the headline workload
(configs[1]: 50 states x 8 mixtures x 39 dims, synthetic), how many log-sum-exps -- and how many epilogue tiles (32 frames
x the 2 states of a 16-row operand tile: the unit a wave decides for) -- would have to take the fp64 path.  CPU only.

    python tools/lse_adaptive_estimate.py [frames]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
wl = bench.synth_workload(1002, 200)
W, n, M, D = wl["W"], wl["n"], wl["M"], wl["D"]
S = W * n
X = wl["X"][:n_frames]
means, vars_, w = wl["means"].reshape(S, M, D), wl["vars"].reshape(S, M, D), wl["w"].reshape(S, M)
logc = np.log(w) - 0.5 * (D * np.log(2 * np.pi) + np.log(vars_).sum(axis=2))            # [S, M]
a = np.empty((len(X), S, M))
for s in range(S):
    d = X[:, None, :] - means[s][None]
    a[:, s, :] = logc[s][None] - 0.5 * (d * d / vars_[s][None]).sum(axis=2)
mx = a.max(axis=2)
ssum = np.exp(a - mx[:, :, None]).sum(axis=2)
nll = -(mx + np.log(ssum))
err_bound = 6e-8 * (ssum - 1.0) / ssum                       # |delta log s| with fp32 exponentials (1 ulp each)
need = err_bound > 1e-10 * np.abs(nll)
T = (len(X) // 32) * 32
tiles = need[:T].reshape(T // 32, 32, S // 2, 2).any(axis=(1, 3))
own = wl["words"][np.searchsorted(wl["off"], np.arange(len(X)), side="right") - 1]     # the frame's own word
print("configs[1] model, %d frames x %d states: %.1f %% of the log-sum-exps need fp64 at 1e-10 |nll| (median (s-1)/s = %.3f);"
      % (len(X), S, 100.0 * need.mean(), float(np.median((ssum - 1) / ssum))))
print("epilogue tiles (32 frames x 2 states) with at least one such entry: %.1f %% -> an adaptive, wave-uniform redo would take the"
      " fp64 path on nearly every tile" % (100.0 * tiles.mean()))
for tol in (1e-9, 1e-8, 1e-7):
    nd = err_bound > tol * np.abs(nll)
    tl = nd[:T].reshape(T // 32, 32, S // 2, 2).any(axis=(1, 3))
    print("  at %.0e |nll|: %.1f %% of entries, %.1f %% of tiles" % (tol, 100.0 * nd.mean(), 100.0 * tl.mean()))
