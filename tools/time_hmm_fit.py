#!/usr/bin/env python3
"""Isolated-word training (`core.train` -> `HMM.fit`, hmm.py:57-124: segmental k-means, then all states' mixtures in
lock-step, then the re-alignment) on synthetic templates: wall time per word model and where the host spends it.
usage: time_hmm_fit.py [templates per word] [words]"""
import cProfile, io, os, pstats, sys, time, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-recognition_amd")):
    sys.path.insert(0, p)
import numpy as np
import bench
import sr.recognition as R

NT = int(sys.argv[1]) if len(sys.argv) > 1 else 200
NW = int(sys.argv[2]) if len(sys.argv) > 2 else 3
wl = bench.synth_workload(1003, NT * 10)
n, M = wl["n"], wl["M"]
by_word = [[wl["X"][wl["off"][u]:wl["off"][u + 1]] for u in np.flatnonzero(wl["words"] == w)[:NT]] for w in range(NW)]
print("%d words x %d templates (%d frames per word), %d states, %d mixtures, 39-dim" % (
    NW, NT, int(np.mean([sum(len(x) for x in ys) for ys in by_word])), n, M), flush=True)
np.random.seed(0)
pr = cProfile.Profile()
t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    pr.enable()
    models = [R.HMM(n).fit(ys, M) for ys in by_word]
    pr.disable()
dt = time.perf_counter() - t0
print("%.2f s per word model" % (dt / NW))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(24)
print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:5500])
costs = [m.evaluate(by_word[w][0]) for w, m in enumerate(models)]
print("evaluate() of the first template under its own model:", np.round(costs, 2))
