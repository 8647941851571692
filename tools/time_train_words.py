#!/usr/bin/env python3
"""Isolated-word training of all word models in one pass (batch.train_words): wall time and host profile.
usage: time_train_words.py [words] [templates per word]"""
import cProfile, io, os, pstats, sys, time, contextlib, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-recognition_amd")):
    sys.path.insert(0, p)
import numpy as np
import bench
from sr.recognition.batch import train_words
W = int(sys.argv[1]) if len(sys.argv) > 1 else 10
R = int(sys.argv[2]) if len(sys.argv) > 2 else 200
wl = bench.synth_workload(1006, W * R, W=W, n=5, M=int(os.environ.get("TW_M", "4")))
order = np.argsort(wl["words"], kind="stable")
words = [[wl["X"][wl["off"][u]:wl["off"][u + 1]] for u in order[wl["words"][order] == w]] for w in range(W)]
def run():
    np.random.seed(0)
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return train_words(words, 5, n_gaussians=4)
run()
pr = cProfile.Profile()
t0 = time.perf_counter(); pr.enable(); run(); pr.disable(); dt = time.perf_counter() - t0
print("%d words x %d templates, %d frames: %.1f ms" % (W, R, int(wl["off"][-1]), dt * 1e3))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:5000])
