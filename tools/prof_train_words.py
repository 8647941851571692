# -*- coding: utf-8 -*-
"""cProfile of batch.train_words on bench.py's C2_train_words workload (10 words x 200 templates): where the host spends it.
The workload is built outside the profile; REPS (default 3) calls are profiled after one warm-up call."""
import contextlib, cProfile, io, os, pstats, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-recognition_amd"))
import numpy as np
import bench
from sr.recognition import _hip
from sr.recognition.batch import train_words
ctx = _hip.default_context()
W, templates, n, ng = 10, 200, 5, 4
wl = bench.synth_workload(1006, W * templates, W=W, n=n, M=ng)
order = np.argsort(wl["words"], kind="stable")
words = [[wl["X"][wl["off"][u]:wl["off"][u + 1]] for u in order[wl["words"][order] == w]] for w in range(W)]


def run():
    np.random.seed(0)
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return train_words(words, n, n_gaussians=ng)


run()
reps = int(os.environ.get("REPS", "3"))
plain = []
for _ in range(reps):
    t0 = time.perf_counter()
    run()
    plain.append((time.perf_counter() - t0) * 1e3)
print("train_words, %d calls without the profiler [ms]: %s" % (reps, " ".join("%.1f" % v for v in plain)))
pr = cProfile.Profile()
pr.enable()
for _ in range(reps):
    run()
pr.disable()
for key, count in (("cumulative", 30), ("tottime", 30)):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(count)
    print("\n".join(l for l in s.getvalue().splitlines() if l.strip()))
os.environ["GMMHMM_TRAIN_WORDS_TIMES"] = "1"
for _ in range(3):
    t0 = time.perf_counter()
    m = run()
    t1 = time.perf_counter()
    del m
    print("call %.1f ms, dropping the result %.1f ms" % ((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))
print("transparent hugepages:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
