# -*- coding: utf-8 -*-
"""Phases of the device-resident EM iteration over word strings (bench.py's C3_word_strings shape): HIP events between the
phases (gh_em_profile) and the wall time per iteration, enqueued and synchronous.  U / K from the environment."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-recognition_amd"))
import bench
from sr.recognition.train import BaumWelchTrainer

U, K = int(os.environ.get("U", "2000")), int(os.environ.get("K", "7"))
wl = bench.synth_workload(1003, U * K)
W = wl["W"]
off = wl["off"][::K]
labels = [[int(w) for w in wl["words"][i * K:(i + 1) * K]] for i in range(U)]
data = [wl["X"][off[u]:off[u + 1]] for u in range(U)]
means0 = wl["means"] + 0.3 * np.random.default_rng(0).normal(size=wl["means"].shape)
tr = BaumWelchTrainer(means0, wl["vars"], wl["w"], [wl["trans"]] * W, data, labels)
assert tr.session is not None and tr.session.word_strings
for _ in range(3):
    tr.iteration()
t0 = time.perf_counter()
for _ in range(10):
    tr.iteration()
sync_ms = (time.perf_counter() - t0) * 100
t0 = time.perf_counter()
for _ in range(10):
    tr.iteration(sync=False)
tr.drain()
enq_ms = (time.perf_counter() - t0) * 100
tr.session.profile(True)
ph = []
for _ in range(5):
    tr.iteration()
    ph.append(tr.session.phase_ms())
ph = np.median(ph, axis=0)
print("frames %d  sync %.3f ms  enqueued %.3f ms  phases: loglik %.3f  fb_seq+ranges %.3f  statistics %.3f  tail..repack %.3f"
      % (tr.batch.N, sync_ms, enq_ms, *ph))
tr.close()
