# -*- coding: utf-8 -*-
"""Latency of the one-utterance reference API calls through the mirror package (after warm-up)."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "speech-recognition_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import sr.recognition as R
from test_gpu_api import make_hmm
from conftest import load_golden

g = load_golden("G4_lattice_decode")
W, n = g["means"].shape[:2]
hmms = [make_hmm(R, g["means"][i], g["vars"][i], g["w"][i], g["word_trans"]) for i in range(W)]
def timeit(fn, reps=20):
    for _ in range(3): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3
x = g["K7_x"]
seq, trans, ends = R.build_state_sequences(hmms, [list(range(W))] * 7)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    print("decode_hmm_states, K=7 lattice (R=%d, T=%d): %.2f ms per call (reference: ~9 s per 300 frames)" % (
        len(seq), len(x), timeit(lambda: R.decode_hmm_states(x, seq, trans, end_points=[[e, -1] for e in ends]))))
    print("HMM.evaluate (5 states, T=%d): %.2f ms per call" % (len(x), timeit(lambda: hmms[0].evaluate(x))))
    print("GMM.evaluate (one frame): %.3f ms per call" % timeit(lambda: hmms[0].gmm_states[0].evaluate(x[0]), reps=100))
