# -*- coding: utf-8 -*-
"""Front-end timing: U utterances of 1 s int16 audio -> resident 39-dim batch (MFCC, deltas, standardise)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "speech-recognition_amd"))
from sr.recognition import _hip

U = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(0)
sig = (rng.normal(size=(U, 16000)) * 3000).astype(np.int16)
sigs = list(sig)
ctx = _hip.default_context(0)
for k in range(4):
    t0 = time.perf_counter()
    b = _hip.Batch(ctx, pcm=sigs, sample_rate=16000)
    ctx.sync()
    dt = time.perf_counter() - t0
    print("run %d: %d utterances, %d frames: %.1f ms wall (incl. host packing + %.0f MB H2D) = %.2e frames/s" % (
        k, U, b.N, dt * 1e3, sig.nbytes / 1e6, b.N / dt), flush=True)
    b.close()
