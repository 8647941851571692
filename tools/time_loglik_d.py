#!/usr/bin/env python3
"""The likelihood kernel at feature dimensions between and beyond the instantiated operand lengths: matrix-core kernel
(operands zero-padded to the next instantiated length) against the vector kernel (GMMHMM_LOGLIK=valu, read once per
process: run twice).  usage: [GMMHMM_LOGLIK=valu] time_loglik_d.py [utterances]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-recognition_amd")):
    sys.path.insert(0, p)
import numpy as np
import bench
from sr.recognition import _hip
U = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
ctx = _hip.Context(0)
for D in (20, 30, 39, 50, 64):
    wl = bench.synth_workload(1000 + D, U, D=D)
    W, n, M = wl["W"], wl["n"], wl["M"]
    S = W * n
    gmm = _hip.PackedGMM(ctx, wl["means"].reshape(S, M, D), wl["vars"].reshape(S, M, D), wl["w"].reshape(S, M))
    b = _hip.Batch(ctx, feats=wl["X"], offsets=wl["off"])
    b.loglik(gmm, fetch=False); ctx.sync()
    t_r = time.perf_counter()
    while time.perf_counter() - t_r < 0.3:
        b.loglik(gmm, fetch=False); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        b.loglik(gmm, fetch=False)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 10
    fl = 2.0 * 2 * D * S * M * b.N
    print("%s D=%d frames=%d: %.3f ms  %.1f TF of algorithmic flops (%.2f of the fp64 peak)" % (os.environ.get("GMMHMM_LOGLIK", "mfma"), D, b.N, dt * 1e3, fl / dt / 1e12, fl / dt / 78.6e12), flush=True)
    b.close(); gmm.close()
