#!/usr/bin/env python3
"""configs[0] x 1000 (10 x 5 states, 1 Gaussian, D dims, 100 000 utterances = 10 M frames): the fused single-Gaussian
decode (gh_viterbi_fused) against the two-kernel form (gh_loglik + gh_viterbi), both dtypes.
usage: time_fused.py [utterances] [D]      (GMMHMM_FUSED_WAVES=<waves per CU> overrides the grid)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-recognition_amd")):
    sys.path.insert(0, p)
import numpy as np
import bench
from sr.recognition import _hip

U = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 13
W, n, M = 10, 5, 1
wl = bench.synth_workload(1001, U, W=W, n=n, M=M, D=D)
ctx = _hip.Context(0)
S = W * n
gmm = _hip.PackedGMM(ctx, wl["means"].reshape(S, M, D), wl["vars"].reshape(S, M, D), wl["w"].reshape(S, M))
lat = _hip.Lattices(ctx, [bench.stacked_graph(W, n, wl["trans"])])


def timeit(fn, reps=10, ramp=0.3):
    fn(); ctx.sync()
    t_r = time.perf_counter()
    while time.perf_counter() - t_r < ramp:
        fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps


DTS = {'f64': (np.float64,), 'f32': (np.float32,)}.get(os.environ.get('DT', ''), (np.float64, np.float32))
for dt in DTS:
    b = _hip.Batch(ctx, feats=wl["X"], offsets=wl["off"], dtype=dt)
    esz = np.dtype(dt).itemsize
    N = b.N
    t_f = timeit(lambda: lat.viterbi(b, want_path=False, want_end_cost=False, fused_gmm=gmm))
    assert ctx.last_fused
    t_fl = timeit(lambda: lat.viterbi(b, want_path=False, want_end_cost=False, fused_gmm=gmm, log_domain=True))
    r = lat.viterbi(b, want_path=False, want_end_cost=False, fused_gmm=gmm)
    acc = float(np.mean(r["best_end"] == wl["words"]))
    t_ll = timeit(lambda: b.loglik(gmm, fetch=False))
    t_v = timeit(lambda: lat.viterbi(b, want_path=False, want_end_cost=False))
    r2 = lat.viterbi(b, want_path=False, want_end_cost=False)
    by = N * (esz * D + 4.0)
    fl = N * S * D * 4.0
    peak = 78.6e12 if dt == np.float64 else 157.3e12
    print("%s D=%d U=%d N=%d: fused (mahalanobis) %.3f ms, fused (GMM.evaluate) %.3f ms (%.2f TB/s = %.1f %% of HBM; %.1f TF = %.1f %% of the vector peak)  two kernels %.3f + %.3f = %.3f ms  "
          "accuracy %.4f  same words %s" % (np.dtype(dt).name, D, U, N, t_fl * 1e3, t_f * 1e3, by / t_f / 1e12, by / t_f / 8e10, fl / t_f / 1e12,
                                            100 * fl / t_f / peak, t_ll * 1e3, t_v * 1e3, (t_ll + t_v) * 1e3, acc,
                                            bool(np.array_equal(r["best_end"], r2["best_end"]))), flush=True)
    b.close()
