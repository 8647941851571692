import os, sys, time, json
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "speech-recognition_amd")
import bench
from sr.recognition import _hip
from sr.recognition.continuous_speech import packed_lattice, packed_loop_lattice
ctx = _hip.default_context()
U, K, W, n, M, D = 2000, 7, 10, 5, 8, 39
rng = np.random.default_rng(1005)
wl = bench.synth_workload(1005, 1, W=W, n=n, M=M, D=D)
S = W * n
T = rng.integers(30, 61, size=(U, K)).sum(axis=1)
off = np.concatenate([[0], np.cumsum(T)]).astype(np.int64)
X = rng.normal(size=(int(off[-1]), D))
gmm = _hip.PackedGMM(ctx, wl["means"].reshape(S, M, D), wl["vars"].reshape(S, M, D), wl["w"].reshape(S, M))
b = _hip.Batch(ctx, feats=X, offsets=off)
b.loglik(gmm, fetch=False)
def t(fn, reps=5):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3
for name, g in (("loop", packed_loop_lattice([wl["trans"]] * W, n)[0]), ("K7", packed_lattice([wl["trans"]] * W, n, [list(range(W))] * K)[0])):
    lat = _hip.Lattices(ctx, [g])
    print(name, "no path %.2f ms | path %.2f ms" % (t(lambda: lat.viterbi(b, want_path=False)), t(lambda: lat.viterbi(b, want_path=True))), flush=True)
