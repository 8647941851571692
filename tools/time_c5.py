#!/usr/bin/env python3
"""C5 decode legs alone (K = 7 lattice and loop grammar) at a given utterance count: wall time of the C-ABI calls;
run under `rocprofv3 --kernel-trace --stats` for the kernel times.  usage: time_c5.py [utterances] [distinct]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-recognition_amd")):
    sys.path.insert(0, p)
import numpy as np
import argparse
import bench
from sr.recognition import _hip
from sr.recognition.continuous_speech import packed_lattice, packed_loop_lattice
U = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ctx = _hip.Context(0)
group = bench.Group(argparse.Namespace(comm="native", backend="nccl"), 0, 1, 0)      # one rank: barrier / max / sum are no-ops
print(json.dumps(bench._continuous_config(ctx, group, U, min(U, int(sys.argv[2]) if len(sys.argv) > 2 else 5000), np.float64), indent=1))
# the decode without a path (end costs only) on a smaller resident batch, K = 7 lattice
K, W, n, M, D = 7, 10, 5, 8, 39
wl = bench.synth_workload(1005, 1, W=W, n=n, M=M, D=D)
rng = np.random.default_rng(7)
Ub = min(U, 20000)
T = rng.integers(210, 421, size=Ub)
off = np.concatenate([[0], np.cumsum(T)]).astype(np.int64)
X = rng.normal(size=(int(off[-1]), D))
gmm = _hip.PackedGMM(ctx, wl["means"].reshape(W * n, M, D), wl["vars"].reshape(W * n, M, D), wl["w"].reshape(W * n, M))
b = _hip.Batch(ctx, feats=X, offsets=off)
b.loglik(gmm, fetch=False)
lat = _hip.Lattices(ctx, [packed_lattice([wl["trans"]] * W, n, [list(range(W))] * K)[0]])
for want_path in (False, True):
    lat.viterbi(b, want_path=want_path)
    t0 = time.perf_counter()
    for _ in range(3):
        lat.viterbi(b, want_path=want_path)
    print("K=7 lattice, %d utterances, %d frames, want_path=%s: %.2f ms wall per call" % (Ub, off[-1], want_path, (time.perf_counter() - t0) / 3 * 1e3))
