#!/usr/bin/env python3
"""The generic (any-graph) Viterbi kernel on a lattice of more than 512 rows -- K = 7 layers of 16 words x 5 states:
568 rows, several passes of the 512-thread workgroup per level -- next to the lean and the layer-form kernels on the
same batch (GMMHMM_VITERBI is read per call).  Run under rocprofv3 --kernel-trace --stats for the kernel times.
usage: time_generic.py [utterances]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-recognition_amd")):
    sys.path.insert(0, p)
import numpy as np
import bench
from sr.recognition import _hip
from sr.recognition.continuous_speech import packed_lattice

U = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
K, W, n, M, D = 7, 16, 5, 8, 39
wl = bench.synth_workload(1005, 1, W=W, n=n, M=M, D=D)
rng = np.random.default_rng(7)
T = rng.integers(210, 421, size=U)
off = np.concatenate([[0], np.cumsum(T)]).astype(np.int64)
X = rng.normal(size=(int(off[-1]), D))
ctx = _hip.Context(0)
gmm = _hip.PackedGMM(ctx, wl["means"].reshape(W * n, M, D), wl["vars"].reshape(W * n, M, D), wl["w"].reshape(W * n, M))
b = _hip.Batch(ctx, feats=X, offsets=off)
b.loglik(gmm, fetch=False)
graph = packed_lattice([wl["trans"]] * W, n, [list(range(W))] * K)[0]
lat = _hip.Lattices(ctx, [graph])
print("%d utterances, %d frames, lattice of %d rows / %d arcs, forms %s" % (U, off[-1], len(graph["row_state"]), len(graph["arc_to"]),
                                                                           sorted(lat.forms())))
ref = None
for name in ("generic", "lean", ""):
    if name:
        os.environ["GMMHMM_VITERBI"] = name
    else:
        os.environ.pop("GMMHMM_VITERBI", None)
    r = lat.viterbi(b, want_path=True)
    t0 = time.perf_counter()
    for _ in range(2):
        r = lat.viterbi(b, want_path=True)
    dt = (time.perf_counter() - t0) / 2
    if ref is None:
        ref = r
    same = np.array_equal(r["end_cost_flat"], ref["end_cost_flat"]) and all(np.array_equal(p, q) for p, q in zip(r["paths"], ref["paths"]))
    print("%-8s %.1f ms per call (wall, with the path copy-back), identical to generic: %s" % (name or "layers", dt * 1e3, same))
