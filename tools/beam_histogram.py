#!/usr/bin/env python3
"""Active-row histogram of the C5 lattice decode (K = 7 layers x 10 words x 5 states, 358 rows): how many cells of a
column are alive at all, where the cell of the best path ranks inside its column, and what a rank beam of a given
width does to the result.  Evidence for DESIGN.md's statement on lattice beam pruning at this graph size."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-recognition_amd")):
    sys.path.insert(0, p)
import numpy as np
import bench
from sr.recognition import _hip
from sr.recognition.continuous_speech import packed_lattice
from sr.recognition.batch import path_to_words

U = int(sys.argv[1]) if len(sys.argv) > 1 else 200
K, W, n, M, D = 7, 10, 5, 8, 39
ctx = _hip.Context(0)
rng = np.random.default_rng(1005)
wl = bench.synth_workload(1005, 1, W=W, n=n, M=M, D=D)
means, vars_, trans = wl["means"], wl["vars"], wl["trans"]
words = rng.integers(0, W, size=(U, K))
xs = []
for u in range(U):
    segs = []
    for wd in words[u]:
        t = int(rng.integers(30, 61))
        st = np.minimum(np.arange(t) * n // t, n - 1)
        idx = (wd * n + st) * M + rng.integers(0, M, size=t)
        segs.append(means.reshape(-1, D)[idx] + np.sqrt(vars_).reshape(-1, D)[idx] * rng.standard_normal((t, D)))
    xs.append(np.concatenate(segs))
S = W * n
gmm = _hip.PackedGMM(ctx, means.reshape(S, M, D), vars_.reshape(S, M, D), wl["w"].reshape(S, M))
graph = packed_lattice([trans] * W, n, [list(range(W))] * K)[0]
R = len(graph["row_state"])
b = _hip.Batch(ctx, xs)
b.loglik(gmm, fetch=False)
lat = _hip.Lattices(ctx, [graph])
lat.set_beam(R)                                   # generic kernel, nothing pruned: full cost matrices
full = lat.viterbi(b, want_path=True, want_costs=True)
alive, rank_best, need = [], [], []
for u in range(U):
    c = full["costs"][u]
    T = c.shape[1]
    alive.append(np.isfinite(c).sum(axis=0))
    be = int(np.asarray(graph["end_rows"])[full["best_end"][u]])
    cells = [(be, T - 1)] + [(int(r), int(t)) for r, t in full["paths"][u]]
    rk = [int(np.sum((c[:, t] < c[r, t]) | ((c[:, t] == c[r, t]) & (np.arange(R) < r)))) for r, t in cells]
    rank_best.append(rk)
    need.append(max(rk) + 1)
alive = np.concatenate(alive)
ranks = np.concatenate([np.asarray(r) for r in rank_best])
out = {"lattice_rows": R, "utterances": U, "columns": int(alive.size),
       "alive_cells_per_column": {"mean": float(alive.mean()), "p50": float(np.percentile(alive, 50)), "p90": float(np.percentile(alive, 90)),
                                  "max": int(alive.max()), "fraction_of_rows": float(alive.mean() / R)},
       "rank_of_best_path_cell": {"p50": float(np.percentile(ranks, 50)), "p90": float(np.percentile(ranks, 90)),
                                  "p99": float(np.percentile(ranks, 99)), "max": int(ranks.max())},
       "beam_needed_to_keep_the_best_path": {"p50": float(np.percentile(need, 50)), "p90": float(np.percentile(need, 90)),
                                             "max": int(np.max(need))},
       "beams": {}}
truth = [list(map(int, w)) for w in words]
for beam in (8, 16, 32, 64, 128, 256):
    lat.set_beam(beam)
    r = lat.viterbi(b, want_path=True)
    same = np.mean([np.array_equal(r["paths"][u], full["paths"][u]) for u in range(U)])
    acc = np.mean([path_to_words(r["paths"][u], graph["row_state"], n) == truth[u] if len(r["paths"][u]) else False for u in range(U)])
    out["beams"][str(beam)] = {"same_path_as_unpruned": float(same), "sequence_accuracy": float(acc)}
print(json.dumps(out, indent=1))
