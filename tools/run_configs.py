#!/usr/bin/env python3
# -*- coding: utf-8 -*-
"""Run the other BASELINE.json configs (parity-test cases, not bench lines) at full or
reduced size on one GPU and print timings + size-independent checks:
  C1  10 HMM x 5 st, M=1, D=13, 100 utts        (HBM-bound likelihood)
  C4  64 HMM x 16 st x M=32, D=39               (LDS-chunked output, 1024 states)
  C5  continuous decode, K=7 layers x 10 words, R=358 lattice rows
usage: python tools/run_configs.py [c1] [c4] [c5] [--utts N]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-recognition_amd")):
    sys.path.insert(0, p)
import bench  # noqa: E402
from sr.recognition import _hip  # noqa: E402
from sr.recognition.continuous_speech import packed_lattice, packed_loop_lattice  # noqa: E402
from sr.recognition.batch import path_to_words  # noqa: E402


def timeit(fn, reps=5, ramp=0.4):
    """Mean wall time of fn() over `reps` calls, after keeping the GPU busy with it for `ramp` seconds (a cold
    MI355X runs ~20 % below its sustained clock for the first few hundred ms)."""
    t_r = time.perf_counter()
    fn()
    while time.perf_counter() - t_r < ramp:
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t0) / reps, out


def isolated(name, seed, U, W, n, M, D, dtype=np.float64):
    ctx = _hip.default_context()
    wl = bench.synth_workload(seed, U, W=W, n=n, M=M, D=D)
    S = W * n
    gmm = _hip.PackedGMM(ctx, wl["means"].reshape(S, M, D), wl["vars"].reshape(S, M, D), wl["w"].reshape(S, M))
    b = _hip.Batch(ctx, feats=wl["X"], offsets=wl["off"], dtype=dtype)
    lat = _hip.Lattices(ctx, [bench.stacked_graph(W, n, wl["trans"])])
    t_ll, _ = timeit(lambda: (b.loglik(gmm, fetch=False), ctx.sync()))
    t_vit, r = timeit(lambda: lat.viterbi(b, want_path=False))
    words = np.argmin(r["end_cost_flat"].reshape(U, W), axis=1)
    esz = np.dtype(dtype).itemsize
    print(json.dumps(dict(config=name, utts=U, frames=int(b.N), states=S, mix=M, dim=D, dtype=str(np.dtype(dtype)),
                          loglik_ms=t_ll * 1e3, viterbi_ms=t_vit * 1e3,
                          frame_states_per_s=b.N * S / (t_ll + t_vit),
                          loglik_TFLOPs=2.0 * 2 * D * S * M * b.N / t_ll / 1e12,
                          loglik_hbm_GBps=esz * (D + S) * b.N / t_ll / 1e9,
                          decode_accuracy=float(np.mean(words == wl["words"])))), flush=True)


def continuous(U, K=7, W=10, n=5, M=8, D=39):
    ctx = _hip.default_context()
    rng = np.random.default_rng(1005)
    wl = bench.synth_workload(1005, 1, W=W, n=n, M=M, D=D)
    means, vars_, trans = wl["means"], wl["vars"], wl["trans"]
    S = W * n
    # U utterances of K words each, 30..60 frames per word
    words = rng.integers(0, W, size=(U, K))
    Tw = rng.integers(30, 61, size=(U, K))
    T = Tw.sum(axis=1)
    off = np.concatenate([[0], np.cumsum(T)]).astype(np.int64)
    N = int(off[-1])
    X = np.empty((N, D))
    pos = 0
    mflat, sflat = means.reshape(-1, D), np.sqrt(vars_).reshape(-1, D)
    for u in range(U):
        for k in range(K):
            t = Tw[u, k]
            st = np.minimum(np.arange(t) * n // t, n - 1)
            idx = (words[u, k] * n + st) * M + rng.integers(0, M, size=t)
            X[pos:pos + t] = mflat[idx] + sflat[idx] * rng.standard_normal((t, D))
            pos += t
    gmm = _hip.PackedGMM(ctx, means.reshape(S, M, D), vars_.reshape(S, M, D), wl["w"].reshape(S, M))
    graph, _ = packed_lattice([trans] * W, n, [list(range(W))] * K)
    lat = _hip.Lattices(ctx, [graph])
    b = _hip.Batch(ctx, feats=X, offsets=off)
    t_ll, _ = timeit(lambda: (b.loglik(gmm, fetch=False), ctx.sync()), reps=3)
    t_vit, r = timeit(lambda: lat.viterbi(b, want_path=True), reps=3)
    dec = [path_to_words(p, graph["row_state"], n) for p in r["paths"]]
    acc = float(np.mean([d == list(wd) for d, wd in zip(dec, words)]))
    R = len(graph["row_state"])
    print(json.dumps(dict(config="C5", utts=U, frames=N, lattice_rows=R, layers=K, loglik_ms=t_ll * 1e3,
                          viterbi_ms=t_vit * 1e3, utterances_per_s=U / (t_ll + t_vit),
                          dp_cells_per_s=N * R / t_vit, sequence_accuracy=acc)), flush=True)
    # the same utterances through the word-loop grammar (N4): 2 + W*n rows, any number of words
    lgraph, _ = packed_loop_lattice([trans] * W, n)
    llat = _hip.Lattices(ctx, [lgraph])
    t_loop, rl = timeit(lambda: llat.viterbi(b, want_path=True), reps=3)
    ldec = [path_to_words(p, lgraph["row_state"], n) for p in rl["paths"]]
    lacc = float(np.mean([d == list(wd) for d, wd in zip(ldec, words)]))
    row_word = np.where(lgraph["row_state"] >= 0, lgraph["row_state"] // n, -1).astype(np.int32)
    t_lab, rlab = timeit(lambda: llat.viterbi_labels(b, row_word, max_labels=b.lengths // (n - 1) + 2), reps=3)
    lab_same = all([int(v) for v in a] == d for a, d in zip(rlab["labels"], ldec))
    row_word_k = np.where(graph["row_state"] >= 0, graph["row_state"] // n, -1).astype(np.int32)
    t_labk, rlabk = timeit(lambda: lat.viterbi_labels(b, row_word_k, max_labels=K + 1), reps=3)
    _best = lambda res: np.minimum.reduceat(res["end_cost_flat"], res["end_off"][:-1].astype(np.int64))
    same = float(np.mean(_best(rl) <= _best(r)))
    print(json.dumps(dict(config="C5 loop grammar", utts=U, frames=N, lattice_rows=len(lgraph["row_state"]),
                          viterbi_ms=t_loop * 1e3, utterances_per_s=U / (t_ll + t_loop),
                          dp_cells_per_s=N * len(lgraph["row_state"]) / t_loop, sequence_accuracy=lacc,
                          loop_cost_le_K_layer_cost=same, labels_only_ms=t_lab * 1e3, labels_equal_path_to_words=lab_same,
                          labels_only_utterances_per_s=U / (t_ll + t_lab), K_layer_labels_only_ms=t_labk * 1e3)), flush=True)


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["c1", "c4", "c5"]
    utts = int(sys.argv[sys.argv.index("--utts") + 1]) if "--utts" in sys.argv else None
    if "c1" in which:
        isolated("C1", 1001, utts or 100, 10, 5, 1, 13)
        isolated("C1x1000 (100k utts)", 1001, 100000, 10, 5, 1, 13)
        isolated("C1x1000 fp32", 1001, 100000, 10, 5, 1, 13, dtype=np.float32)
    if "c4" in which:
        isolated("C4 (reduced utts)", 1004, utts or 2000, 64, 16, 32, 39)
        isolated("C4 fp32 (reduced utts)", 1004, utts or 2000, 64, 16, 32, 39, dtype=np.float32)
    if "c5" in which:
        continuous(utts or 2000)
