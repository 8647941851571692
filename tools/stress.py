# -*- coding: utf-8 -*-
"""Handle-lifetime / shape stress: many small random models and batches through the C ABI, spot-checked against
the oracle; free device memory is compared before and after (leak check)."""
import ctypes as C
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "speech-recognition_amd"))
from oracle import ref_numpy as O
from sr.recognition import _hip

hip = C.CDLL("libamdhip64.so")
def free_mem():
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value

ctx = _hip.default_context(0)
rng = np.random.default_rng(123)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
def one(k):
    S, M, D = int(rng.integers(1, 70)), int(rng.choice([1, 2, 3, 4, 5, 8, 9, 16, 17, 33])), int(rng.integers(1, 45))
    U = int(rng.integers(1, 9))
    lens = rng.integers(1, 60, size=U)
    means = rng.normal(size=(S, M, D)); vars_ = rng.uniform(0.3, 2.0, size=(S, M, D)); w = rng.dirichlet(np.ones(M), size=S)
    xs = [rng.normal(size=(int(t), D)) for t in lens]
    dt = np.float64 if k % 3 else np.float32
    gmm = _hip.PackedGMM(ctx, means, vars_, w)
    b = _hip.Batch(ctx, xs, dtype=dt)
    nll = b.loglik(gmm)
    ref = O.gmm_neg_loglik_batch(np.concatenate(xs), means, vars_, w)
    np.testing.assert_allclose(nll, ref, rtol=1e-10 if dt == np.float64 else 2e-3)
    n = min(S, int(rng.integers(1, 7)))
    trans = np.full((n, n), np.inf)
    for i in range(n):
        trans[i, i] = 0.3
        if i + 1 < n: trans[i + 1, i] = 1.2
    to, fr = np.nonzero(~np.isinf(trans))
    g = dict(row_state=np.arange(n), arc_to=to, arc_from=fr, arc_cost=trans[to, fr], start_rows=[0], end_rows=[n - 1])
    lat = _hip.Lattices(ctx, [g])
    r = lat.viterbi(b, want_path=True, want_costs=(k % 5 == 0))
    fb = lat.forward_backward(b, want_occ=(k % 2 == 0))
    for u in range(U):
        E = nll[b.offsets[u]:b.offsets[u + 1], :n].T.astype(np.float64)
        if lens[u] > 1:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                costs, path = O.decode_states(E, np.zeros(n, dtype=bool), trans)
            if np.isfinite(costs[-1, -1]):
                np.testing.assert_allclose(r["end_cost"][u][0], costs[-1, -1], rtol=1e-9 if dt == np.float64 else 1e-4)
                np.testing.assert_array_equal(r["paths"][u], path.reshape(-1, 2))
    lat.close(); b.close(); gmm.close()

def one_seq(k):
    """Round-2 paths: a transcripts handle (sequence form; now and then a case that needs the expanded twin), forced
    alignment + regrouping, forward-backward with occupancies, fused statistics, lock-step k-means with resident
    assignments and device gathers; one utterance per call against the oracle's reference-shaped DP."""
    from sr.recognition.continuous_speech import packed_lattice, transcript_state_sets
    W, n = int(rng.integers(2, 7)), int(rng.integers(2, 7))
    M, D = int(rng.choice([1, 2, 4, 8])), int(rng.integers(2, 20))
    S = W * n
    means = rng.normal(size=(S, M, D)) * 2; vars_ = rng.uniform(0.5, 1.5, size=(S, M, D)); w = rng.dirichlet(np.ones(M), size=S)
    wt = []
    for _ in range(W):
        t = np.full((n, n), np.inf)
        for i in range(n):
            t[i, i] = rng.uniform(0.05, 0.5)
            if i + 1 < n: t[i + 1, i] = rng.uniform(0.8, 2.0)
            if i + 2 < n and rng.random() < 0.3: t[i + 2, i] = rng.uniform(1.5, 3.0)
        wt.append(t)
    U = int(rng.integers(1, 9))
    seqs = [[int(v) for v in rng.integers(0, W, size=int(rng.integers(1, 6)))] for _ in range(U)]
    xs = []
    for l in seqs:
        segs = []
        for wd in l:
            Tw = int(rng.integers(n, 3 * n + 2))
            st = np.minimum(np.arange(Tw) * n // Tw, n - 1)
            segs.append(means[wd * n + st, 0] + rng.normal(size=(Tw, D)))
        xs.append(np.concatenate(segs))
    if k % 7 == 0:
        xs[0] = xs[0][:1]                                    # T == 1: the expanded twin takes over
        seqs[0] = seqs[0][:1]
    gmm = _hip.PackedGMM(ctx, means, vars_, w)
    b = _hip.Batch(ctx, xs)
    nll = b.loglik(gmm, fetch=True, state_sets=transcript_state_sets(seqs, n, W))
    lat = _hip.Lattices.from_transcripts(ctx, wt, n, seqs)
    ug = np.arange(U, dtype=np.int32)
    r = lat.viterbi(b, utt_lattice=ug, want_path=True)
    seg = lat.align_segments(b, utt_lattice=ug)
    fb = lat.forward_backward(b, utt_lattice=ug, want_occ=True, want_self_xi=True)
    stats = b.bw_accumulate(gmm)
    assert abs(stats[:, :, 0].sum() - fb["occ"].sum()) <= 1e-9 * max(1.0, fb["occ"].sum())
    u = int(rng.integers(0, U))
    if len(xs[u]) > 1:
        g = packed_lattice(wt, n, [[l] for l in seqs[u]])[0]
        R = len(g["row_state"])
        dense = np.full((R, R), np.inf); dense[g["arc_to"], g["arc_from"]] = g["arc_cost"]
        is_nes = g["row_state"] < 0
        E = np.zeros((R, len(xs[u])))
        E[~is_nes] = nll[b.offsets[u]:b.offsets[u + 1]][:, g["row_state"][~is_nes]].T
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            costs, path = O.decode_states(E, is_nes, dense, end_points=[[int(e), -1] for e in g["end_rows"]])
        np.testing.assert_allclose(r["end_cost"][u], costs[np.asarray(g["end_rows"]), -1], rtol=1e-10)
        np.testing.assert_array_equal(r["paths"][u], path)
        want = np.full(len(xs[u]), -1)
        for row, lo, hi in O.cut_segments(path, is_nes):
            want[lo:hi] = g["row_state"][row]
        np.testing.assert_array_equal(seg["frame_state"][b.offsets[u]:b.offsets[u + 1]], want)
    # lock-step k-means on the frames regrouped per state (device gather, resident assignments)
    fs = seg["frame_state"]
    used = np.flatnonzero(fs >= 0)
    if len(used) > 4:
        order = used[np.argsort(fs[used], kind="stable")]
        sid, cnt = np.unique(fs[order], return_counts=True)
        off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        gb = b.gather(order)
        kk = 2
        cent = rng.normal(size=(len(sid), kk, D))
        gb.resident_clusters(reset=True, fetch=False)
        _, changed, sums = gb.kmeans_assign_multi(off, cent, clusters=_hip.RESIDENT, want_sums=True)
        cl = gb.resident_clusters()
        X = np.concatenate(xs)[order]
        for j in range(len(sid)):
            for c in range(kk):
                m = cl[off[j]:off[j + 1]] == c
                assert sums[j, c, D] == m.sum()
                if m.any() and D >= 2:
                    np.testing.assert_array_equal(sums[j, c, :D] / sums[j, c, D], np.mean(X[off[j]:off[j + 1]][m], axis=0))
        gb.close()
    lat.close(); b.close(); gmm.close()


one(0)
one_seq(0)
ctx.sync()
f0 = free_mem()
for k in range(1, iters):
    one(k)
    one_seq(k)
ctx.sync()
f1 = free_mem()
print("stress ok: %d iterations; free device memory before %.1f MB, after %.1f MB (delta %.1f MB)" % (iters, f0 / 1e6, f1 / 1e6, (f0 - f1) / 1e6))
