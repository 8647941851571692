# -*- coding: utf-8 -*-
"""Drive the HOST code of libgmmhmm with random shapes (run by tests/test_host_sanitized.py in a python whose process has
the ASan runtime preloaded and GMMHMM_LIB pointing at the sanitized, HIP-stubbed library): plan builders, chunk planners,
upload layouts, transcript expansion, session set-up.  Kernels do nothing there and "device" memory reads back as zeros, so
results mean nothing and library errors (GH_ERR_*) are tolerated; what counts is that no call touches memory it does not own
(ASan) and does no undefined arithmetic (UBSan) -- either ends the process.

    python tests/hipstub/drive.py N_CASES SEED"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "speech-recognition_amd"))
sys.path.insert(0, ROOT)
import numpy as np

from sr.recognition import _hip, _pack

n_cases, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
ctx = _hip.default_context()
done = {"cases": 0, "errors": 0}


def lengths(U, allow_short=True):
    tt = rng.integers(2 if not allow_short else 0, 60, size=U)
    if allow_short and U:
        tt[rng.integers(0, U)] = rng.choice([0, 1, 1, 2])
    return [int(x) for x in tt]


def ltr_trans(n, skip=False):
    tr = np.full((n, n), np.inf)
    for i in range(n):
        tr[i, i] = rng.uniform(0.05, 1.0)
        if i + 1 < n:
            tr[i + 1, i] = rng.uniform(0.5, 3.0)
        if skip and i + 2 < n:
            tr[i + 2, i] = rng.uniform(1.0, 4.0)
    return tr


TOLERATED = (_hip.BackendError, NameError, AssertionError, IndexError, ValueError, np.linalg.LinAlgError, ZeroDivisionError, FloatingPointError)


def t(fn):
    """One step of a case: a tolerated error (refused shape; a zero 'device' result that makes no sense to the wrapper)
    ends the step, not the case."""
    try:
        return fn()
    except TOLERATED as e:
        done["step_errors"] = done.get("step_errors", 0) + 1
        if os.environ.get("DRIVE_VERBOSE"):
            print("    step: %s: %s" % (type(e).__name__, str(e)[:140]), flush=True)
        return None


def attempt(fn):
    done["cases"] += 1
    try:
        fn()
    except TOLERATED as e:
        done["errors"] += 1          # (a refused shape or a meaningless zero result: fine -- memory errors do not come back as exceptions)
        if os.environ.get("DRIVE_VERBOSE"):
            print("  %s: %s: %s" % (fn.__name__, type(e).__name__, str(e)[:160]), flush=True)


def case_batches():
    D = int(rng.choice([1, 2, 13, 39, 40, 64]))
    U = int(rng.integers(0, 9))
    xs = [rng.normal(size=(tt, D)) for tt in lengths(U)]
    dt = rng.choice([np.float64, np.float32])
    b = _hip.Batch(ctx, xs, dtype=dt) if U else _hip.Batch(ctx, feats=np.zeros((0, D)), offsets=[0], dtype=dt)
    b.features()
    if b.N:
        rows = rng.integers(0, b.N, size=int(rng.integers(0, 2 * b.N + 1)))
        g = b.gather(rows)
        g.close()
        starts = rng.integers(0, b.N, size=4)
        lens = np.minimum(rng.integers(0, 7, size=4), b.N - starts)
        g = b.gather_runs(starts.astype(np.int64), lens.astype(np.int64), (np.cumsum(lens) - lens).astype(np.int64), int(lens.sum()))
        g.close()
    if rng.random() < 0.5 and b.N:
        w = _hip.Batch(ctx, feats=np.concatenate(xs).astype(np.float32), offsets=b.offsets, dtype=np.float64, wire=np.float32, pin=bool(rng.random() < 0.5))
        w.close()
    b.close()


def case_frontend():
    U = int(rng.integers(1, 5))
    ceps = [rng.normal(size=(int(rng.integers(2, 40)), 13)) for _ in range(U)]
    b = _hip.Batch(ctx, cepstra=ceps, frontend_mode=int(rng.choice([0, 1, 2])), dtype=rng.choice([np.float64, np.float32]))
    b.close()
    sig = [rng.normal(size=int(rng.integers(400, 6000))).astype(rng.choice([np.float32, np.float64])) for _ in range(U)]
    if rng.random() < 0.5:
        sig = [(s * 3000).astype(np.int16) for s in sig]
    _hip.mfcc(ctx, sig, 16000)
    b = _hip.Batch(ctx, pcm=sig, sample_rate=16000, frontend_mode=int(rng.choice([0, 2])))
    b.close()


def model(S, M, D):
    return _hip.PackedGMM(ctx, rng.normal(size=(S, M, D)), rng.uniform(0.5, 2, size=(S, M, D)), rng.dirichlet(np.ones(M), size=S))


def case_loglik_viterbi():
    W, n = int(rng.integers(1, 7)), int(rng.choice([2, 3, 5, 8, 12, 16]))
    M, D = int(rng.choice([1, 2, 4, 8, 9, 32])), int(rng.choice([2, 13, 39, 47, 64]))
    S = W * n
    gmm = model(S, M, D)
    U = int(rng.integers(1, 12))
    xs = [rng.normal(size=(tt, D)) for tt in lengths(U, allow_short=False)]
    b = _hip.Batch(ctx, xs, dtype=rng.choice([np.float64, np.float32]))
    words = rng.integers(0, W, size=U).astype(np.int32)
    kind = rng.integers(0, 3)
    if kind == 0:
        b.loglik(gmm, fetch=bool(rng.random() < 0.5))
    elif kind == 1:
        b.loglik(gmm, fetch=False, state_ranges=(words * n, words * n + n))
    else:
        off = np.arange(U + 1, dtype=np.int64) * 2
        lo = np.repeat(words * n, 2).astype(np.int32)
        b.loglik(gmm, fetch=False, state_sets=(off, lo, lo + n))
    trans = [ltr_trans(n, skip=bool(rng.random() < 0.3)) for _ in range(W)]
    # chains per word; all words stacked in one graph; K-word transcripts
    graphs = [_pack.graph_from_dense(np.arange(n) + w * n, trans[w], [0], [n - 1]) for w in range(W)]
    lat = _hip.Lattices(ctx, graphs)
    if kind == 0:
        t(lambda: lat.viterbi(b, utt_lattice=words, want_path=bool(rng.random() < 0.7), want_costs=bool(rng.random() < 0.3)))
        t(lambda: lat.forward_backward(b, utt_lattice=words, want_occ=bool(rng.random() < 0.5), want_self_xi=bool(rng.random() < 0.5)))
    else:
        t(lambda: lat.viterbi(b, utt_lattice=words, want_path=True))
    lat.close()
    if kind == 0:
        big = np.full((S, S), np.inf)
        for w in range(W):
            big[w * n:(w + 1) * n, w * n:(w + 1) * n] = trans[w]
        st = _hip.Lattices(ctx, [_pack.graph_from_dense(np.arange(S), big, np.arange(W) * n, np.arange(W) * n + n - 1)])
        t(lambda: st.viterbi(b, want_path=False, want_end_cost=bool(rng.random() < 0.5)))
        t(lambda: st.viterbi(b, want_path=True))
        if M == 1:
            t(lambda: st.viterbi(b, want_path=False, fused_gmm=gmm, log_domain=bool(rng.random() < 0.5)))
        t(lambda: st.viterbi_labels(b, np.repeat(np.arange(W), n).astype(np.int32)))
        st.close()
    K = int(rng.integers(1, 8))
    seqs = [[int(x) for x in rng.integers(0, W, size=int(rng.integers(1, K + 1)))] for _ in range(int(rng.integers(1, 5)))]
    fa = _hip.Lattices.from_transcripts(ctx, trans, n, seqs)
    ul = rng.integers(0, len(seqs), size=U).astype(np.int32)
    if kind != 0:
        b.loglik(gmm, fetch=False)
    t(lambda: fa.viterbi(b, utt_lattice=ul, want_path=True))
    t(lambda: fa.align_segments(b, utt_lattice=ul))
    t(lambda: fa.align_runs(b, utt_lattice=ul))
    t(lambda: fa.forward_backward(b, utt_lattice=ul, want_occ=True, want_self_xi=True))
    fa.close()
    b.close()
    gmm.close()


def case_em_sessions():
    strings = bool(rng.random() < 0.5)
    W, n = int(rng.integers(1, 6)), int(rng.choice([3, 5, 8] if strings else [3, 5, 8, 9, 16]))
    M, D = int(rng.choice([1, 4, 8, 12, 32, 33])), int(rng.choice([5, 13, 23, 39, 44]))
    S = W * n
    U = int(rng.integers(1, 14))
    trans = np.array([ltr_trans(n, skip=bool(rng.random() < 0.3)) for _ in range(W)])
    means, vars_, w = rng.normal(size=(S, M, D)), rng.uniform(0.5, 2, size=(S, M, D)), rng.dirichlet(np.ones(M), size=S)
    if not strings:
        xs = [rng.normal(size=(tt, D)) for tt in lengths(U, allow_short=False)]
        b = _hip.Batch(ctx, xs)
        s = _hip.EMSession(ctx, b, means, vars_, w, trans, rng.integers(0, W, size=U), 1e-6, occ_floor=float(rng.choice([0.0, 1e-3])))
    else:
        K = int(rng.integers(1, 8))
        ts = [[int(x) for x in rng.integers(0, W, size=int(rng.integers(1, K + 1)))] for _ in range(int(rng.integers(1, 6)))]
        xs = [rng.normal(size=(int(rng.integers(2 * K + 2, 12 * K + 4)), D)) for _ in range(U)]
        b = _hip.Batch(ctx, xs)
        s = _hip.EMSession(ctx, b, means, vars_, w, trans, rng.integers(0, len(ts), size=U), 1e-6, transcripts=ts,
                           update_transitions=bool(rng.random() < 0.5))
    for _ in range(2):
        t(lambda: s.iteration(sync=bool(rng.random() < 0.5)))
    t(s.history)
    t(s.model)
    t(s.packed)
    s.close()
    b.close()


def case_fit_session():
    D, k = int(rng.choice([2, 3, 13, 16, 39, 40, 64])), int(rng.choice([1, 2, 3, 4, 8, 16]))
    S = int(rng.integers(1, 7))
    lens = rng.integers(0, 1300, size=S)
    if rng.random() < 0.3:
        lens[rng.integers(0, S)] = rng.choice([0, 1, 15, 16, 17, 512, 513])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    N = int(off[-1])
    b = _hip.Batch(ctx, feats=rng.normal(size=(N, D)), offsets=[0, N])
    if rng.random() < 0.3:
        os.environ["GMMHMM_REFIT"] = "tiles"
    try:
        fit = _hip.FitSession(ctx, b, off, int(max(k, rng.choice([2, 8]))))
    finally:
        os.environ.pop("GMMHMM_REFIT", None)
    t(lambda: fit.segment_means())
    c0 = rng.normal(size=(S, k, D))
    fit.kmeans(k, c0, rng.integers(0, k, size=N).astype(np.uint8), max_iteration=int(rng.integers(1, 20)))
    t(lambda: fit.clusters())
    mean, var, ww = rng.normal(size=(S, k, D)), rng.uniform(0.5, 2, size=(S, k, D)), rng.dirichlet(np.ones(k), size=S)
    fit.em(k, mean, var, ww, np.zeros_like(mean), np.ones_like(mean), np.zeros_like(ww), np.maximum(lens, 1).astype(np.float64),
           max_iteration=int(rng.integers(1, 20)))
    fit.set_ids(rng.integers(0, k, size=N).astype(np.int32))
    t(lambda: fit.group_stats(k, None if rng.random() < 0.5 else (rng.random(S) < 0.6).astype(np.uint8)))
    fit.close()
    # the call-by-call forms
    b.kmeans_assign_multi(off, c0, var=rng.uniform(0.5, 2, size=(S, D)), want_sums=True)
    if k <= 32:
        b.em_accumulate_multi(off, mean, var, ww)
    b.close()


def case_skmeans_session():
    W, n, D = int(rng.integers(1, 5)), int(rng.choice([2, 3, 5, 8])), int(rng.choice([2, 13, 39]))
    n_temps = rng.integers(1, 6, size=W)
    xs = [rng.normal(size=(int(rng.integers(max(5, n), 40)), D)) for _ in range(int(n_temps.sum()))]
    b = _hip.Batch(ctx, xs)
    word_off = b.offsets[np.concatenate([[0], np.cumsum(n_temps)])]
    fit = _hip.FitSession(ctx, b, word_off, max(n, 2))
    fit.set_ids(rng.integers(0, n, size=b.N).astype(np.int32))
    t(lambda: fit.group_stats(n))
    fit.dtw(n, rng.normal(size=(W, n, D)), np.array([ltr_trans(n) for _ in range(W)]), np.repeat(np.arange(W), n_temps).astype(np.int32),
            None if rng.random() < 0.5 else np.ones(W, dtype=np.uint8))
    t(lambda: fit.clusters())
    fit.close()
    b.dtw(ltr_trans(n), y=rng.normal(size=(n, D)), var=None if rng.random() < 0.5 else rng.uniform(0.5, 2, size=(n, D)),
          beam=float(rng.choice([0, 0, 30.0])))
    b.close()


def case_decode_lattices():
    """The continuous decoder's graphs: K-layer lattice and word-loop grammar over 1 .. 64 words of 2 .. 16 states -- the form
    recognition of gh_lattices_create (narrow / four-set / wide layer form, loop and wide loop form, or none), the decision-word
    planning and the launch arguments of gh_viterbi / gh_viterbi_labels."""
    from sr.recognition.continuous_speech import packed_lattice, packed_loop_lattice
    W = int(rng.choice([1, 3, 10, 16, 17, 40, 64]))
    n = int(rng.choice([2, 5, 8, 12, 16]))
    K = int(rng.integers(1, 17))
    if K * (W * n + 1) > 6000:
        K = max(1, 6000 // (W * n + 1))
    M, D = int(rng.choice([1, 4])), int(rng.choice([6, 13]))
    gmm = model(W * n, M, D)
    U = int(rng.integers(1, 10))
    xs = [rng.normal(size=(tt, D)) for tt in lengths(U, allow_short=False)]
    b = _hip.Batch(ctx, xs, dtype=rng.choice([np.float64, np.float32]))
    b.loglik(gmm, fetch=False)
    trans = [ltr_trans(n, skip=bool(rng.random() < 0.3)) for _ in range(W)]
    for graph in (packed_lattice(trans, n, [list(range(W))] * K)[0], packed_loop_lattice(trans, n, float(rng.choice([0.0, 1.5])))[0]):
        lat = _hip.Lattices(ctx, [graph])
        t(lat.forms)
        t(lambda: lat.viterbi(b, want_path=bool(rng.random() < 0.7)))
        row_word = np.where(graph["row_state"] >= 0, graph["row_state"] // n, -1).astype(np.int32)
        t(lambda: lat.viterbi_labels(b, row_word))
        t(lambda: lat.viterbi_labels(b, row_word, as_lists=False))
        lat.close()
    b.close()
    gmm.close()


CASES = [(case_batches, 3), (case_frontend, 1), (case_loglik_viterbi, 4), (case_em_sessions, 3), (case_fit_session, 3), (case_skmeans_session, 2),
         (case_decode_lattices, 3)]
pool = [f for f, wgt in CASES for _ in range(wgt)]
# a small scratch budget part of the time: the chunk planners cut the launches
for i in range(n_cases):
    if i % 3 == 1:
        os.environ["GMMHMM_SCRATCH_BUDGET"] = "8M"
    else:
        os.environ.pop("GMMHMM_SCRATCH_BUDGET", None)
    attempt(pool[int(rng.integers(0, len(pool)))])
import ctypes
ctx.lib.hipstub_launches.restype = ctypes.c_long
print("driven %d cases (%d ended in a library / shape error, %d steps did), %d kernel launches accepted by the stub" %
      (done["cases"], done["errors"], done.get("step_errors", 0), ctx.lib.hipstub_launches()))
