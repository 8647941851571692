# -*- coding: utf-8 -*-
"""BASELINE.json configs[2] and configs[3] at their own shapes (`configs_untested`), plus the fp32
likelihood kernel off standardised data.

configs[3] (C4): 64 HMMs x 16 states x 32 mixtures, 39-dim -- S = 1024 states, the `MP == 32` epilogue of the MFMA
kernel together with its mid-range LDS flushes (chunked [32, S] output tile).  >= 100 k frames: sampled rows against
the oracle, batching independence (bitwise), fp32 against fp64, decode accuracy over the 1024-row stacked graph,
gh_loglik_subset on it.
configs[2] (C3): the full EM loop on the 10 x 5 x 8-mix x 39-dim model with 2 000 utterances: statistics of sampled
states against numpy, monotone likelihood, two ranks == one rank.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import ref_numpy as O

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------ configs[3]
@pytest.fixture(scope="module")
def c4():
    import bench
    from sr.recognition import _hip
    ctx = _hip.default_context()
    wl = bench.synth_workload(1004, 1040, W=64, n=16, M=32, D=39)      # > 100 k frames
    W, n, M, D = wl["W"], wl["n"], wl["M"], wl["D"]
    S = W * n
    fm, fv, fw = wl["means"].reshape(S, M, D), wl["vars"].reshape(S, M, D), wl["w"].reshape(S, M)
    gmm = _hip.PackedGMM(ctx, fm, fv, fw)
    batch = _hip.Batch(ctx, feats=wl["X"], offsets=wl["off"])
    nll = batch.loglik(gmm, fetch=True)
    yield dict(hip=_hip, ctx=ctx, wl=wl, gmm=gmm, batch=batch, nll=nll, S=S, fm=fm, fv=fv, fw=fw,
               graph=bench.stacked_graph(W, n, wl["trans"]))
    batch.close()
    gmm.close()


def test_c4_shape_is_what_baseline_names(c4):
    wl, nll = c4["wl"], c4["nll"]
    assert (wl["W"], wl["n"], wl["M"], wl["D"]) == (64, 16, 32, 39) and c4["S"] == 1024
    assert nll.shape[0] >= 100000 and nll.shape[1] == 1024 and np.all(np.isfinite(nll))


def test_c4_loglik_sampled_rows_vs_oracle(c4):
    wl, nll, S = c4["wl"], c4["nll"], c4["S"]
    fm, fv, fw = c4["fm"], c4["fv"], c4["fw"]
    N = nll.shape[0]
    rng = np.random.default_rng(4)
    # the reference's own arithmetic (linear-domain GMM.evaluate, per frame and state) on a few rows ...
    for r in [0, 31, 32, N - 1] + list(rng.integers(0, N, size=2)):
        states = np.concatenate([[0, 1, 15, 16, 1023], rng.integers(0, S, size=27)])
        ref = np.array([O.gmm_evaluate(wl["X"][r], fm[s], fv[s], fw[s]) for s in states])
        np.testing.assert_allclose(nll[r, states], ref, rtol=1e-10)
    # ... and the vectorised log-domain oracle on whole rows (every state, block boundaries included)
    rows = np.concatenate([[0, 31, 32, 63, 64, N - 33, N - 32, N - 1], rng.integers(0, N, size=120)])
    ref = O.gmm_neg_loglik_batch(wl["X"][rows], fm, fv, fw)
    np.testing.assert_allclose(nll[rows], ref, rtol=1e-10)


def test_c4_batching_independence_bitwise(c4):
    hip, ctx, wl, nll = c4["hip"], c4["ctx"], c4["wl"], c4["nll"]
    N = nll.shape[0]
    h = N // 2 + 13                                   # not a multiple of the 32-frame block
    m = min(N - h, 30011)
    b2 = hip.Batch(ctx, feats=wl["X"][h:h + m], offsets=np.array([0, m], dtype=np.int64))
    np.testing.assert_array_equal(b2.loglik(c4["gmm"]), nll[h:h + m])
    b2.close()


def test_c4_fp32_against_fp64_and_decode(c4):
    hip, ctx, wl, nll = c4["hip"], c4["ctx"], c4["wl"], c4["nll"]
    b32 = hip.Batch(ctx, feats=wl["X"], offsets=wl["off"], dtype=np.float32)
    n32 = b32.loglik(c4["gmm"], fetch=True)
    np.testing.assert_allclose(n32, nll, rtol=1e-3)            # north star: 1e-3 relative in fp32
    lat = hip.Lattices(ctx, [c4["graph"]])
    U, W = b32.U, wl["W"]
    ec32 = lat.viterbi(b32, want_path=False)["end_cost_flat"].reshape(U, W)
    ec64 = lat.viterbi(c4["batch"], want_path=False)["end_cost_flat"].reshape(U, W)
    assert np.mean(np.argmin(ec64, axis=1) == wl["words"]) == 1.0      # 64 words x 16 states, 1024-row stacked graph
    assert np.mean(np.argmin(ec32, axis=1) == wl["words"]) == 1.0
    # the winning word's cost through the oracle's reference-shaped DP for a few utterances
    n = wl["n"]
    g = c4["graph"]
    for u in (0, U // 2, U - 1):
        wd = int(wl["words"][u])
        E = nll[wl["off"][u]:wl["off"][u + 1], wd * n:(wd + 1) * n].T
        costs, _ = O.decode_states(E, np.zeros(n, dtype=bool), wl["trans"])
        np.testing.assert_allclose(ec64[u, wd], costs[-1, -1], rtol=1e-12)
    b32.close()
    lat.close()


def test_c4_loglik_subset_own_word_states(c4):
    """gh_loglik_subset at S = 1024: every utterance only asks for its own word's 16 states; the requested entries are
    the bits of the full matrix."""
    hip, ctx, wl, nll = c4["hip"], c4["ctx"], c4["wl"], c4["nll"]
    n = wl["n"]
    U = 300
    off = wl["off"][:U + 1]
    b = hip.Batch(ctx, feats=wl["X"][:off[-1]], offsets=off)
    lo = (wl["words"][:U] * n).astype(np.int32)
    hi = lo + n
    sub = b.loglik(c4["gmm"], fetch=True, state_ranges=(lo, hi))
    for u in range(U):
        np.testing.assert_array_equal(sub[off[u]:off[u + 1], lo[u]:hi[u]], nll[off[u]:off[u + 1], lo[u]:hi[u]])
    b.close()


# ------------------------------------------------------------------------------------------------ configs[2]
def c3_problem(U=2000):
    import bench
    wl = bench.synth_workload(1003, U)
    W = wl["W"]
    data = [wl["X"][wl["off"][u]:wl["off"][u + 1]] for u in range(U)]
    labels = [[int(w)] for w in wl["words"]]
    means0 = wl["means"] + 0.3 * np.random.default_rng(0).normal(size=wl["means"].shape)
    return means0, wl["vars"], wl["w"], [wl["trans"]] * W, data, labels


def test_c3_em_statistics_of_sampled_states_vs_numpy():
    """One E-step of BaumWelchTrainer on the configs[2] model with 2 000 utterances (~200 k frames): the statistics of
    sampled states equal occupancy-weighted responsibilities computed with numpy from the same occupancies."""
    from sr.recognition.train import BaumWelchTrainer
    from sr.recognition import _hip
    means, vars_, w, trans, data, labels = c3_problem()
    tr = BaumWelchTrainer(means, vars_, w, trans, data, labels)
    stats, xi, ll = tr.e_step()
    assert xi.shape == (tr.S,) and np.all(xi >= 0) and np.all(xi <= stats[:, :, 0].sum(axis=1) + 1e-6)
    assert np.isfinite(ll) and ll < 0
    gmm = _hip.PackedGMM(tr.ctx, tr.means, tr.vars, tr.weights)
    tr.batch.loglik(gmm, fetch=False, state_sets=tr.state_sets)
    occ = tr.lat.forward_backward(tr.batch, utt_lattice=tr.utt_graph, want_occ=True)["occ"]
    gmm.close()
    X = np.concatenate(data)
    N, D = X.shape
    np.testing.assert_allclose(occ.sum(axis=1), 1.0, rtol=1e-9)          # every frame sits in exactly one state (softly)
    np.testing.assert_allclose(stats[:, :, 0].sum(), N, rtol=1e-9)
    for s in (0, 7, 23, 49):
        m_, v_, w_ = tr.means[s], tr.vars[s], tr.weights[s]
        logc = np.log(w_) - 0.5 * (D * np.log(2 * np.pi) + np.log(v_).sum(axis=1))
        sel = np.nonzero(occ[:, s] > 0)[0]
        x = X[sel]
        ll_c = logc[None, :] - 0.5 * (((x[:, None, :] - m_[None]) ** 2) / v_[None]).sum(axis=2)
        r = np.exp(ll_c - ll_c.max(axis=1, keepdims=True))
        r = occ[sel, s][:, None] * r / r.sum(axis=1, keepdims=True)
        ref = np.zeros((tr.M, 1 + 2 * D))
        for m in range(tr.M):
            d = x - m_[m]
            ref[m, 0] = r[:, m].sum()
            ref[m, 1:1 + D] = (r[:, [m]] * d).sum(axis=0)
            ref[m, 1 + D:] = (r[:, [m]] * d * d).sum(axis=0)
        np.testing.assert_allclose(stats[s], ref, rtol=1e-8, atol=1e-9)
    tr.close()


def test_c4_shape_em_statistics_vs_numpy_and_generic_kernels(monkeypatch):
    """EM at the configs[3] shape -- 64 words x 16 states x 32 mixtures, ~100 k frames.  Three routes to the same
    statistics: (a) the matrix-core route -- chain forward-backward with 16 lanes per utterance, statistics kernel with a
    wave per 16 components of a state, normalised by the likelihood kernel's own per-state sums (M > 8); (b) the GENERIC
    route (fb_kernel, bw_stats_kernel<false, ...>: densities on the VALU) that was the only one until round 3;
    (c) numpy on the occupancies.  Then the device-resident session: its first E-step equals (a), iterations raise the
    likelihood, and it equals the call-by-call trainer."""
    import bench
    from sr.recognition.train import BaumWelchTrainer
    from sr.recognition import _hip
    U = 1000
    wl = bench.synth_workload(1004, U, W=64, n=16, M=32, D=39)
    W = wl["W"]
    data = [wl["X"][wl["off"][u]:wl["off"][u + 1]] for u in range(U)]
    labels = [[int(w)] for w in wl["words"]]
    means0 = wl["means"] + 0.3 * np.random.default_rng(0).normal(size=wl["means"].shape)
    tr = BaumWelchTrainer(means0, wl["vars"], wl["w"], [wl["trans"]] * W, data, labels)
    assert tr.session is not None and tr.batch.N >= 95000        # (round 3: the session covers n = 16, M = 32)
    stats, xi, ll = tr.e_step()                                  # (a), call by call
    assert "fb_chain" in tr.lat.forms()                          # 16-row chains: 16 lanes per utterance
    assert stats.shape == (1024, 32, 79) and np.isfinite(ll) and ll < 0
    monkeypatch.setenv("GMMHMM_FB", "generic")
    monkeypatch.setenv("GMMHMM_BW", "generic")
    stats_g, xi_g, ll_g = tr.e_step()                            # (b)
    monkeypatch.delenv("GMMHMM_FB")
    monkeypatch.delenv("GMMHMM_BW")
    np.testing.assert_allclose(ll, ll_g, rtol=1e-12)
    np.testing.assert_allclose(xi, xi_g, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(stats, stats_g, rtol=1e-8, atol=1e-9)
    gmm = _hip.PackedGMM(tr.ctx, tr.means, tr.vars, tr.weights)
    tr.batch.loglik(gmm, fetch=False, state_sets=tr.state_sets)
    occ = tr.lat.forward_backward(tr.batch, utt_lattice=tr.utt_graph, want_occ=True)["occ"]
    gmm.close()
    X = wl["X"]
    N, D = X.shape
    np.testing.assert_allclose(stats[:, :, 0].sum(), N, rtol=1e-9)
    np.testing.assert_allclose(stats[:, :, 0].sum(axis=1), occ.sum(axis=0), rtol=1e-9, atol=1e-9)
    for s in (0, 17, 333, 640, 1023):
        m_, v_, w_ = tr.means[s], tr.vars[s], tr.weights[s]
        logc = np.log(w_) - 0.5 * (D * np.log(2 * np.pi) + np.log(v_).sum(axis=1))
        sel = np.nonzero(occ[:, s] > 0)[0]
        x = X[sel]
        ll_c = logc[None, :] - 0.5 * (((x[:, None, :] - m_[None]) ** 2) / v_[None]).sum(axis=2)
        r = np.exp(ll_c - ll_c.max(axis=1, keepdims=True))
        r = occ[sel, s][:, None] * r / r.sum(axis=1, keepdims=True)
        ref = np.zeros((tr.M, 1 + 2 * D))
        for m in range(tr.M):
            d = x - m_[m]
            ref[m, 0] = r[:, m].sum()
            ref[m, 1:1 + D] = (r[:, [m]] * d).sum(axis=0)
            ref[m, 1 + D:] = (r[:, [m]] * d * d).sum(axis=0)
        np.testing.assert_allclose(stats[s], ref, rtol=1e-8, atol=1e-9)
    del occ
    # the device-resident iteration: its E-step is (a)'s, and it walks with the call-by-call trainer
    ref_tr = BaumWelchTrainer(means0, wl["vars"], wl["w"], [wl["trans"]] * W, data, labels, device_resident=False)
    h = [tr.iteration()]
    packed = tr.session.packed()
    np.testing.assert_allclose(packed[:tr.n_stats].reshape(stats.shape), stats, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(packed[tr.n_stats:tr.n_stats + tr.S], xi, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(packed[tr.n_stats + tr.S], ll, rtol=1e-11)
    h_ref = [ref_tr.iteration()]
    for _ in range(2):
        h.append(tr.iteration())
        h_ref.append(ref_tr.iteration())
    np.testing.assert_allclose(h, h_ref, rtol=1e-10)
    np.testing.assert_allclose(tr.means, ref_tr.means, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(tr.vars, ref_tr.vars, rtol=1e-6)
    np.testing.assert_allclose(tr.weights, ref_tr.weights, rtol=1e-7, atol=1e-11)
    assert h[1] > h[0] and h[2] >= h[1] - 1e-9 * abs(h[1]) and np.all(tr.vars > 0)
    tr.close()
    ref_tr.close()


def test_wide_mixture_statistics_need_the_models_own_likelihoods():
    """M > 8: gh_bw_accumulate normalises with the batch's likelihood matrix.  When that matrix was computed with ANOTHER
    model (or the model was updated in place since), the library notices (model / matrix serial numbers) and takes the
    generic kernel, which computes its own densities: same statistics either way."""
    from sr.recognition import _hip
    from sr.recognition.continuous_speech import packed_lattice
    ctx = _hip.default_context()
    rng = np.random.default_rng(3)
    W, n, M, D = 3, 9, 12, 13
    means = rng.normal(size=(W * n, M, D)) * 2
    vars_ = rng.uniform(0.5, 1.5, size=(W * n, M, D))
    w = rng.dirichlet(np.ones(M), size=W * n)
    trans = np.full((n, n), np.inf)
    for i in range(n):
        trans[i, i] = -np.log(0.8)
        if i:
            trans[i, i - 1] = -np.log(0.2)
    words = rng.integers(0, W, size=40)
    xs = []
    for wd in words:
        T = int(rng.integers(2 * n, 5 * n))
        st = wd * n + np.minimum(np.arange(T) * n // T, n - 1)
        xs.append(means[st, rng.integers(0, M, size=T)] + rng.normal(size=(T, D)))
    graphs = [packed_lattice([trans] * W, n, [[k]])[0] for k in range(W)]
    lat = _hip.Lattices(ctx, graphs)
    b = _hip.Batch(ctx, xs)
    g = _hip.PackedGMM(ctx, means, vars_, w)
    other = _hip.PackedGMM(ctx, means + 0.5, vars_, w)

    def stats_after(loglik_model):
        b.loglik(g, fetch=False)
        lat.forward_backward(b, utt_lattice=words.astype(np.int32), want_occ=True, fetch_occ=False)
        if loglik_model is not g:
            b.loglik(loglik_model, fetch=False)          # the matrix now belongs to another model; gamma is still g's
        return b.bw_accumulate(g)

    own = stats_after(g)
    stale = stats_after(other)
    np.testing.assert_allclose(stale, own, rtol=1e-9, atol=1e-11)
    assert np.isclose(own[:, :, 0].sum(), b.N, rtol=1e-9)
    for h in (b, lat, g, other):
        h.close()


def test_c3_em_likelihood_is_monotone_and_recovers_the_model():
    from sr.recognition.train import BaumWelchTrainer
    means, vars_, w, trans, data, labels = c3_problem()
    tr = BaumWelchTrainer(means, vars_, w, trans, data, labels, update_transitions=False)   # (fixed transitions: the EM guarantee)
    hist = tr.fit(5)
    assert all(b >= a - 1e-9 * abs(a) for a, b in zip(hist, hist[1:])), hist
    assert hist[-1] > hist[0]
    assert np.all(tr.vars > 0) and np.all(np.isfinite(tr.means))
    np.testing.assert_allclose(tr.weights.sum(axis=1), 1.0, rtol=1e-9)
    tr.close()


def _c3_worker(rank, world, port, out_dir, backend):
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "speech-recognition_amd"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = rank if backend == "nccl" else 0
    os.environ["GMMHMM_DEVICE"] = str(dev)
    if backend == "nccl":
        import torch
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sr.recognition.train import BaumWelchTrainer
        from sr.recognition.parallel import shard_utterances, StatsAllReducer
        means, vars_, w, trans, data, labels = c3_problem()
        mine = shard_utterances([len(x) for x in data], world)[rank]
        tr = BaumWelchTrainer(means, vars_, w, trans, [data[i] for i in mine], [labels[i] for i in mine], device=dev,
                              reducer=StatsAllReducer(gpu_index=dev))
        hist = tr.fit(3)
        np.savez(os.path.join(out_dir, "c3rank%d.npz" % rank), means=tr.means, vars=tr.vars, w=tr.weights,
                 hist=np.array(hist), frames=tr.batch.N)
        tr.close()
    finally:
        dist.destroy_process_group()


def test_c3_two_ranks_equal_one_rank(tmp_path):
    import torch.multiprocessing as mp
    from sr.recognition.train import BaumWelchTrainer
    from test_gpu_dist import _free_port, _n_gpus
    backend = "nccl" if _n_gpus() >= 2 else "gloo"
    mp.spawn(_c3_worker, args=(2, _free_port(), str(tmp_path), backend), nprocs=2, join=True)
    means, vars_, w, trans, data, labels = c3_problem()
    tr = BaumWelchTrainer(means, vars_, w, trans, data, labels)
    hist = tr.fit(3)
    r0, r1 = np.load(tmp_path / "c3rank0.npz"), np.load(tmp_path / "c3rank1.npz")
    assert int(r0["frames"]) + int(r1["frames"]) == tr.batch.N
    assert abs(int(r0["frames"]) - int(r1["frames"])) <= 150            # longest-first greedy balance
    for r in (r0, r1):
        np.testing.assert_allclose(r["hist"], hist, rtol=1e-10)
        np.testing.assert_allclose(r["means"], tr.means, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(r["vars"], tr.vars, rtol=1e-7)
        np.testing.assert_allclose(r["w"], tr.weights, rtol=1e-8, atol=1e-12)
    tr.close()


# --------------------------------------------------------------------------- fp32 off standardised features
@pytest.mark.parametrize("offset", [0.0, 3.0, 10.0, 100.0, 1000.0])
def test_fp32_loglik_on_unstandardised_features(offset):
    """The GEMM form sum_d P_d [x^2 | x]_d + C cancels terms of size (x^2 + mu^2) / sigma^2.  The reference path only
    ever sees per-utterance standardised features (sr/core.py:41-44), where fp32 holds the north star's 1e-3 with
    room to spare; this pins what happens when a caller skips that step and every dimension sits `offset` standard
    deviations away from zero: the fp32 kernel keeps 1e-3 relative."""
    from sr.recognition import _hip
    ctx = _hip.default_context()
    rng = np.random.default_rng(int(offset) + 5)
    S, M, D, N = 50, 8, 39, 4000
    means = rng.normal(size=(S, M, D)) + offset
    vars_ = rng.uniform(0.5, 1.5, size=(S, M, D))
    w = rng.dirichlet(np.ones(M), size=S)
    X = means[rng.integers(0, S, N), rng.integers(0, M, N)] + rng.normal(size=(N, D)) * np.sqrt(1.0)
    ref = O.gmm_neg_loglik_batch(X, means, vars_, w)
    gmm = _hip.PackedGMM(ctx, means, vars_, w)
    b64 = _hip.Batch(ctx, feats=X, offsets=[0, N])
    np.testing.assert_allclose(b64.loglik(gmm), ref, rtol=1e-9)
    b32 = _hip.Batch(ctx, feats=X, offsets=[0, N], dtype=np.float32)
    got = b32.loglik(gmm)
    # inputs themselves are rounded to fp32: |d nll| <= sum |x - mu| / var * |x| * 2^-24
    np.testing.assert_allclose(got, ref, rtol=1e-3)
    b32.close()
    b64.close()
    gmm.close()


def test_c3_training_loop_transitions_convergence_and_pickles(tmp_path):
    """The soft form of the reference's training loop (continuous_speech.py:144-179) at the configs[2] shape:
    transition costs re-estimated from expected counts, one pickle per word model and iteration, the reference's stop
    rule -- and the emitted pickles are HMM objects that score like the trainer's own parameters."""
    import pickle
    from sr.recognition.train import BaumWelchTrainer
    from sr.recognition import _hip
    import sr.recognition as R
    means, vars_, w, trans, data, labels = c3_problem(800)
    out = str(tmp_path / "models")
    tr = BaumWelchTrainer(means, vars_, w, trans, data, labels, output_path=out)
    t0 = [t.copy() for t in tr.transitions]
    stats, xi, _ = tr.e_step()
    counts = stats[:, :, 0].sum(axis=1)
    # expected self transitions of a few states from the forward / backward matrices of the generic kernel
    gmm = _hip.PackedGMM(tr.ctx, tr.means, tr.vars, tr.weights)
    tr.batch.loglik(gmm, fetch=False, state_sets=tr.state_sets)
    nll = tr.batch.loglik(gmm, fetch=True)
    r = tr.lat.forward_backward(tr.batch, utt_lattice=tr.utt_graph, want_matrices=True, want_self_xi=True)
    np.testing.assert_allclose(r["self_xi"], xi, rtol=1e-9, atol=1e-9)       # generic kernel (atomics) == chain kernel
    ref = np.zeros(tr.S)
    n = tr.n
    for u in range(0, tr.batch.U, 40):
        wd = labels[u][0]
        al, be, lp = r["alpha"][u], r["beta"][u], r["logp"][u]
        E = nll[tr.batch.offsets[u]:tr.batch.offsets[u + 1], wd * n:(wd + 1) * n].T      # [n, T]
        rows = [k for k in range(al.shape[0]) if al.shape[0] == n or 1 <= k <= n]          # (chain may be NES-wrapped)
        for j, k in enumerate(rows[:n]):
            c_self = t0[wd][j, j]
            ref[wd * n + j] += np.exp(al[k, :-1] - c_self - E[j, 1:] + be[k, 1:] - lp).sum()
    got = np.zeros(tr.S)
    sub = _hip.Batch(tr.ctx, [data[u] for u in range(0, tr.batch.U, 40)])
    lo = np.array([labels[u][0] * n for u in range(0, tr.batch.U, 40)], dtype=np.int32)
    sub.loglik(gmm, fetch=False, state_ranges=(lo, lo + n))
    got = tr.lat.forward_backward(sub, utt_lattice=tr.utt_graph[::40], want_self_xi=True)["self_xi"]
    np.testing.assert_allclose(got, ref, rtol=1e-8, atol=1e-10)
    sub.close()
    gmm.close()
    hist = tr.fit(30, until_converged=True)
    assert tr.converged and len(hist) < 30, len(hist)
    assert hist[-1] > hist[0]
    for wi in range(tr.W):
        t = tr.transitions[wi]
        assert np.all(np.isfinite(np.diag(t))) and not np.allclose(np.diag(t), np.diag(t0[wi]))
        for si in range(n - 1):       # a proper pair of probabilities on every non-final state
            np.testing.assert_allclose(np.exp(-t[si, si]) + np.exp(-t[si + 1, si]), 1.0, rtol=1e-9)
        assert np.all(np.isinf(t[np.triu_indices(n, 1)]))
    files = sorted(os.listdir(out))
    assert files == sorted("%d.pkl" % i for i in range(tr.W))
    models = [pickle.load(open(os.path.join(out, "%d.pkl" % i), "rb")) for i in range(tr.W)]
    assert all(type(m) is R.HMM and sorted(m.__dict__) == ["gmm_states", "mu", "n_segments", "segments", "sigma",
                                                           "transitions", "use_em", "use_gmm"] for m in models)
    for wi in (0, 7):
        np.testing.assert_array_equal(models[wi].transitions, tr.transitions[wi])
        np.testing.assert_array_equal(np.array([[d.mean for d in g.dists] for g in models[wi].gmm_states]),
                                      tr.means.reshape(tr.W, n, tr.M, tr.D)[wi])
    # the pickled models recognise their own training words
    from sr.recognition.batch import IsolatedWordRecognizer
    rec = IsolatedWordRecognizer(models)
    words, _ = rec.recognize(data[:200])
    assert np.mean(words == np.array([l[0] for l in labels[:200]])) == 1.0
    tr.close()
