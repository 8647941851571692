# -*- coding: utf-8 -*-
"""Lock-step refit of all states (gh_kmeans_assign_multi / gh_em_accumulate_multi, `lockstep.LockstepFitter`):
the multi-state launches against the single-state entry points and numpy, the converged mask, and the sharded
(two-rank) run with one collective per lock-step iteration.  The end-to-end parity of the training loops built on it
(HMM.fit, continuous_train vs goldens captured from the reference) is in test_gpu_api.py."""
import os
import sys

import numpy as np
import pytest

from oracle import ref_numpy as O

pytestmark = pytest.mark.gpu


def _problem(seed=3, S=7, D=13, k=4):
    rng = np.random.default_rng(seed)
    segs = []
    for s in range(S):
        n = int(rng.integers(5, 400)) if s != 2 else 64          # (one segment of exactly one tile)
        c = rng.normal(size=(k, D)) * 3
        segs.append(c[rng.integers(0, k, n)] + rng.normal(size=(n, D)) * rng.uniform(0.5, 1.5))
    return rng, segs


def test_kmeans_assign_multi_equals_single_state_calls():
    from sr.recognition import _hip
    ctx = _hip.default_context()
    rng, segs = _problem()
    S, D, k = len(segs), segs[0].shape[1], 4
    off = np.concatenate([[0], np.cumsum([len(x) for x in segs])]).astype(np.int64)
    b = _hip.Batch(ctx, feats=np.concatenate(segs), offsets=[0, int(off[-1])])
    cent = rng.normal(size=(S, k, D)) * 2
    var = rng.uniform(0.5, 2.0, size=(S, D))
    for v in (var, None):
        clusters, changed, sums = b.kmeans_assign_multi(off, cent, var=v, want_sums=True)
        for s in range(S):
            ref = b.kmeans_assign(cent[s], var=None if v is None else v[s], first=int(off[s]), count=len(segs[s]))
            np.testing.assert_array_equal(clusters[off[s]:off[s + 1]], ref)
            assert changed[s] == len(segs[s])                                  # everything moved away from -1
            for c in range(k):
                x = segs[s][ref == c]
                np.testing.assert_allclose(sums[s, c, :D], x.sum(axis=0), rtol=1e-12, atol=1e-12)
                assert sums[s, c, D] == len(x)
        # a second sweep with the same centroids changes nothing; inactive states are left alone
        active = np.ones(S, dtype=np.uint8)
        active[[1, 4]] = 0
        before = clusters.copy()
        clusters[off[1]:off[2]] = 7
        c2, changed2, _ = b.kmeans_assign_multi(off, cent, var=v, clusters=clusters, active=active)
        assert np.all(changed2 == 0)
        np.testing.assert_array_equal(c2[off[1]:off[2]], 7)
        c2[off[1]:off[2]] = before[off[1]:off[2]]
        np.testing.assert_array_equal(c2, before)
    b.close()


def test_em_accumulate_multi_equals_single_state_calls():
    from sr.recognition import _hip
    ctx = _hip.default_context()
    rng, segs = _problem(seed=8, S=6, D=39, k=8)
    S, D, k = len(segs), 39, 8
    off = np.concatenate([[0], np.cumsum([len(x) for x in segs])]).astype(np.int64)
    b = _hip.Batch(ctx, feats=np.concatenate(segs), offsets=[0, int(off[-1])])
    mean = rng.normal(size=(S, k, D)) * 2
    var = rng.uniform(0.5, 2.0, size=(S, k, D))
    w = rng.dirichlet(np.ones(k), size=S)
    segs[3][5] += 500.0                     # a frame every component underflows on (linear domain): all-zero row
    b.close()
    b = _hip.Batch(ctx, feats=np.concatenate(segs), offsets=[0, int(off[-1])])
    active = np.ones(S, dtype=np.uint8)
    active[1] = 0
    stats, ll = b.em_accumulate_multi(off, mean, var, w, active=active)
    assert np.all(stats[1] == 0) and ll[1] == 0
    for s in range(S):
        if not active[s]:
            continue
        ref, rl = b.em_accumulate(mean[s], var[s], w[s], first=int(off[s]), count=len(segs[s]))
        np.testing.assert_allclose(stats[s], ref, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(ll[s], rl, rtol=1e-12)
        # and numpy: responsibilities in the reference's linear domain
        p = np.array([O.gmm_evaluate(x, mean[s], var[s], w[s], neg_log=False) for x in segs[s]])
        rs = p.sum(axis=1, keepdims=True)
        r = np.where(rs == 0, 0.0, p / np.where(rs == 0, 1.0, rs))
        np.testing.assert_allclose(stats[s][:, 0], r.sum(axis=0), rtol=1e-9)
    np.testing.assert_allclose(stats[3][:, 0].sum(), len(segs[3]) - 1, rtol=1e-9)     # the underflowing frame moved nothing
    b.close()


@pytest.mark.parametrize("k,D", [(1, 5), (2, 6), (3, 13), (5, 39), (8, 40), (7, 2)])
def test_tile_kernels_for_every_split_of_the_components_over_their_two_waves(k, D):
    """The two tile kernels run two waves per tile, each with half of the clusters / components (k = 1: the second wave has
    none; odd k: uneven halves; D even: padded tile rows): assignments == the one-state kernel's, statistics == the
    one-state kernel's (another kernel: 1e-11) and numpy's responsibilities."""
    from sr.recognition import _hip
    ctx = _hip.default_context()
    rng, segs = _problem(seed=100 + k, S=5, D=D, k=max(k, 2))
    S = len(segs)
    off = np.concatenate([[0], np.cumsum([len(x) for x in segs])]).astype(np.int64)
    b = _hip.Batch(ctx, feats=np.concatenate(segs), offsets=[0, int(off[-1])])
    cent = rng.normal(size=(S, k, D)) * 2
    var1 = rng.uniform(0.5, 2.0, size=(S, D))
    for v in (var1, None):
        clusters, changed, sums = b.kmeans_assign_multi(off, cent, var=v, want_sums=True)
        for s in range(S):
            ref = b.kmeans_assign(cent[s], var=None if v is None else v[s], first=int(off[s]), count=len(segs[s]))
            np.testing.assert_array_equal(clusters[off[s]:off[s + 1]], ref)
            x = segs[s]
            d2 = ((x[:, None, :] - cent[s][None]) ** 2 / (1.0 if v is None else v[s][None, None, :])).sum(axis=2)
            assert np.mean(np.argmin(d2, axis=1) == ref) > 0.999            # (numpy's own summation order: ties aside)
            for c in range(k):
                assert sums[s, c, D] == np.sum(ref == c)
    mean = rng.normal(size=(S, k, D)) * 2
    var = rng.uniform(0.5, 2.0, size=(S, k, D))
    w = rng.dirichlet(np.ones(k), size=S)
    stats, ll = b.em_accumulate_multi(off, mean, var, w)
    for s in range(S):
        ref, rl = b.em_accumulate(mean[s], var[s], w[s], first=int(off[s]), count=len(segs[s]))
        np.testing.assert_allclose(stats[s], ref, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(ll[s], rl, rtol=1e-12)
        p = np.array([O.gmm_evaluate(x, mean[s], var[s], w[s], neg_log=False) for x in segs[s]])
        rs = p.sum(axis=1, keepdims=True)
        r = np.where(rs == 0, 0.0, p / np.where(rs == 0, 1.0, rs))
        np.testing.assert_allclose(stats[s][:, 0], r.sum(axis=0), rtol=1e-9, atol=1e-12)
    b.close()


def test_lockstep_fitter_equals_state_after_state():
    """LockstepFitter.split_and_fit on 6 states == the sequential loop of the reference (kmeans + GMM.em per state, in
    order) under the same seed: parameters 1e-9, the same number of EM iterations per state."""
    import contextlib
    import io
    import sr.recognition as R
    from sr.recognition.lockstep import LockstepFitter
    rng, segs = _problem(seed=21, S=6, D=6, k=4)
    n_g = 8
    mus = [x.mean(axis=0) for x in segs]
    sig = [x.var(axis=0) for x in segs]

    def fresh():
        return [R.GMM(m.copy(), s.copy(), n_g) for m, s in zip(mus, sig)]
    np.random.seed(11)
    seq, seq_log = fresh(), []
    for st, x, m in zip(seq, segs, mus):
        centroids = m.reshape(1, -1)
        weights = np.full(n_g, 1 / len(x))
        for i in range(int(np.log(n_g))):
            k = 2 ** (i + 1)
            centroids = np.concatenate([centroids * 0.9, centroids * 1.1], axis=0)
            clusters, centroids, variance = R.kmeans(x, k, centroids, dist_fun=R.mahalanobis)
            ids, cnt = np.unique(clusters, return_counts=True)
            for c in ids:
                weights[c] = cnt[c] / len(x)
            st.update_models(centroids, variance, weights[:k])
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                st.em(x, k)
            seq_log.append(buf.getvalue().rsplit("EM converged at iteration:", 1)[-1].split()[0])
    np.random.seed(11)
    lock = fresh()
    f = LockstepFitter(segs)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        f.split_and_fit(lock, start_centroids=mus, weight_divisor=[len(x) for x in segs], n_gaussians=n_g)
    f.close()
    for a, b_ in zip(seq, lock):
        np.testing.assert_allclose(np.array([d.mean for d in b_.dists]), np.array([d.mean for d in a.dists]), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(np.array([d.cov for d in b_.dists]), np.array([d.cov for d in a.dists]), rtol=1e-9)
        np.testing.assert_allclose(b_.w, a.w, rtol=1e-9)
        np.testing.assert_allclose(b_.mu_old, a.mu_old, rtol=1e-9, atol=1e-12)
    lock_iters = sorted(int(t.split()[0]) for t in buf.getvalue().split("EM converged at iteration:")[1:])
    assert lock_iters == sorted(int(v) for v in seq_log)


def _shard_worker(rank, world, port, out_dir):
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "speech-recognition_amd"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GMMHMM_DEVICE"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import contextlib
        import io
        import sr.recognition as R
        from sr.recognition.lockstep import LockstepFitter
        from sr.recognition.parallel import StatsAllReducer
        rng, segs = _problem(seed=21, S=6, D=6, k=4)
        mine = [x[rank::world] for x in segs]                       # every rank holds part of every state's frames
        states = [R.GMM(x.mean(axis=0), x.var(axis=0), 4) for x in segs]
        for st, x in zip(states, segs):                             # identical start on every rank
            st.update_models(np.stack([x.mean(axis=0) * 0.9, x.mean(axis=0) * 1.1]), np.stack([x.var(axis=0)] * 2),
                             np.array([0.5, 0.5]))
        f = LockstepFitter(mine, reducer=StatsAllReducer())
        assert f.sharded
        with contextlib.redirect_stdout(io.StringIO()):
            f.em(states, 2)
        n_em = f.collectives
        np.random.seed(100 + rank)
        cl, cent, cov = f.kmeans(2, np.stack([np.stack([x.mean(axis=0) * 0.9, x.mean(axis=0) * 1.1]) for x in segs]))
        np.savez(os.path.join(out_dir, "ls%d.npz" % rank), means=np.array([[d.mean for d in g.dists[:2]] for g in states]),
                 vars=np.array([[d.cov for d in g.dists[:2]] for g in states]), w=np.array([g.w[:2] for g in states]),
                 cent=cent, cov=cov, counts=np.array([np.bincount(c, minlength=2) for c in cl]), n_em=n_em,
                 n_global=f.n_global)
        f.close()
    finally:
        dist.destroy_process_group()


def test_lockstep_fitter_sharded_over_two_ranks(tmp_path):
    """Two ranks (gloo, both on GPU 0), every state's frames split between them: EM from a common start equals the
    one-rank EM (the all-reduced statistics are the same sums), both ranks end with the same model, the k-means
    converges to centroids both ranks agree on."""
    import contextlib
    import io
    import torch.multiprocessing as mp
    import sr.recognition as R
    from sr.recognition.lockstep import LockstepFitter
    from test_gpu_dist import _free_port
    mp.spawn(_shard_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "ls0.npz"), np.load(tmp_path / "ls1.npz")
    rng, segs = _problem(seed=21, S=6, D=6, k=4)
    states = [R.GMM(x.mean(axis=0), x.var(axis=0), 4) for x in segs]
    for st, x in zip(states, segs):
        st.update_models(np.stack([x.mean(axis=0) * 0.9, x.mean(axis=0) * 1.1]), np.stack([x.var(axis=0)] * 2), np.array([0.5, 0.5]))
    f = LockstepFitter(segs)
    with contextlib.redirect_stdout(io.StringIO()):
        f.em(states, 2)
    f.close()
    means = np.array([[d.mean for d in g.dists[:2]] for g in states])
    vars_ = np.array([[d.cov for d in g.dists[:2]] for g in states])
    for r in (r0, r1):
        np.testing.assert_array_equal(r["n_global"], [len(x) for x in segs])
        np.testing.assert_allclose(r["means"], means, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(r["vars"], vars_, rtol=1e-7)
        np.testing.assert_allclose(r["w"], np.array([g.w[:2] for g in states]), rtol=1e-8)
    np.testing.assert_array_equal(r0["cent"], r1["cent"])
    np.testing.assert_array_equal(r0["cov"], r1["cov"])
    assert np.all((r0["counts"] + r1["counts"]).sum(axis=1) == [len(x) for x in segs])
    assert int(r0["n_em"]) == int(r1["n_em"]) and int(r0["n_em"]) >= 2          # one collective per lock-step iteration (+ the frame counts)


def _ct_problem():
    from conftest import load_golden
    g = load_golden("G11_continuous_train")
    W, U = int(g["n_words"]), int(g["n_utts"])
    data = [g["x%d" % i] for i in range(U)]
    labels = [[int(v) for v in g["labels%d" % i]] for i in range(U)]
    return g, W, data, labels


def _ct_models(R, g, W):
    from test_gpu_api import make_hmm
    return [make_hmm(R, g["init%d_means" % wi], g["init%d_vars" % wi], g["init%d_w" % wi], g["init%d_transitions" % wi])
            for wi in range(W)]


def _ct_worker(rank, world, port, out_dir):
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "speech-recognition_amd"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GMMHMM_DEVICE"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import contextlib
        import io
        import warnings
        import sr.recognition as R
        from sr.recognition.parallel import StatsAllReducer, shard_utterances
        g, W, data, labels = _ct_problem()
        mine = shard_utterances([len(x) for x in data], world)[rank]
        models = _ct_models(R, g, W)
        out = os.path.join(out_dir, "rank%d" % rank)
        os.makedirs(out, exist_ok=True)
        np.random.seed(7 + rank)
        with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            R.continuous_train([data[i] for i in mine], models, [labels[i] for i in mine], out, n_gaussians=4,
                               n_segments=5, max_iteration=2, reducer=StatsAllReducer())
    finally:
        dist.destroy_process_group()


def test_continuous_train_sharded_over_two_ranks(tmp_path):
    """continuous_train with the utterances split over two ranks: both ranks write the SAME models after every
    iteration (alignment local, refit in lock-step over the ranks, counts all-reduced), and those models align the
    training utterances at finite cost."""
    import pickle
    import torch.multiprocessing as mp
    import sr.recognition as R
    from test_gpu_dist import _free_port
    mp.spawn(_ct_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    g, W, data, labels = _ct_problem()
    for wi in range(W):
        a = pickle.load(open(tmp_path / "rank0" / ("%d.pkl" % wi), "rb"))
        b = pickle.load(open(tmp_path / "rank1" / ("%d.pkl" % wi), "rb"))
        np.testing.assert_array_equal(a.transitions, b.transitions)
        for sa, sb in zip(a.gmm_states, b.gmm_states):
            np.testing.assert_array_equal(np.array([d.mean for d in sa.dists]), np.array([d.mean for d in sb.dists]))
            np.testing.assert_array_equal(np.array([d.cov for d in sa.dists]), np.array([d.cov for d in sb.dists]))
            np.testing.assert_array_equal(sa.w, sb.w)
            assert np.all(np.isfinite(np.array([d.mean for d in sa.dists])))
    models = [pickle.load(open(tmp_path / "rank0" / ("%d.pkl" % wi), "rb")) for wi in range(W)]
    for x, lab in list(zip(data, labels))[:4]:
        seq, trans, ends = R.build_state_sequences(models, [[l] for l in lab])
        costs, path = R.decode_hmm_states(x, seq, trans, end_points=[[e, -1] for e in ends])
        assert np.isfinite(costs[ends[-1], -1]) and len(path) > 0


def test_batch_gather_and_resident_clusters():
    """gh_batch_gather: rows of a resident batch regrouped on the device equal the host's fancy index, fp64 and fp32;
    gh_kmeans_assign_multi with the assignments resident in the batch walks through the same assignments as with a
    host array, and its frame-order sums divided by the counts are np.mean of the clusters bit for bit."""
    from sr.recognition import _hip
    ctx = _hip.default_context()
    rng = np.random.default_rng(3)
    xs = [rng.normal(size=(int(rng.integers(1, 40)), 7)) for _ in range(30)]
    X = np.concatenate(xs)
    rows = rng.permutation(len(X))[:500]
    for dt in (np.float64, np.float32):
        b = _hip.Batch(ctx, xs, dtype=dt)
        g = b.gather(rows, offsets=[0, 100, 100, 500])
        got = g.features()
        assert [len(u) for u in got] == [100, 0, 400]
        np.testing.assert_array_equal(np.concatenate(got), X.astype(dt)[rows])
        g.close()
        b.close()
    b = _hip.Batch(ctx, feats=X, offsets=[0, len(X)])
    off = np.array([0, 200, 200, len(X)], dtype=np.int64)
    cent = rng.normal(size=(3, 4, 7))
    var = rng.uniform(0.5, 2.0, size=(3, 7))
    host = np.full(len(X), -1, dtype=np.int32)
    b.resident_clusters(reset=True, fetch=False)
    for it in range(3):
        host, ch_h, sums_h = b.kmeans_assign_multi(off, cent, var=var, clusters=host, want_sums=True)
        none, ch_r, sums_r = b.kmeans_assign_multi(off, cent, var=var, clusters=_hip.RESIDENT, want_sums=True)
        assert none is None
        np.testing.assert_array_equal(ch_r, ch_h)
        np.testing.assert_array_equal(sums_r, sums_h)
        np.testing.assert_array_equal(b.resident_clusters(), host)
        for s in (0, 2):
            seg, cl = X[off[s]:off[s + 1]], host[off[s]:off[s + 1]]
            for c in range(4):
                if (cl == c).any():
                    np.testing.assert_array_equal(sums_h[s, c, :7] / sums_h[s, c, 7], np.mean(seg[cl == c, :], axis=0))
                    cent[s, c] = sums_h[s, c, :7] / sums_h[s, c, 7]
    assert (b.resident_clusters(reset=True) == -1).all()
    b.close()


def test_device_session_equals_the_host_loop():
    """The device-resident refit (gh_fit_*: partition variances, lock-step k-means and EM with the stop rules on the
    device) against the host-orchestrated loop with np.cov (compat_cov=True): same cluster ids, same iteration counts,
    centroids bit for bit (they are sums of the same frames in the same order), variances / EM parameters to 1e-9."""
    import sr.recognition as R
    from sr.recognition.lockstep import LockstepFitter
    from test_gpu_api import quiet
    rng = np.random.default_rng(3)
    D, k = 7, 4
    segs = [rng.normal(size=(int(n), D)) * rng.uniform(0.5, 2.0, size=D) + rng.normal(size=D) * 3 for n in (400, 130, 900, 65, 64, 257)]
    S = len(segs)
    dev, host = LockstepFitter(segs), LockstepFitter(segs, compat_cov=True)
    assert dev.fit is not None and host.fit is None
    np.testing.assert_array_equal(dev.segment_means(), host.segment_means())
    parts = [rng.integers(0, k, size=len(x)) for x in segs]
    c0 = np.stack([np.stack([x.mean(axis=0) * f for f in (0.8, 0.9, 1.1, 1.2)]) for x in segs])
    cl_d, ce_d, cov_d = dev.kmeans(k, c0, partitions=parts)
    cl_h, ce_h, cov_h = host.kmeans(k, c0, partitions=parts)
    np.testing.assert_allclose(cov_d, cov_h, rtol=1e-12)
    for a, b in zip(cl_d, cl_h):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(ce_d, ce_h)
    np.testing.assert_array_equal(dev.last_counts, np.array([np.bincount(c, minlength=k) for c in cl_h]))
    # frame-order sums and the np.array_equal stop rule in EVERY iteration (GMMHMM_KMEANS_EXACT=1) against the default
    # (sums over 512-frame pieces while iterating, one frame-order pass at the end): same assignments, same centroid bits
    os.environ["GMMHMM_KMEANS_EXACT"] = "1"
    try:
        cl_e, ce_e, cov_e = dev.kmeans(k, c0, partitions=parts)
    finally:
        del os.environ["GMMHMM_KMEANS_EXACT"]
    for a, b in zip(cl_e, cl_h):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(ce_e, ce_h)
    np.testing.assert_allclose(cov_e, cov_d, rtol=1e-12)      # (two frame-order passes against one streaming pass around the shift point)
    # EM from the k-means result, "old" parameters as a fresh GMM holds them
    def states():
        out = []
        for s in range(S):
            g = R.GMM(segs[s].mean(axis=0), segs[s].var(axis=0), k)
            g.update_models(ce_h[s].copy(), cov_h[s].copy(), np.bincount(cl_h[s], minlength=k) / len(segs[s]))
            out.append(g)
        return out
    sd, sh = states(), states()
    with quiet():
        dev.em(sd, k, max_iteration=60)
        host.em(sh, k, max_iteration=60)
    for a, b in zip(sd, sh):
        np.testing.assert_allclose(a.w, b.w, rtol=1e-9)
        np.testing.assert_allclose(a.w_old, b.w_old, rtol=1e-9)
        np.testing.assert_allclose(a.mu_old, b.mu_old, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(a.sigma_old, b.sigma_old, rtol=1e-9)
        for da, db in zip(a.dists, b.dists):
            np.testing.assert_allclose(da.mean, db.mean, rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(da.cov, db.cov, rtol=1e-9)
    dev.close()
    host.close()
