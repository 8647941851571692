#!/usr/bin/env python3
# -*- coding: utf-8 -*-
"""
Capture golden input/output vectors by RUNNING the reference's own
`sr.recognition` (pure Python, /root/reference, read-only) on small seeded
inputs, and write them as compressed .npz fixtures under tests/golden/.

Runs only in the build container (the reference never travels to the GPU box);
the fixtures hold data only -- inputs and the reference's outputs -- no source.

Import recipe: `import sr` itself needs pyaudio, so a
stub parent package is registered and only `sr.recognition` is imported; numpy
2.x dropped `np.int` / `np.alltrue`, which the reference still uses, so they are
aliased for the duration of this script.

Usage:  python -B tools/make_goldens.py [--only G4,G11]
"""
import argparse
import ast
import contextlib
import importlib
import io
import os
import sys
import tempfile
import types
import warnings

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def load_reference():
    if not os.path.isdir(os.path.join(REF, "sr", "recognition")):
        sys.exit("reference not present at %s -- goldens can only be generated in the build container" % REF)
    sys.dont_write_bytecode = True
    np.int = int            # decode.py:24,95; kmeans.py:126,138,200
    np.alltrue = np.all     # continuous_speech.py:173
    pkg = types.ModuleType("sr")
    pkg.__path__ = [os.path.join(REF, "sr")]
    sys.modules["sr"] = pkg
    return importlib.import_module("sr.recognition")


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        yield


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    arrays["numpy_version"] = np.array(np.__version__)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


# ----------------------------------------------------------------- synthetic inputs
def synth_model(rng, n_words, n_states, M, D):
    """means ~ N(0,1), vars ~ U[0.5,1.5], weights ~ Dirichlet(1),
    left-to-right costs: self -log .9, next -log .1, last self -log 1."""
    means = rng.normal(size=(n_words, n_states, M, D))
    vars_ = rng.uniform(0.5, 1.5, size=(n_words, n_states, M, D))
    w = rng.dirichlet(np.ones(M), size=(n_words, n_states))
    trans = np.full((n_states, n_states), np.inf)
    for i in range(n_states):
        trans[i, i] = -np.log(0.9) if i < n_states - 1 else -np.log(1.0)
        if i < n_states - 1:
            trans[i + 1, i] = -np.log(0.1)
    return means, vars_, w, trans


def synth_utt(rng, means, vars_, words, tmin, tmax):
    """Frames drawn from a uniformly segmented state path through `words`."""
    n_states, M, D = means.shape[1:]
    xs = []
    for wd in words:
        T = int(rng.integers(tmin, tmax + 1))
        st = np.minimum((np.arange(T) * n_states) // T, n_states - 1)
        comp = rng.integers(0, M, size=T)
        xs.append(means[wd, st, comp] + np.sqrt(vars_[wd, st, comp]) * rng.normal(size=(T, D)))
    return np.concatenate(xs, axis=0)


def make_gmm(R, mean_md, var_md, w_m):
    g = R.GMM(mean_md[0].copy(), var_md[0].copy(), len(w_m))
    g.update_models(mean_md.copy(), var_md.copy(), w_m.copy())
    return g


def make_hmm(R, means, vars_, w, trans):
    h = R.HMM(means.shape[0])
    h.gmm_states = [make_gmm(R, means[s], vars_[s], w[s]) for s in range(means.shape[0])]
    h.transitions = trans.copy()
    h.mu = means[:, 0].copy()
    h.sigma = vars_[:, 0].copy()
    return h


def pack_gmm(g):
    return (np.array([d.mean for d in g.dists]), np.array([d.cov for d in g.dists]), np.array(g.w))


def pack_hmm(h):
    p = [pack_gmm(g) for g in h.gmm_states]
    return np.array([a for a, _, _ in p]), np.array([b for _, b, _ in p]), np.array([c for _, _, c in p])


# ----------------------------------------------------------------------- fixtures
def g1(R):
    for tag, M, D, seed in (("m1d13", 1, 13, 11), ("m8d39", 8, 39, 12)):
        rng = np.random.default_rng(seed)
        means, vars_, w, _ = synth_model(rng, 10, 5, M, D)
        X = np.stack([synth_utt(rng, means, vars_, [i % 10], 1, 1)[0] for i in range(64)])
        X[::7] *= 1.5
        states = [make_gmm(R, means[a, b], vars_[a, b], w[a, b]) for a in range(10) for b in range(5)]
        nll = np.array([[s.evaluate(x) for s in states] for x in X])
        comp = np.array([[s.evaluate(x, False) for s in states] for x in X])
        save("G1_gmm_evaluate_" + tag, means=means.reshape(50, M, D), vars=vars_.reshape(50, M, D),
             w=w.reshape(50, M), X=X, nll=nll, comp=comp)


def g2(R):
    rng = np.random.default_rng(21)
    v1 = rng.normal(size=(32, 39))
    v2 = rng.normal(size=(32, 39))
    var = rng.uniform(0.2, 3.0, size=(32, 39))
    save("G2_mahalanobis", v1=v1, v2=v2, var=var,
         out=np.array([R.mahalanobis(a, b, c) for a, b, c in zip(v1, v2, var)]))


def g3(R):
    for tag, W, M, D, U, seed in (("c1", 10, 1, 13, 4, 31), ("c2", 3, 8, 39, 2, 32)):
        rng = np.random.default_rng(seed)
        means, vars_, w, trans = synth_model(rng, W, 5, M, D)
        hmms = [make_hmm(R, means[i], vars_[i], w[i], trans) for i in range(W)]
        out = dict(means=means, vars=vars_, w=w, trans=trans)
        words = rng.integers(0, W, size=U)
        out["words"] = words
        for u in range(U):
            x = synth_utt(rng, means, vars_, [words[u]], 40, 70)
            out["x%d" % u] = x
            ev = []
            for i, h in enumerate(hmms):
                costs, path = R.decode_hmm_states(x, h.gmm_states, h.transitions)
                out["costs_%d_%d" % (u, i)] = costs
                out["path_%d_%d" % (u, i)] = path
                ev.append(h.evaluate(x))
            out["evaluate_%d" % u] = np.array(ev)
        save("G3_isolated_decode_" + tag, **out)


def g4(R):
    rng = np.random.default_rng(41)
    W, n, M, D = 10, 5, 2, 13
    means, vars_, w, trans = synth_model(rng, W, n, M, D)
    hmms = [make_hmm(R, means[i], vars_[i], w[i], trans) for i in range(W)]
    out = dict(means=means, vars=vars_, w=w, word_trans=trans)
    for K in (1, 2, 3, 7):
        seq, tr, ends = R.build_state_sequences(hmms, [list(range(W))] * K)
        words = rng.integers(0, W, size=K)
        x = synth_utt(rng, means, vars_, words, 14, 22)
        with quiet():
            costs, path = R.decode_hmm_states(x, seq, tr, end_points=[[e, -1] for e in ends])
        ids = {g.id: (wi, si) for wi, h in enumerate(hmms) for si, g in enumerate(h.gmm_states)}
        row_word = np.array([-1 if type(s) is R.NES else ids[s.id][0] for s in seq])
        row_state = np.array([-1 if type(s) is R.NES else ids[s.id][1] for s in seq])
        fi, fj = np.nonzero(~np.isinf(tr))
        out.update({"K%d_x" % K: x, "K%d_words" % K: words, "K%d_costs" % K: costs, "K%d_path" % K: path,
                    "K%d_arc_to" % K: fi, "K%d_arc_from" % K: fj, "K%d_arc_cost" % K: tr[fi, fj],
                    "K%d_row_word" % K: row_word, "K%d_row_state" % K: row_state,
                    "K%d_ends" % K: np.array(ends), "K%d_R" % K: np.array(len(seq)),
                    "K%d_digits" % K: np.array(reference_postprocess(R, path, seq, hmms))})
    # forced-alignment lattice with a repeated word (continuous_speech.py:80)
    labels = [3, 3, 7]
    seq, tr, ends = R.build_state_sequences(hmms, [[l] for l in labels])
    x = synth_utt(rng, means, vars_, labels, 14, 22)
    costs, path = R.decode_hmm_states(x, seq, tr, end_points=[[e, -1] for e in ends])
    fi, fj = np.nonzero(~np.isinf(tr))
    out.update(forced_labels=np.array(labels), forced_x=x, forced_costs=costs, forced_path=path,
               forced_arc_to=fi, forced_arc_from=fj, forced_arc_cost=tr[fi, fj], forced_ends=np.array(ends),
               forced_digits=np.array(reference_postprocess(R, path, seq, hmms)))
    save("G4_lattice_decode", **out)


_MAIN_SNIPPETS = None


def reference_postprocess(R, matched, seq, models):
    """Execute main.py's own post-processing statements (main.py:18-22, 39-52,
    61-67) on a decode result, by compiling those AST nodes of the reference
    file -- main.py cannot be imported (its body needs wav files and pyaudio)."""
    global _MAIN_SNIPPETS
    if _MAIN_SNIPPETS is None:
        tree = ast.parse(open(os.path.join(REF, "main.py")).read())
        body = next(n for n in tree.body if isinstance(n, ast.If)).body
        fdef = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == "split_result")
        loop = [n for n in body if isinstance(n, ast.For)]
        id_map = loop[1]                       # for model_idx in range(len(models)) ...
        main_loop = loop[-1]                   # for x, l in zip(data, labels)
        stmts = main_loop.body[1:5]            # matched = matched[:,0][::-1] ... map(id -> model)
        mod = ast.Module(body=[fdef], type_ignores=[])
        mod2 = ast.Module(body=[id_map], type_ignores=[])
        mod3 = ast.Module(body=stmts, type_ignores=[])
        _MAIN_SNIPPETS = [compile(ast.fix_missing_locations(m), "main.py", "exec") for m in (mod, mod2, mod3)]
    ns = dict(np=np, NES=R.NES, GMM=R.GMM, models=models, gmm_id_modelidx_map={}, seq=seq, matched=matched)
    for code in _MAIN_SNIPPETS:
        exec(code, ns)
    return list(ns["matched"])


def g5(R):
    rng = np.random.default_rng(51)
    means, vars_, w, trans = synth_model(rng, 1, 5, 1, 13)
    x = synth_utt(rng, means, vars_, [0], 45, 45)
    y, var = means[0, :, 0], vars_[0, :, 0]
    out = dict(x=x, y=y, var=var, trans=trans)
    eu = lambda *a: np.linalg.norm(a[0] - a[1])
    with quiet():
        out["costs_euclid"], out["path_euclid"] = R.dtw(x, y, eu, trans)
        out["costs_mahal"], out["path_mahal"] = R.dtw(x, y, R.mahalanobis, trans, var)
        out["costs_beam3"], out["path_beam3"] = R.dtw(x, y, R.mahalanobis, trans, var, beam=3)
        out["costs_beam2"], out["path_beam2"] = R.dtw(x, y, eu, trans, beam=2)
        # skip-transition topology (calc_transition_costs with an empty segment)
        seg_lens = np.array([[9, 9, 0, 9, 9], [8, 8, 0, 8, 8]])
        tr2 = R.calc_transition_costs(2, seg_lens)
        out["trans_skip"] = tr2
        out["seg_lens_skip"] = seg_lens
        out["costs_skip"], out["path_skip"] = R.dtw(x, y, R.mahalanobis, tr2, var)
    save("G5_dtw", **out)


def g6(R):
    rng = np.random.default_rng(61)
    means, vars_, w, trans = synth_model(rng, 1, 5, 2, 6)
    h = make_hmm(R, means[0], vars_[0], w[0], trans)
    out = dict(means=means[0], vars=vars_[0], w=w[0], trans=trans)
    x1 = synth_utt(rng, means, vars_, [0], 1, 1)
    x2 = synth_utt(rng, means, vars_, [0], 2, 2)
    x9 = synth_utt(rng, means, vars_, [0], 9, 9)
    with quiet():
        out["t1_x"] = x1
        out["t1_costs"], p = R.decode_hmm_states(x1, h.gmm_states, h.transitions)
        out["t1_path_shape"] = np.array(p.shape)
        out["t2_x"] = x2
        out["t2_costs"], out["t2_path"] = R.decode_hmm_states(x2, h.gmm_states, h.transitions)
        # duplicated state: rows 3 and 4 share one GMM object and identical arcs -> equal end costs
        states = h.gmm_states[:4] + [h.gmm_states[3]]
        tr = np.full((5, 5), np.inf)
        for i in range(4):
            tr[i, i] = 0.3
        tr[1, 0] = tr[2, 1] = tr[3, 2] = 1.1
        tr[4, 2] = 1.1
        tr[4, 4] = 0.3
        out["tie_x"] = x9
        out["tie_trans"] = tr
        out["tie_costs"], out["tie_path"] = R.decode_hmm_states(x9, states, tr, end_points=[[3, -1], [4, -1]])
        out["tie_costs_rev"], out["tie_path_rev"] = R.decode_hmm_states(x9, states, tr, end_points=[[4, -1], [3, -1]])
        # equal-cost predecessors: first (lowest origin) wins
        tr3 = np.full((3, 3), np.inf)
        tr3[0, 0] = tr3[1, 1] = tr3[2, 2] = 0.5
        tr3[1, 0] = 0.7
        tr3[2, 0] = 0.7
        tr3[2, 1] = 0.5
        st3 = [h.gmm_states[0], h.gmm_states[1], h.gmm_states[1]]
        out["ptie_trans"] = tr3
        out["ptie_costs"], out["ptie_path"] = R.decode_hmm_states(x9, st3, tr3)
    save("G6_decode_edges", **out)


def g7(R):
    rng = np.random.default_rng(71)
    D, M, N = 6, 4, 160
    cm = rng.normal(size=(3, D)) * 2
    data = np.concatenate([cm[i] + rng.normal(size=(N // 4 + 10 * i, D)) * (0.5 + 0.3 * i) for i in range(3)])
    mu0, var0 = data.mean(0), data.var(0)
    init_means = mu0 + rng.normal(size=(M, D))
    init_vars = np.tile(var0, (M, 1))
    init_w = np.array([0.3, 0.2, 0.25, 0.25])
    out = dict(data=data, mu0=mu0, var0=var0, init_means=init_means, init_vars=init_vars, init_w=init_w)
    for k in (2, 3):
        for iters, tag in ((1, "it1"), (10000, "conv")):
            g = R.GMM(mu0.copy(), var0.copy(), M)
            g.update_models(init_means.copy(), init_vars.copy(), init_w.copy())
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                g.em(data, k, max_iteration=iters)
            m, v, w = pack_gmm(g)
            txt = buf.getvalue()
            n_it = int(txt.rsplit("EM converged at iteration:", 1)[1].split()[0]) + 1 if "converged" in txt else iters
            out.update({"k%d_%s_means" % (k, tag): m, "k%d_%s_vars" % (k, tag): v, "k%d_%s_w" % (k, tag): w,
                        "k%d_%s_iters" % (k, tag): np.array(n_it),
                        "k%d_%s_mu_old" % (k, tag): g.mu_old, "k%d_%s_sigma_old" % (k, tag): g.sigma_old,
                        "k%d_%s_w_old" % (k, tag): g.w_old})
    save("G7_gmm_em", **out)


def g16(R):
    """GMM.em with frames whose weighted densities underflow in the reference's LINEAR domain (hmm_state.py:42-43,
    128-133): three frames far from every component (all-zero responsibility row, row sum 0 -> 1e-5) and three
    frames that only one component still resolves."""
    rng = np.random.default_rng(716)
    D, M, N = 6, 4, 120
    cm = rng.normal(size=(2, D)) * 2
    data = np.concatenate([cm[i] + rng.normal(size=(N // 2, D)) * (0.6 + 0.3 * i) for i in range(2)])
    far = data.mean(0) + 90.0 * np.sqrt(data.var(0)) * np.array([1, -1, 1, 1, -1, 1.0])      # every component -> 0
    data = np.concatenate([data[:40], far[None] * np.array([[1.0], [1.1], [0.9]]), data[40:]])
    mu0, var0 = data[:40].mean(0), data[:40].var(0)
    init_means = np.concatenate([cm + rng.normal(size=(2, D)) * 0.3, mu0 + rng.normal(size=(2, D))])
    init_vars = np.tile(var0, (M, 1))
    init_vars[1] *= 30.0                                                                      # a wide component
    half = init_means[0] + 38.0 * np.sqrt(init_vars[0])                                       # only the wide one resolves
    data = np.concatenate([data, half[None] + rng.normal(size=(3, D)) * 0.1])
    init_w = np.array([0.3, 0.2, 0.25, 0.25])
    out = dict(data=data, mu0=mu0, var0=var0, init_means=init_means, init_vars=init_vars, init_w=init_w)
    for k in (2, 3):
        g = R.GMM(mu0.copy(), var0.copy(), M)
        g.update_models(init_means.copy(), init_vars.copy(), init_w.copy())
        p = np.array([g.evaluate(x, return_neg_log_likelihood=False)[:k] for x in data])
        out["k%d_p0" % k] = p                                                                  # first E-step, linear domain
        for iters, tag in ((1, "it1"), (3, "it3")):
            g = R.GMM(mu0.copy(), var0.copy(), M)
            g.update_models(init_means.copy(), init_vars.copy(), init_w.copy())
            with quiet():
                g.em(data, k, max_iteration=iters)
            m, v, w = pack_gmm(g)
            out.update({"k%d_%s_means" % (k, tag): m, "k%d_%s_vars" % (k, tag): v, "k%d_%s_w" % (k, tag): w})
    save("G16_gmm_em_underflow", **out)


def g8(R):
    rng = np.random.default_rng(81)
    D = 8
    cm = rng.normal(size=(4, D)) * 1.5
    data = np.concatenate([cm[i] + rng.normal(size=(50 + 7 * i, D)) for i in range(4)])
    data = data[rng.permutation(len(data))]
    out = dict(data=data)
    for k, dist, tag in ((2, "mahalanobis", "k2m"), (4, "mahalanobis", "k4m"), (4, "euclid", "k4e")):
        c0 = data[:k] * 1.0
        np.random.seed(0)
        if dist == "euclid":
            cl, ce, cov = R.kmeans(data, k, c0.copy())
        else:
            cl, ce, cov = R.kmeans(data, k, c0.copy(), dist_fun=R.mahalanobis)
        out.update({tag + "_c0": c0, tag + "_clusters": cl, tag + "_centroids": ce, tag + "_cov": cov})
    save("G8_kmeans", **out)


def _train_set(seed, n_utts, D, n_states=5, tmin=30, tmax=45, M=2):
    rng = np.random.default_rng(seed)
    means, vars_, w, _ = synth_model(rng, 1, n_states, M, D)
    means *= 2.0
    return [synth_utt(rng, means, vars_, [0], tmin, tmax) for _ in range(n_utts)]


def g9(R):
    ys = _train_set(91, 6, 6)
    with quiet():
        h = R.HMM(5).fit([y.copy() for y in ys], 1, use_gmm=False)
    out = {"y%d" % i: y for i, y in enumerate(ys)}
    out.update(n=np.array(len(ys)), mu=h.mu, sigma=h.sigma, transitions=h.transitions,
               seg_lens=np.array([len(s) for s in h.segments]))
    for i, s in enumerate(h.segments):
        out["seg%d" % i] = s
    # the helper functions on their own
    path = np.array([[4, 9], [4, 8], [3, 7], [2, 6], [2, 5], [2, 4], [1, 3], [0, 2], [0, 1], [0, 0]])
    out["gsp_path"] = path
    out["gsp_out"] = R.get_segments_from_path(path, 5)
    lens = np.array([[6, 7, 6, 7, 9], [8, 8, 8, 8, 11]])
    out["ctc_lens"] = lens
    out["ctc_out"] = R.calc_transition_costs(2, lens)
    save("G9_hmm_fit_single", **out)


def g10(R):
    ys = _train_set(101, 6, 6, M=3)
    out = {"y%d" % i: y for i, y in enumerate(ys)}
    out["n"] = np.array(len(ys))
    for ng, em in ((4, True), (8, True), (4, False)):
        tag = "g%d_%s" % (ng, "em" if em else "km")
        np.random.seed(5)
        with quiet():
            h = R.HMM(5).fit([y.copy() for y in ys], ng, use_gmm=True, use_em=em)
        m, v, w = pack_hmm(h)
        out.update({tag + "_mu": h.mu, tag + "_sigma": h.sigma, tag + "_transitions": h.transitions,
                    tag + "_means": m, tag + "_vars": v, tag + "_w": w,
                    tag + "_seg_lens": np.array([len(s) for s in h.segments]),
                    tag + "_evaluate": np.array([h.evaluate(y) for y in ys])})
    save("G10_hmm_fit_gmm", **out)


def g11(R):
    rng = np.random.default_rng(111)
    W, n, D, ng = 3, 5, 5, 4
    means, vars_, w, _ = synth_model(rng, W, n, 2, D)
    means *= 2.5
    iso = [[synth_utt(rng, means, vars_, [wd], 40, 55) for _ in range(8)] for wd in range(W)]
    np.random.seed(7)
    with quiet():
        hmms = [R.HMM(n).fit([y.copy() for y in iso[wd]], ng) for wd in range(W)]
    for h in hmms:
        for s in h.gmm_states:
            s.parent = h
    label_seqs = [[0, 1], [2, 0, 1], [1, 1], [2, 2, 0], [0, 2], [1, 0, 2], [2, 1], [0, 0, 1]]
    data = [synth_utt(rng, means, vars_, ls, 28, 40) for ls in label_seqs]
    out = dict(n_utts=np.array(len(data)), n_words=np.array(W))
    for i, (x, ls) in enumerate(zip(data, label_seqs)):
        out["x%d" % i] = x
        out["labels%d" % i] = np.array(ls)
    for wi, h in enumerate(hmms):
        m, v, ww = pack_hmm(h)
        out.update({"init%d_means" % wi: m, "init%d_vars" % wi: v, "init%d_w" % wi: ww,
                    "init%d_transitions" % wi: h.transitions,
                    "init%d_mu_old" % wi: np.array([g.mu_old for g in h.gmm_states]),
                    "init%d_sigma_old" % wi: np.array([g.sigma_old for g in h.gmm_states]),
                    "init%d_w_old" % wi: np.array([g.w_old for g in h.gmm_states])})
    import copy
    import pickle
    for iters in (1, 3):
        models = copy.deepcopy(hmms)
        np.random.seed(9)
        with tempfile.TemporaryDirectory() as tmp, quiet():
            R.continuous_train([x.copy() for x in data], models, label_seqs, tmp, n_gaussians=ng,
                               n_segments=n, max_iteration=iters)
            res = [pickle.load(open(os.path.join(tmp, "%d.pkl" % i), "rb")) for i in range(W)]
        for wi, h in enumerate(res):
            m, v, ww = pack_hmm(h)
            out.update({"it%d_%d_means" % (iters, wi): m, "it%d_%d_vars" % (iters, wi): v,
                        "it%d_%d_w" % (iters, wi): ww, "it%d_%d_transitions" % (iters, wi): h.transitions})
    save("G11_continuous_train", **out)


def g17(R):
    """continuous_train once more, at a shape closer to the real task: 13-dim features, 8 mixtures (three binary splits),
    4 words of 3 states, label strings of 2-4 words with repeats; 1 and 2 outer iterations."""
    rng = np.random.default_rng(1717)
    W, n, D, ng = 4, 3, 13, 8
    means, vars_, w, _ = synth_model(rng, W, n, 2, D)
    means *= 2.0
    iso = [[synth_utt(rng, means, vars_, [wd], 40, 60) for _ in range(10)] for wd in range(W)]
    np.random.seed(17)
    with quiet():
        hmms = [R.HMM(n).fit([y.copy() for y in iso[wd]], ng) for wd in range(W)]
    for h in hmms:
        for s in h.gmm_states:
            s.parent = h
    label_seqs = [[0, 1, 2], [3, 3], [2, 0, 1, 3], [1, 1, 0], [3, 2], [0, 3, 1], [2, 2, 2], [1, 0], [3, 1, 2, 0], [0, 0, 3],
                  [2, 3, 1], [1, 2]]
    data = [synth_utt(rng, means, vars_, ls, 40, 60) for ls in label_seqs]
    out = dict(n_utts=np.array(len(data)), n_words=np.array(W), n_gaussians=np.array(ng), n_segments=np.array(n))
    for i, (x, ls) in enumerate(zip(data, label_seqs)):
        out["x%d" % i] = x
        out["labels%d" % i] = np.array(ls)
    for wi, h in enumerate(hmms):
        m, v, ww = pack_hmm(h)
        out.update({"init%d_means" % wi: m, "init%d_vars" % wi: v, "init%d_w" % wi: ww,
                    "init%d_transitions" % wi: h.transitions,
                    "init%d_mu_old" % wi: np.array([g.mu_old for g in h.gmm_states]),
                    "init%d_sigma_old" % wi: np.array([g.sigma_old for g in h.gmm_states]),
                    "init%d_w_old" % wi: np.array([g.w_old for g in h.gmm_states])})
    import copy
    import pickle
    for iters in (1, 2):
        models = copy.deepcopy(hmms)
        np.random.seed(19)
        with tempfile.TemporaryDirectory() as tmp, quiet():
            R.continuous_train([x.copy() for x in data], models, label_seqs, tmp, n_gaussians=ng,
                               n_segments=n, max_iteration=iters)
            res = [pickle.load(open(os.path.join(tmp, "%d.pkl" % i), "rb")) for i in range(W)]
        for wi, h in enumerate(res):
            m, v, ww = pack_hmm(h)
            out.update({"it%d_%d_means" % (iters, wi): m, "it%d_%d_vars" % (iters, wi): v,
                        "it%d_%d_w" % (iters, wi): ww, "it%d_%d_transitions" % (iters, wi): h.transitions})
    save("G17_continuous_train_8mix", **out)


ALL = dict(G1=g1, G2=g2, G3=g3, G4=g4, G5=g5, G6=g6, G7=g7, G8=g8, G9=g9, G10=g10, G11=g11, G16=g16, G17=g17)


def g12(R):
    """N2: a pickle written by the REFERENCE's classes (data: parameters only) --
    the mirror package must load it under the same module path and score identically."""
    import pickle
    rng = np.random.default_rng(121)
    means, vars_, w, trans = synth_model(rng, 2, 3, 2, 4)
    hmms = [make_hmm(R, means[i], vars_[i], w[i], trans) for i in range(2)]
    x = synth_utt(rng, means, vars_, [1], 12, 12)
    blob = pickle.dumps(hmms, protocol=2)
    with quiet():
        ev = np.array([h.evaluate(x) for h in hmms])
    save("G12_reference_pickle", pickle=np.frombuffer(blob, dtype=np.uint8), x=x, evaluate=ev,
         ids=np.array([[str(g.id) for g in h.gmm_states] for h in hmms]))


ALL["G12"] = g12


def g13(R):
    """N3: the feature stacking in front of the hot path.  sr/core.py cannot be imported (pyaudio,
    python_speech_features), so its `delta_feature` is compiled from its AST; `standardize` comes from
    the importable sr.feature."""
    tree = ast.parse(open(os.path.join(REF, "sr", "core.py")).read())
    fdef = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "delta_feature")
    ns = dict(np=np)
    exec(compile(ast.fix_missing_locations(ast.Module(body=[fdef], type_ignores=[])), "core.py", "exec"), ns)
    delta_feature = ns["delta_feature"]
    standardize = importlib.import_module("sr.feature").standardize
    rng = np.random.default_rng(131)
    out = {}
    for i, T in enumerate((2, 3, 57, 130)):
        ceps = rng.normal(size=(T, 13)) * rng.uniform(0.5, 20.0, size=13) + rng.normal(size=13) * 5
        df = delta_feature(ceps)
        ddf = delta_feature(df)
        feats = standardize(np.concatenate([ceps, df, ddf], axis=1))
        out.update({"ceps%d" % i: ceps, "delta%d" % i: df, "ddelta%d" % i: ddf, "feats%d" % i: feats})
    out["n"] = np.array(4)
    save("G13_feature_stack", **out)


ALL["G13"] = g13

def g14(R):
    """N4: word LOOP grammar run through the reference's OWN decode_hmm_states.  The reference has no loop
    grammar builder, but its DP reads same-column origins only from rows already visited in the column
    (decode.py:97-98,109-111), so a loop is expressible by row order: [start NES | states 1..n-1 of every
    word | loop NES | state 0 of every word].  Stored: the graph, the reference's costs / path / digits on
    it, and the reference's K-layer end costs for K = 1..6 (the loop cost must equal their minimum)."""
    rng = np.random.default_rng(141)
    W, n, M, D = 4, 3, 2, 6
    means, vars_, w, trans = synth_model(rng, W, n, M, D)
    hmms = [make_hmm(R, means[i], vars_[i], w[i], trans) for i in range(W)]
    Rr = 2 + W * n
    loop_row = 1 + W * (n - 1)
    row_of = np.empty((W, n), dtype=np.int64)
    for wd in range(W):
        row_of[wd, 1:] = 1 + wd * (n - 1) + np.arange(n - 1)
        row_of[wd, 0] = loop_row + 1 + wd
    ids = {g.id: (wi, si) for wi, h in enumerate(hmms) for si, g in enumerate(h.gmm_states)}
    out = dict(means=means, vars=vars_, w=w, word_trans=trans, n_utts=np.array(3), Kmax=np.array(6))
    for pen_i, penalty in enumerate((0.0, 2.5)):
        tr = np.full((Rr, Rr), np.inf)
        seq = [None] * Rr
        seq[0], seq[loop_row] = R.NES(), R.NES()
        ends = []
        for wd, h in enumerate(hmms):
            rows = row_of[wd]
            for i in range(n):
                seq[rows[i]] = h.gmm_states[i]
            tr[np.ix_(rows, rows)] = h.transitions
            tr[rows[0], 0] = 0
            tr[rows[0], loop_row] = penalty
            tr[loop_row, rows[n - 1]] = 0
            ends.append(int(rows[n - 1]))
        fi, fj = np.nonzero(~np.isinf(tr))
        row_word = np.array([-1 if type(s) is R.NES else ids[s.id][0] for s in seq])
        row_state = np.array([-1 if type(s) is R.NES else ids[s.id][1] for s in seq])
        pp = "p%d_" % pen_i
        out.update({pp + "penalty": np.array(penalty), pp + "arc_to": fi, pp + "arc_from": fj, pp + "arc_cost": tr[fi, fj],
                    pp + "row_word": row_word, pp + "row_state": row_state, pp + "ends": np.array(ends)})
        urng = np.random.default_rng(1410)
        for u, nw in enumerate((1, 3, 5)):
            words = urng.integers(0, W, size=nw)
            x = synth_utt(urng, means, vars_, words, 7, 11)
            with quiet():
                costs, path = R.decode_hmm_states(x, seq, tr, end_points=[[e, -1] for e in ends])
            out.update({pp + "x%d" % u: x, pp + "words%d" % u: words, pp + "costs%d" % u: costs, pp + "path%d" % u: path,
                        pp + "digits%d" % u: np.array(reference_postprocess(R, path, seq, hmms))})
            if pen_i == 0:   # the reference's exactly-K-words lattices on the same utterance
                kc = []
                for K in range(1, 7):
                    sk, tk, ek = R.build_state_sequences(hmms, [list(range(W))] * K)
                    with quiet():
                        ck, _ = R.decode_hmm_states(x, sk, tk, end_points=[[e, -1] for e in ek])
                    kc.append(min(ck[e, -1] for e in ek))
                out["layer_costs%d" % u] = np.array(kc)
    save("G14_loop_grammar", **out)


ALL["G14"] = g14

def g15(R):
    """N3, first half: the reference's own mfcc_features (sr/feature/feature.py:43-82) on int16 wav files written
    to a temporary directory: lengths around the frame step / frame length edges, tones + noise, one silent file
    (log10 of eps) and one 8 kHz file (200-sample frames padded to 256, rfft zero-extends to 512)."""
    import tempfile
    from scipy.io import wavfile
    mfcc_features = importlib.import_module("sr.feature").mfcc_features
    rng = np.random.default_rng(151)
    cases = [(16000, 1), (16000, 159), (16000, 160), (16000, 161), (16000, 400), (16000, 4001), (16000, 16000),
             (8000, 3000)]
    out = dict(n=np.array(len(cases) + 1))
    with tempfile.TemporaryDirectory() as td:
        for i, (sr_, n) in enumerate(cases):
            t = np.arange(n) / sr_
            sig = 6000 * np.sin(2 * np.pi * 440 * t) + 3000 * np.sin(2 * np.pi * 1870 * t + 1.0) + 800 * rng.normal(size=n)
            sig = np.clip(np.round(sig), -32768, 32767).astype(np.int16)
            path = os.path.join(td, "c%d.wav" % i)
            wavfile.write(path, sr_, sig)
            fb, mf = mfcc_features(path)
            out.update({"rate%d" % i: np.array(sr_), "signal%d" % i: sig, "fbank%d" % i: fb, "mfcc%d" % i: mf})
        i = len(cases)
        sig = np.zeros(800, dtype=np.int16)
        path = os.path.join(td, "silent.wav")
        wavfile.write(path, 16000, sig)
        fb, mf = mfcc_features(path)
        out.update({"rate%d" % i: np.array(16000), "signal%d" % i: sig, "fbank%d" % i: fb, "mfcc%d" % i: mf})
    save("G15_mfcc", **out)


ALL["G15"] = g15

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    R = load_reference()
    names = [s for s in args.only.split(",") if s] or list(ALL)
    for nm in names:
        ALL[nm](R)
