#!/usr/bin/env python3
"""Host-side split of one soft-EM iteration (configs[2] per-GPU shard): where the wall time goes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-recognition_amd")):
    sys.path.insert(0, p)
import numpy as np
import bench
from sr.recognition.train import BaumWelchTrainer
from sr.recognition import _hip
from sr.recognition.parallel import m_step
U = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
wl = bench.synth_workload(1003, U)
W = wl["W"]
data = [wl["X"][wl["off"][u]:wl["off"][u + 1]] for u in range(U)]
labels = [[int(w)] for w in wl["words"]]
means0 = wl["means"] + 0.3 * np.random.default_rng(0).normal(size=wl["means"].shape)
tr = BaumWelchTrainer(means0, wl["vars"], wl["w"], [wl["trans"]] * W, data, labels)
for _ in range(3):
    tr.iteration()
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
R = 10
for _ in range(R):
    t0 = time.perf_counter(); gmm = _hip.PackedGMM(tr.ctx, tr.means, tr.vars, tr.weights); tick("PackedGMM", t0)
    t0 = time.perf_counter(); tr.batch.loglik(gmm, fetch=False, state_sets=tr.state_sets); tick("loglik_subset", t0)
    t0 = time.perf_counter(); r = tr.lat.forward_backward(tr.batch, utt_lattice=tr.utt_graph, want_occ=True, fetch_occ=False, want_self_xi=True); tick("forward_backward", t0)
    t0 = time.perf_counter(); stats = tr.batch.bw_accumulate(gmm); tick("bw_accumulate", t0)
    t0 = time.perf_counter(); gmm.close(); tick("gmm.close", t0)
    t0 = time.perf_counter()
    counts = stats[:, :, 0].sum(axis=1); mu, sg, w = m_step(stats, counts, tr.means); tick("m_step", t0)
    t0 = time.perf_counter(); tr._update_transitions(counts, r["self_xi"]); tick("transitions+lattices", t0)
t0 = time.perf_counter()
for _ in range(R):
    tr.iteration()
tick("iteration (whole)", t0)
for k, v in T.items():
    print("%-24s %.3f ms" % (k, v / R * 1e3))
