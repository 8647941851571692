#!/usr/bin/env python3
"""Time one soft-EM iteration (configs[2] shape: C2 model, per-rank shard) on one GPU.
usage: time_em.py [utterances] [words per transcript, default 1 = isolated words]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-recognition_amd")):
    sys.path.insert(0, p)
import numpy as np
import bench
from sr.recognition.train import BaumWelchTrainer
from sr.recognition import _hip

U = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
wl = bench.synth_workload(1003, U)
W, n, M, D = wl["W"], wl["n"], wl["M"], wl["D"]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1
data = [wl["X"][wl["off"][u]:wl["off"][u + 1]] for u in range(U)]
labels = [[int(w)] for w in wl["words"]]
if K > 1:   # transcripts of K words: K isolated-word utterances back to back
    data = [np.concatenate(data[i:i + K]) for i in range(0, U - K + 1, K)]
    labels = [[l[0] for l in labels[i:i + K]] for i in range(0, U - K + 1, K)]
    U = len(data)
rng = np.random.default_rng(0)
means0 = wl["means"] + 0.3 * rng.normal(size=wl["means"].shape)   # perturbed start
t0 = time.perf_counter()
tr = BaumWelchTrainer(means0, wl["vars"], wl["w"], [wl["trans"]] * W, data, labels, occ_floor=float(os.environ.get("EM_OCC_FLOOR", "0")))
print("setup s", time.perf_counter() - t0, flush=True)
ctx = tr.ctx
hist = []
for it in range(4):
    t0 = time.perf_counter()
    ll = tr.iteration()
    dt = time.perf_counter() - t0
    hist.append(ll)
    print("iteration %d: %.1f ms  loglik %.6e" % (it, dt * 1e3, ll), flush=True)
if tr.session is not None:   # the device-resident iteration: 20 more, enqueued back to back
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(20):
        tr.iteration(sync=False)
    tr.drain(); dt = time.perf_counter() - t0
    print("device-resident session: %.3f ms per iteration (20 enqueued)" % (dt / 20 * 1e3), flush=True)
# split of the E-step (call by call)
tr._ensure_call_path()
gmm = _hip.PackedGMM(ctx, tr.means, tr.vars, tr.weights)
def t(fn, reps=3):
    fn(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync(); return (time.perf_counter() - t0) / reps * 1e3
print(json.dumps(dict(utts=U, frames=int(tr.batch.N),
                      loglik_full_ms=t(lambda: tr.batch.loglik(gmm, fetch=False)),
                      loglik_own_states_ms=t(lambda: tr.batch.loglik(gmm, fetch=False, state_sets=tr.state_sets)),
                      fwdbwd_ms=t(lambda: tr.lat.forward_backward(tr.batch, utt_lattice=tr.utt_graph, want_occ=True, fetch_occ=False, want_self_xi=True)),
                      bw_stats_ms=t(lambda: tr.batch.bw_accumulate(gmm)),
                      monotone=bool(all(b >= a - 1e-7 * abs(a) for a, b in zip(hist, hist[1:]))))))
