#!/usr/bin/env python3
"""The fp64 likelihood kernel with its log-sum-exp exponentials in fp64 (default) and in fp32 (gh_ctx_set_compat bit 1):
kernel time (HIP events around the launches), max |delta nll|, and how many Viterbi decodes change -- configs[1]
(10 000 isolated-word utterances: recognised word and state path) and configs[4] (distinct seven-word utterances through the
K = 7 lattice and the loop grammar: state path).
usage: time_lse.py [c2_utterances] [c5_distinct_utterances]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-recognition_amd")):
    sys.path.insert(0, p)
import numpy as np
import bench
from sr.recognition import _hip
from sr.recognition.continuous_speech import packed_lattice, packed_loop_lattice

U2 = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
U5 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
ctx = _hip.Context(0)
out = {}


def kernel_ms(b, gmm, reps=20, ramp=0.5):
    b.loglik(gmm, fetch=False); ctx.sync()
    t_r = time.perf_counter()
    while time.perf_counter() - t_r < ramp:
        b.loglik(gmm, fetch=False); ctx.sync()      # (launches are asynchronous: without the sync the ramp queues seconds of work)
    ctx.sync()
    e0, e1 = ctx.new_event(), ctx.new_event()
    ctx.record(e0)
    for _ in range(reps):
        b.loglik(gmm, fetch=False)
    ctx.record(e1)
    ctx.sync()
    return ctx.elapsed_ms(e0, e1) / reps


# ---- configs[1]
wl = bench.synth_workload(1002, U2)
W, n, M, D = wl["W"], wl["n"], wl["M"], wl["D"]
S = W * n
gmm = _hip.PackedGMM(ctx, wl["means"].reshape(S, M, D), wl["vars"].reshape(S, M, D), wl["w"].reshape(S, M))
b = _hip.Batch(ctx, feats=wl["X"], offsets=wl["off"])
lat = _hip.Lattices(ctx, [bench.stacked_graph(W, n, wl["trans"])])
res = {}
for name, fe in (("fp64", False), ("f32exp", True)):
    ctx.set_compat(underflow=True, lse_f32=fe)
    ms = kernel_ms(b, gmm)
    nll = b.loglik(gmm)
    r = lat.viterbi(b, want_path=True)
    res[name] = dict(ms=ms, nll=nll.copy(), words=r["best_end"].copy(), paths=r["paths"])
flops = 2.0 * 2 * D * S * M * b.N
d = np.abs(res["fp64"]["nll"] - res["f32exp"]["nll"])
out["C2"] = dict(utterances=U2, frames=int(b.N), kernel_ms_fp64=res["fp64"]["ms"], kernel_ms_f32exp=res["f32exp"]["ms"],
                 frac_fp64=flops / (res["fp64"]["ms"] * 1e-3) / 78.6e12, frac_f32exp=flops / (res["f32exp"]["ms"] * 1e-3) / 78.6e12,
                 max_abs_delta_nll=float(d.max()), max_rel_delta_nll=float((d / np.abs(res["fp64"]["nll"])).max()),
                 word_mismatch_rate=float(np.mean(res["fp64"]["words"] != res["f32exp"]["words"])),
                 path_mismatch_rate=float(np.mean([not np.array_equal(x, y) for x, y in zip(res["fp64"]["paths"], res["f32exp"]["paths"])])))
print(json.dumps(out["C2"]), flush=True)
b.close(); lat.close()

# ---- configs[4]: distinct utterances of K = 7 words
K = 7
rng = np.random.default_rng(1005)
wl5 = bench.synth_workload(1005, 1, W=W, n=n, M=M, D=D)
means, vars_, trans = wl5["means"], wl5["vars"], wl5["trans"]
words = rng.integers(0, W, size=(U5, K))
Tw = rng.integers(30, 61, size=(U5, K))
seg_len = Tw.reshape(-1)
seg_off = np.concatenate([[0], np.cumsum(seg_len)])
Nb = int(seg_off[-1])
seg = np.repeat(np.arange(len(seg_len)), seg_len)
t = np.arange(Nb) - seg_off[seg]
st = np.minimum(t * n // seg_len[seg], n - 1)
idx = (words.reshape(-1)[seg] * n + st) * M + rng.integers(0, M, size=Nb)
X = means.reshape(-1, D)[idx] + np.sqrt(vars_).reshape(-1, D)[idx] * rng.standard_normal((Nb, D))
off = np.concatenate([[0], np.cumsum(Tw.sum(axis=1))]).astype(np.int64)
gmm5 = _hip.PackedGMM(ctx, means.reshape(S, M, D), vars_.reshape(S, M, D), wl5["w"].reshape(S, M))
b5 = _hip.Batch(ctx, feats=X, offsets=off)
for key, graph in (("C5_K7_lattice", packed_lattice([trans] * W, n, [list(range(W))] * K)[0]), ("C5_loop_grammar", packed_loop_lattice([trans] * W, n)[0])):
    lat5 = _hip.Lattices(ctx, [graph])
    got = {}
    for name, fe in (("fp64", False), ("f32exp", True)):
        ctx.set_compat(underflow=True, lse_f32=fe)
        b5.loglik(gmm5, fetch=False)
        got[name] = lat5.viterbi(b5, want_path=True)["paths"]
    out[key] = dict(distinct_utterances=U5, frames=int(b5.N),
                    path_mismatch_rate=float(np.mean([not np.array_equal(x, y) for x, y in zip(got["fp64"], got["f32exp"])])))
    print(key, json.dumps(out[key]), flush=True)
    lat5.close()
ctx.set_compat(underflow=True, lse_f32=False)
print(json.dumps(out))
