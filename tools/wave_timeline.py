# -*- coding: utf-8 -*-
"""Per-wave phase timeline of the likelihood kernel (diagnostic build -DGH_MF_TIMING, see
tools/variant_bench.sh):  GMMHMM_LIB=tools/bin/libgmmhmm_timing.so python tools/wave_timeline.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "speech-recognition_amd"))
import bench
from sr.recognition import _hip

# shape: WT_SHAPE="utterances:mixtures:dim" (default 10000:8:39 = configs[1]; "100000:1:13" = configs[0] x 1000)
_shape = [int(v) for v in os.environ.get("WT_SHAPE", "10000:8:39").split(":")]
w = bench.synth_workload(1001, _shape[0], M=_shape[1], D=_shape[2])
means, vars_, wts = w["means"], w["vars"], w["w"]
S = means.shape[0] * means.shape[1]
ctx = _hip.default_context(0)
gmm = _hip.PackedGMM(ctx, means.reshape(S, *means.shape[2:]), vars_.reshape(S, *vars_.shape[2:]), wts.reshape(S, -1))
_dt = np.float32 if os.environ.get("WT_DTYPE", "f64") == "f32" else np.float64   # WT_DTYPE=f32: the fp32 kernel
b = _hip.Batch(ctx, feats=w["X"], offsets=w["off"], dtype=_dt)
for _ in range(int(os.environ.get('WARM', 3))):
    b.loglik(gmm, fetch=False)
nll = b.loglik(gmm, fetch=True)
rows = nll[::32]
tk = rows[:, :5]
wall0, walld = rows[:, 5], rows[:, 6]
print("blocks", len(rows))
names = ["prologue+Bbuild", "ring/C init", "tile loop", "flush"]
d = np.diff(tk, axis=1)
for i, n in enumerate(names):
    print("%-16s mean %9.0f  p50 %9.0f  p95 %9.0f  cycles" % (n, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 95)))
tot = tk[:, 4]
print("wave total cycles mean %.0f; wall (100 MHz ticks) mean %.1f -> clock %.3f GHz" % (tot.mean(), walld.mean(), tot.sum() / walld.sum() * 0.1))
span = (wall0 + walld).max() - wall0.min()
print("kernel span %.1f us; sum of wave time %.1f us; avg resident waves %.1f (per SIMD %.2f)" % (
    span / 100, walld.sum() / 100, walld.sum() / span, walld.sum() / span / 1024))
order = np.argsort(wall0)
st = wall0[order] - wall0.min()
print("start times (us) of waves #0, 1000, 2000, 2304, 3000, 10000, last:", [round(st[min(i, len(st) - 1)] / 100, 1) for i in (0, 1000, 2000, 2304, 3000, 10000, len(st) - 1)])
