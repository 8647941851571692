#!/usr/bin/env python3
"""continuous_train (embedded Viterbi training, continuous_speech.py:56-179) on synthetic digit strings: wall time per
outer iteration and where the host spends it (cProfile, top entries).
usage: time_ctrain.py [utterances] [words per utterance] [outer iterations]"""
import cProfile, io, os, pstats, sys, tempfile, time, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "speech-recognition_amd")):
    sys.path.insert(0, p)
import numpy as np
if os.environ.get('PYSWITCH'):
    sys.setswitchinterval(float(os.environ['PYSWITCH']))
import bench
from sr.recognition import continuous_speech as cs
from sr.recognition.model_io import models_from_arrays

U = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 7
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2
wl = bench.synth_workload(1003, U * K)
W, n, M, D = wl["W"], wl["n"], wl["M"], wl["D"]
iso = [wl["X"][wl["off"][u]:wl["off"][u + 1]] for u in range(U * K)]
data = [np.concatenate(iso[i * K:(i + 1) * K]) for i in range(U)]
labels = [[int(w) for w in wl["words"][i * K:(i + 1) * K]] for i in range(U)]
rng = np.random.default_rng(0)
means0 = wl["means"] + 0.3 * rng.normal(size=wl["means"].shape)
models = models_from_arrays(means0, wl["vars"], wl["w"], [wl["trans"]] * W, mu=means0[:, :, 0], sigma=wl["vars"][:, :, 0])
print("%d utterances x %d words, %d frames, %d-dim, %d mixtures" % (U, K, sum(len(x) for x in data), D, M), flush=True)
np.random.seed(0)
out = tempfile.mkdtemp()
class Stamps(io.StringIO):
    """stdout of continuous_train: remembers when every outer iteration started"""
    def __init__(self):
        super().__init__()
        self.t = []
    def write(self, text):
        if text.startswith("Continuous training iteration:"):
            self.t.append(time.perf_counter())
        return len(text)
pr = cProfile.Profile()
stamps = Stamps()
profile = os.environ.get("CTRAIN_PROFILE", "1") != "0"
t0 = time.perf_counter()
with contextlib.redirect_stdout(stamps):
    if profile:
        pr.enable()
    cs.continuous_train(data, models, labels, out, n_gaussians=M, n_segments=n, max_iteration=iters)
    if profile:
        pr.disable()
t_end = time.perf_counter()
dt = t_end - t0
per = np.diff(stamps.t + [t_end])
print("%.2f s for %d outer iterations (%.2f s each)" % (dt, iters, dt / iters))
print("per outer iteration [ms]: " + " ".join("%.0f" % (1e3 * x) for x in per) + "   (the first ones grow the scratch arenas)")
print("steady state: %.1f ms per outer iteration (median of the last %d)" % (1e3 * float(np.median(per[len(per) // 2:])), len(per) - len(per) // 2))
if not profile:
    sys.exit(0)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000])
