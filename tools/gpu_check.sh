#!/bin/bash
# scratch check between evidence runs: train_words tests, stage times, the threaded copy against numpy on this box
mkdir -p gpurun_out/r05e
timeout -k 10 600 python -m pytest tests/test_gpu_train_words.py tests/test_gpu_api.py -x -q -m gpu > gpurun_out/r05e/tw_tests.log 2>&1 || { tail -20 gpurun_out/r05e/tw_tests.log; exit 1; }
tail -2 gpurun_out/r05e/tw_tests.log
REPS=5 timeout -k 10 300 python3 tools/prof_train_words.py > gpurun_out/r05e/train_words_stages.txt 2>&1 || { tail gpurun_out/r05e/train_words_stages.txt; exit 1; }
grep "train_words \[ms\]\|^call \|without" gpurun_out/r05e/train_words_stages.txt
timeout -k 10 120 python3 - <<'PY'
import sys, time
sys.path.insert(0, "speech-recognition_amd")
import numpy as np
from sr.recognition import _hostcopy
rng = np.random.default_rng(0)
lens = rng.integers(50, 150, size=2000)
parts = [rng.normal(size=(int(n), 39)) for n in lens]
out = np.empty((int(lens.sum()), 39))
for thr in (1, 2, 4, 8, 16):
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); _hostcopy.concat_rows(parts, out, thr); ts.append((time.perf_counter() - t0) * 1e3)
    print("concat_rows, %d threads: %.2f ms" % (thr, min(ts)))
ts = []
for _ in range(7):
    t0 = time.perf_counter(); np.concatenate(parts, out=out); ts.append((time.perf_counter() - t0) * 1e3)
print("np.concatenate(out=): %.2f ms" % min(ts))
PY
