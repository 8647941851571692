# -*- coding: utf-8 -*-
"""bench.py's C2_train_words leg alone (isolated-word training of 10 word models in lock-step), for rocprofv3 / cProfile."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-recognition_amd"))
import bench
from sr.recognition import _hip

ctx = _hip.default_context()
for _ in range(int(os.environ.get("REPS", "3"))):
    r = bench._train_words_config(ctx)
print(json.dumps(r))
