# -*- coding: utf-8 -*-
"""cProfile of bench.py's C2_train_words leg (batch.train_words on 10 words x 200 templates): where the host spends it."""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-recognition_amd"))
import bench
from sr.recognition import _hip
ctx = _hip.default_context()
bench._train_words_config(ctx)
pr = cProfile.Profile()
pr.enable()
r = bench._train_words_config(ctx)
pr.disable()
print(r)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:9000])
