# -*- coding: utf-8 -*-
"""The randomised sweeps of tools/ as tests, a few dozen trials each (the evidence runs take hundreds: profiles/r05e_stress_*.txt):

  * stress_refit.py        streaming refit kernels (ordinary + TAIL launches, eight-wave row sums) against the tile kernels on
                           random shapes, the oracle as the arbiter where they differ;
  * stress_decode.py       layer / wide layer / loop / wide loop / sequence Viterbi kernels against the row-per-lane lean kernel
                           (costs bitwise, ends, paths, labels);
  * stress_train_words.py  `train_words` against the word-after-word `HMM.fit` loop under the same numpy seed;
  * stress_ctrain.py       `continuous_train`'s device-resident path against the `compat_cov` host path.

Each runs as a child process (its own context, its own numpy generator) and must report 0 trials with differences."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,trials,seed", [("stress_refit.py", 60, 31), ("stress_decode.py", 60, 32),
                                                 ("stress_train_words.py", 40, 33), ("stress_ctrain.py", 20, 34)])
def test_randomised_sweep(script, trials, seed):
    env = dict(os.environ)
    for k in ("GMMHMM_REFIT", "GMMHMM_SEGSUM", "GMMHMM_VITERBI", "GMMHMM_REFIT_TAIL", "STRESS_ONLY", "STRESS_TRACE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), str(trials), str(seed)], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    tail = (p.stdout[-3000:] + "\n" + p.stderr[-2000:])
    assert p.returncode == 0, tail
    assert p.stdout.strip().splitlines()[-1] == "%d trials, 0 with differences" % trials, tail
