# -*- coding: utf-8 -*-
"""MI355X-native drop-in for the `sr` package's recognition core.

Only `sr.recognition` (the GMM-HMM hot path) is provided; the reference's wav /
MFCC / audio-capture facade (sr/core.py) is out of scope.  The names the
reference re-exports from `sr` (sr/__init__.py:2) are re-exported here too.
"""
from .core import delta_feature  # noqa: F401  (reference sr/__init__.py:1; the wav / file drivers are out of scope)
from .recognition import HMMState, HMM, decode_hmm_states, GMM, build_state_sequences, NES  # noqa: F401
