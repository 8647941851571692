# -*- coding: utf-8 -*-
"""`sr.core` -- the feature stacking that sits immediately in front of the recognition hot
path: delta features and the [cepstra | delta | delta-delta] ->
standardize tail of the reference's `load_wav_as_mfcc` (sr/core.py:13-22,41-44), on the GPU.
Reading wav files and computing MFCCs (sr/core.py:25-40, scipy / python_speech_features) and
the file-based train / test drivers are outside this repository's scope; their batched
equivalents on in-memory features are `sr.recognition.batch`."""
import numpy as np

from .recognition import _hip

__all__ = ["delta_feature", "stack_features", "stack_features_batch"]


def delta_feature(feat):
    """feat[i+1] - feat[i-1], one-sided at both ends (sr/core.py:13-22); [T, C] -> [T, C]."""
    feat = np.asarray(feat, dtype=np.float64)
    b = _hip.Batch(_hip.default_context(), cepstra=[feat], frontend_mode=1)
    try:
        C = feat.shape[1]
        return b.features()[0][:, C:2 * C].copy()
    finally:
        b.close()


def stack_features(ceps):
    """[ceps | delta | delta-delta], standardised per column: what `load_wav_as_mfcc` returns for
    one utterance once the MFCCs exist (sr/core.py:41-44)."""
    b = stack_features_batch([ceps])
    try:
        return b.features()[0].copy()
    finally:
        b.close()


def stack_features_batch(ceps_list, dtype=np.float64, device=None):
    """The same for many utterances in one launch; returns the resident `_hip.Batch` (D = 3C) ready
    for `loglik` / decoding -- the features never visit the host."""
    return _hip.Batch(_hip.default_context(device), cepstra=ceps_list, dtype=dtype, frontend_mode=0)
