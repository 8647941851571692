# -*- coding: utf-8 -*-
"""`sr.feature` -- the reference's feature extraction in front of the recognition hot path
(sr/feature/feature.py), on the GPU: `mfcc_features` (:43-82) and `standardize` (:85-88).
`mfcc_features` keeps the reference's signature (a wav path in, `(filter_banks, mfcc)` out);
`mfcc_from_signals` is the batched entry point on in-memory audio, and
`features_from_signals` chains MFCC -> delta -> delta-delta -> standardize on the device into a resident
batch (what `load_wav_as_mfcc`, sr/core.py:25-44, returns -- for many utterances, without a host round trip)."""
import numpy as np

from ..recognition import _hip

__all__ = ["standardize", "mfcc_features", "mfcc_from_signals", "features_from_signals"]


def standardize(data):
    """(data - mean) / std per column, population std.  Like the reference the mean is
    subtracted from the caller's array IN PLACE (`data -= mean`) and a new array is returned."""
    data -= np.mean(data, axis=0)
    ctx = _hip.default_context()
    b = _hip.Batch(ctx, cepstra=[data], frontend_mode=2)
    try:
        return b.features()[0].copy()
    finally:
        b.close()


def mfcc_from_signals(signals, sample_rate=16000, frame_size=0.025, frame_stride=0.01, low_freq=80, high_freq=None,
                      device=None):
    """mfcc_features for a list of 1-D sample arrays (int16 as read from a wav file, or float):
    returns ([T_u,40] log10 mel filterbank energies, [T_u,13] cepstra), one launch for all of them."""
    return _hip.mfcc(_hip.default_context(device), signals, sample_rate, (frame_size, frame_stride, low_freq, high_freq))


def mfcc_features(path_file, frame_size=0.025, frame_stride=0.01, low_freq=80, high_freq=None):
    """feature.py:43-82: wav file -> (filter_banks [T,40], mfcc [T,13])."""
    from scipy.io import wavfile
    sample_rate, signal = wavfile.read(path_file)
    fb, mf = mfcc_from_signals([signal], sample_rate, frame_size, frame_stride, low_freq, high_freq)
    return fb[0], mf[0]


def features_from_signals(signals, sample_rate=16000, dtype=np.float64, device=None, **mfcc_kw):
    """Audio in, resident 39-dimensional batch out: MFCC -> [ceps | delta | delta-delta] -> standardize,
    all on the device; returns the `_hip.Batch` ready for `loglik` / decoding."""
    prm = (mfcc_kw.get("frame_size", 0.025), mfcc_kw.get("frame_stride", 0.01), mfcc_kw.get("low_freq", 80),
           mfcc_kw.get("high_freq"))
    return _hip.Batch(_hip.default_context(device), pcm=signals, sample_rate=sample_rate, mfcc_params=prm, dtype=dtype)
