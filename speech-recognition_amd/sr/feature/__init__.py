# -*- coding: utf-8 -*-
"""`sr.feature` -- only the part that feeds the recognition hot path: `standardize`
(reference: sr/feature/feature.py:85-88).  The wav -> MFCC extraction itself
(`mfcc_features`, feature.py:43-82) is outside this repository's scope."""
import numpy as np

from ..recognition import _hip

__all__ = ["standardize"]


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
