# -*- coding: utf-8 -*-
"""`sr.recognition` -- GMM-HMM training / decoding core, MI355X-native.

Same importable names as the reference package (sr/recognition/__init__.py:1-5):
dtw, decode_hmm_states, HMM, kmeans, skmeans, align_gmm_states, calc_transition_costs,
get_segments_from_path, segment_data, combine_templates, calc_variance, cluster_centroids,
build_state_sequences, continuous_train, GMM, NES, HMMState, MultivariateNormal, mahalanobis.
All likelihood and dynamic-programming arithmetic runs in HIP kernels through
libgmmhmm.so (see `_hip.py`); there is no CPU fallback.  Batched entry points that
the one-utterance-at-a-time reference API cannot express live in `sr.recognition.batch`.
"""
from .decode import *  # noqa: F401,F403
from .hmm import *  # noqa: F401,F403
from .kmeans import *  # noqa: F401,F403
from .continuous_speech import *  # noqa: F401,F403
from .hmm_state import *  # noqa: F401,F403


def set_compat(underflow=True, device=None):
    """Reference-compatible corner cases of the kernels (extension; see `gh_ctx_set_compat` in include/gmmhmm.h).
    underflow=True (the default of every context): a state whose every weighted density underflows fp64 costs +inf -- the
    reference's `GMM.evaluate` sums in the linear domain (hmm_state.py:114-120) -- so `decode_hmm_states` / `HMM.evaluate` /
    the batch decoders treat such frames as unreachable exactly where the reference does.  underflow=False keeps the
    likelihoods in the log domain: finite costs however far a frame lies from every mean (useful for features that were
    not standardised; not what the reference computes).  Applies to the default context of `device`;
    GMMHMM_COMPAT=0 in the environment switches it off for every new context."""
    from . import _hip
    _hip.default_context(device).set_compat(underflow=underflow)
