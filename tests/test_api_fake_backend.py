# -*- coding: utf-8 -*-
"""The `sr.recognition` mirror API tests of tests/test_gpu_api.py, run on CPU with the oracle-backed
test double of the ctypes binding (tests/fake_hip.py).  This exercises the HOST LOGIC of the mirror
(object packing, lattice construction, end-point handling, segment bookkeeping, k-means / EM /
continuous-training loops, pickling, error mapping) against the reference's golden vectors in the
GPU-less tier; the GPU tier runs the very same functions against the real HIP library."""
import pytest

import fake_hip
import test_gpu_api as G


@pytest.fixture(autouse=True)
def fake_backend(monkeypatch, built_library):
    from sr.recognition import _hip, _pack
    fake_hip.install(monkeypatch, _hip)
    _pack._gmm_cache.clear()
    _pack._lat_cache.clear()
    yield
    _pack._gmm_cache.clear()
    _pack._lat_cache.clear()


@pytest.fixture(scope="module")
def R():
    import sr.recognition as R
    return R


# same test bodies, collected here without the module-level gpu mark of test_gpu_api
test_gmm_evaluate_and_pdf = G.test_gmm_evaluate_and_pdf
test_mahalanobis = G.test_mahalanobis
test_decode_hmm_states_isolated = G.test_decode_hmm_states_isolated
test_build_state_sequences_and_lattice_decode = G.test_build_state_sequences_and_lattice_decode
test_decode_edges = G.test_decode_edges
test_loop_grammar_decode = G.test_loop_grammar_decode
test_mfcc_features_matches_reference = G.test_mfcc_features_matches_reference
test_dtw = G.test_dtw
test_gmm_em = G.test_gmm_em
test_kmeans = G.test_kmeans
test_hmm_fit_single_gaussian = G.test_hmm_fit_single_gaussian
test_hmm_fit_gmm = G.test_hmm_fit_gmm
test_continuous_train = G.test_continuous_train
test_continuous_train_8mix = G.test_continuous_train_8mix
test_reference_pickle_scores_identically = G.test_reference_pickle_scores_identically
test_feature_stack_matches_reference = G.test_feature_stack_matches_reference
