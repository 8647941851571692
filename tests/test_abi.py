# -*- coding: utf-8 -*-
"""The C-ABI shared library loads without a GPU, exports every symbol include/gmmhmm.h
declares, the ctypes binding covers exactly that set, and the product path fails LOUDLY
(no CPU fallback) when no GPU is present."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, has_gpu


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gmmhmm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gh_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("gh_ctx_create", "gh_gmm_create", "gh_batch_create", "gh_loglik", "gh_lattices_create",
                 "gh_viterbi", "gh_dtw", "gh_kmeans_assign", "gh_em_accumulate", "gh_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol(built_library):
    lib = ctypes.CDLL(built_library)
    for name in declared_symbols():
        assert getattr(lib, name) is not None, name


def test_binding_matches_header(built_library):
    from sr.recognition import _hip
    assert sorted(_hip.SIGNATURES) == declared_symbols()
    lib = _hip.load_library(built_library)
    assert lib.gh_version() >= 1


def test_library_is_gfx950_code(built_library):
    blob = open(built_library, "rb").read()
    assert b"gfx950" in blob and b"gfx90a" not in blob and b"gfx942" not in blob


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu(built_library):
    from sr.recognition import _hip
    import sr.recognition as R
    lib = _hip.load_library(built_library)
    h = ctypes.c_void_p()
    assert lib.gh_ctx_create(0, ctypes.byref(h)) == -4  # GH_ERR_NODEVICE
    assert b"no HIP device" in lib.gh_last_error()
    with pytest.raises(_hip.BackendError):
        _hip.Context(0)
    g = R.GMM(np.zeros(3), np.ones(3), 2)
    with pytest.raises(_hip.BackendError):
        g.evaluate(np.zeros(3))
    with pytest.raises(_hip.BackendError):
        R.decode_hmm_states(np.zeros((4, 3)), [g, g], np.zeros((2, 2)))
    with pytest.raises(_hip.BackendError):
        R.kmeans(np.zeros((6, 3)), 2, np.zeros((2, 3)))


def test_missing_library_is_an_error(tmp_path):
    from sr.recognition import _hip
    with pytest.raises(_hip.BackendError):
        _hip.load_library(str(tmp_path / "libgmmhmm.so"))


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "speech-recognition_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                assert "oracle" not in open(os.path.join(d, f)).read(), os.path.join(d, f)
