# -*- coding: utf-8 -*-
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "speech-recognition_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")

for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    """Asked through the library itself: a HIP runtime dlopen'ed on the side would be a second runtime in the process."""
    lib = os.path.join(PKG, "lib", "libgmmhmm.so")
    if not os.path.exists(lib):
        return False
    import ctypes
    try:
        fn = ctypes.CDLL(lib).gh_device_count
    except (OSError, AttributeError):
        return False
    fn.restype = ctypes.c_int
    return fn() > 0


@pytest.fixture(scope="session")
def built_library():
    """libgmmhmm.so, cross-compiled for gfx950 if it is not there yet (no GPU needed)."""
    lib = os.path.join(PKG, "lib", "libgmmhmm.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()
    return lib
