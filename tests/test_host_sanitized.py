# -*- coding: utf-8 -*-
"""ASan / UBSan over the HOST layer of libgmmhmm (sanitizer row): ~5 000 lines of
host C++ inside csrc/*.hip -- plan builders, the chunk planner, UploadLayout, transcript expansion, session set-up -- only
ever ran on the GPU box, where no sanitizer is available.  Here their host halves are compiled as they are
(hipcc --cuda-host-only -fsanitize=address,undefined), linked against a test-only stand-in for the HIP runtime
(tests/hipstub/hipstub.cpp: device memory = host memory, launches = no-ops) and driven through the product's own ctypes
binding with random shapes (tests/hipstub/drive.py: ragged / empty / one-frame batches, graphs and transcripts, the chunk
planner under GMMHMM_SCRATCH_BUDGET=8M, EM and refit sessions, segmental k-means, front-end).  No GPU needed.

First run of this test (round 5) found a read past fc.skip_c[16] in gh_lattices_create for graphs with more than 16
emitting rows (gh_lattice.hip); fixed there."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "hipstub"))


@pytest.fixture(scope="module")
def sanitized():
    import build_sanitized as bs
    if not os.path.exists(bs.HIPCC):
        pytest.skip("hipcc not installed")
    rt = bs.asan_runtime()
    if rt is None:
        pytest.skip("clang's shared ASan runtime not found")
    lib = bs.build()
    syms = subprocess.run(["nm", "-D", "--undefined-only", lib], stdout=subprocess.PIPE, text=True).stdout
    assert "__asan_init" in syms and "__ubsan_handle" in syms, "the library is not instrumented"
    assert "hipModuleLaunchKernel" not in syms and "libamdhip64" not in subprocess.run(["ldd", lib], stdout=subprocess.PIPE, text=True).stdout
    return lib, rt


@pytest.mark.parametrize("seed", [101, 102])
def test_host_layer_runs_clean_under_asan_and_ubsan(sanitized, seed):
    lib, rt = sanitized
    env = dict(os.environ, LD_PRELOAD=rt, GMMHMM_LIB=lib, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    for k in ("GMMHMM_SCRATCH_BUDGET", "GMMHMM_REFIT", "GMMHMM_KMEANS_EXACT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipstub", "drive.py"), "150", str(seed)], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    report = p.stderr[-6000:]
    assert "AddressSanitizer" not in p.stderr and "runtime error:" not in p.stderr, report
    assert p.returncode == 0, report
    last = p.stdout.strip().splitlines()[-1]
    assert last.startswith("driven 150 cases"), p.stdout[-2000:]
    launches = int(last.split("steps did), ")[1].split()[0])
    assert launches > 1000            # (the cases reached the launch sites: the plans behind them were built)
