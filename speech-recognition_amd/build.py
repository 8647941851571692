# -*- coding: utf-8 -*-
"""Build libgmmhmm.so (hand-written HIP for gfx950) in-tree with hipcc.

    python speech-recognition_amd/build.py [--force]

Objects go to speech-recognition_amd/build/, the library to
speech-recognition_amd/lib/libgmmhmm.so (git-ignored; it travels to the GPU box
with the gpurun snapshot).  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "lib", "libgmmhmm.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-value", "-Wno-unused-result", "-Wno-pass-failed"]
FLAGS += os.environ.get("GMMHMM_EXTRA_FLAGS", "").split()   # diagnostic builds (e.g. -DGH_MF_TIMING); remember to rebuild
# per-file extras: MFMA results straight into VGPRs (no v_accvgpr_read/write around the epilogue): +1.5 % measured
EXTRA = {"gh_loglik_mfma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
         "gh_bw_fused.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
         # multiply-adds contracted per source expression, not across statements after optimisation: the ordinary and the TAIL
         # instantiation of a refit kernel then round alike (hipcc's default, fast, left them 1e-15 apart)
         "gh_refit_mfma.hip": ["-ffp-contract=on"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "gmmhmm.h"))
    jobs = []
    objs = []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([HIPCC] + FLAGS + EXTRA.get(src, []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stdout))
        return r.stdout

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


def build_hostcopy(force=False, verbose=True):
    """The CPython helper for the host-side copies (csrc/hostcopy.c -> sr/recognition/_hostcopy.<abi>.so), with the system
    compiler; in-tree like the library, so that it travels with the snapshot."""
    import sysconfig
    src = os.path.join(CSRC, "hostcopy.c")
    out = os.path.join(HERE, "sr", "recognition", "_hostcopy" + sysconfig.get_config_var("EXT_SUFFIX"))
    if force or _stale(out, [src]):
        cmd = [os.environ.get("CC", "gcc"), "-O2", "-shared", "-fPIC", "-Wall", "-I" + sysconfig.get_paths()["include"], src, "-o", out, "-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("building _hostcopy failed:\n%s\n%s" % (" ".join(cmd), r.stdout))
    return out


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
    print(build_hostcopy(force="--force" in sys.argv))
