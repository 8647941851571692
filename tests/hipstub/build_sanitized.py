# -*- coding: utf-8 -*-
"""TEST-ONLY build: the HOST halves of speech-recognition_amd/csrc/*.hip under AddressSanitizer + UndefinedBehaviorSanitizer,
linked against tests/hipstub/hipstub.cpp instead of the HIP runtime -> libgmmhmm_san.so (no GPU, no device code).

    python tests/hipstub/build_sanitized.py [out_dir]      (default: <tmp>/gmmhmm_san_build: nothing of it belongs in the tree)

hipcc --cuda-host-only compiles the product sources as they are (kernels become launch stubs); every translation unit
refers to its device image as an external `__hip_fatbin_<hash>`, which a generated C file defines as a few zero bytes."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "speech-recognition_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SAN = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]
FLAGS = ["-O1", "-g", "-std=c++17", "--offload-arch=gfx950", "--cuda-host-only", "-fPIC", "-Wno-unused-value", "-Wno-unused-result",
         "-Wno-pass-failed"] + SAN


def asan_runtime():
    """The shared ASan runtime of hipcc's clang (to LD_PRELOAD into an uninstrumented python)."""
    out = subprocess.run([HIPCC, "-print-file-name=libclang_rt.asan-x86_64.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    if out and os.path.exists(out):
        return out
    hits = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    return hits[0] if hits else None


def build(out_dir=None, verbose=False):
    import tempfile
    out_dir = out_dir or os.environ.get("GMMHMM_SAN_DIR") or os.path.join(tempfile.gettempdir(), "gmmhmm_san_build_%d" % os.getuid())
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, "libgmmhmm_san.so")
    headers = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(ROOT, "include", "gmmhmm.h")]
    newest_h = max(os.path.getmtime(h) for h in headers)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + [os.path.join(HERE, "hipstub.cpp")]
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(out_dir, os.path.splitext(os.path.basename(s))[0] + ".o")
        objs.append(o)
        if not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), newest_h):
            jobs.append([HIPCC] + FLAGS + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("sanitized build failed:\n%s\n%s" % (" ".join(cmd), r.stdout[-4000:]))

    with ThreadPoolExecutor(max_workers=int(os.environ.get("GMMHMM_SAN_JOBS", "6"))) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(lib):
        syms = subprocess.run(["nm", "-u"] + objs, stdout=subprocess.PIPE, text=True).stdout.split()
        fat = sorted({s for s in syms if s.startswith("__hip_fatbin_")})
        fat_c = os.path.join(out_dir, "fatbin_syms.c")
        with open(fat_c, "w") as f:
            f.write("/* generated: the device images the host halves refer to (never read: __hipRegisterFatBinary is a stub) */\n")
            for s in fat:
                f.write("const char %s[64] = {0};\n" % s)
        run(["gcc", "-fPIC", "-c", fat_c, "-o", os.path.join(out_dir, "fatbin_syms.o")])
        # (linked by clang++ itself, not through hipcc: the HIP runtime must not even be a dependency of this library)
        clangxx = os.path.join(os.path.dirname(asan_runtime() or ""), "..", "..", "..", "..", "..", "bin", "clang++")
        clangxx = os.path.normpath(clangxx) if os.path.exists(os.path.normpath(clangxx)) else "/opt/rocm/lib/llvm/bin/clang++"
        run([clangxx, "-shared", "-fPIC", "-shared-libsan"] + SAN + ["-o", lib] + objs + [os.path.join(out_dir, "fatbin_syms.o")])
    return lib


if __name__ == "__main__":
    print(build(sys.argv[1] if len(sys.argv) > 1 else None, verbose=True))
    print("asan runtime:", asan_runtime())
