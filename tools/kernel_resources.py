#!/usr/bin/env python3
"""Register use and occupancy of every kernel, from the compiler's own remarks (-Rpass-analysis=kernel-resource-usage):
a kernel that sits a few registers over an occupancy cliff (VGPRs + AGPRs > 256 -> one wave per SIMD, > 128 -> two, ...)
shows up here, not in a profile.  usage: kernel_resources.py [file.hip ...]   (default: every file under csrc/)"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "speech-recognition_amd", "csrc")
files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
KEYS = {"VGPRs": "V", "AGPRs": "A", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occ",
        "LDS Size [bytes/block]": "lds"}
for f in files:
    extra = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"] if f.endswith("gh_loglik_mfma.hip") else []
    p = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-value",
                        "-Wno-unused-result", "-Rpass-analysis=kernel-resource-usage", "-c", f, "-o", "/dev/null"] + extra,
                       capture_output=True, text=True)
    cur, rows = None, []
    for line in p.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        for key, short in KEYS.items():
            m = re.search(r"remark:\s+" + re.escape(key) + r": (\d+)", line)
            if m and cur is not None:
                cur[short] = int(m.group(1))
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    for r, nm in zip(rows, names):
        nm = nm.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        print("%-22s %-64s V=%3d A=%3d occ=%d scratch=%d lds=%d" % (os.path.basename(f), nm[:64], r.get("V", 0), r.get("A", 0),
                                                                   r.get("occ", 0), r.get("scratch", 0), r.get("lds", 0)))
