# -*- coding: utf-8 -*-
"""Fold the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- collected in SEPARATE runs of
`python3 bench.py ...`) into profiles/<tag>_pmc_traffic_<dtype>.json, which bench.py reports as
roofline.traffic.  gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE tallies
a 128-byte request as 64 bytes for coalesced streaming reads, so it is doubled; units are KB.

    python tools/pmc_traffic.py FETCH.csv WRITE.csv OUT.json N_FRAMES DTYPE"""
import csv, json, sys, collections

fetch_csv, write_csv, out, n_frames, dtype = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
esz = 8 if dtype == "f64" else 4


def collect(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


f, w = collect(fetch_csv, "FETCH_SIZE"), collect(write_csv, "WRITE_SIZE")
kernels = {}
for name in f:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if not (short.startswith("loglik") or short.startswith("viterbi")):
        continue
    fm, wm = sum(f[name]) / len(f[name]), sum(w[name]) / max(1, len(w[name]))
    alg = n_frames * esz * (39 + 50) if short.startswith("loglik") else n_frames * esz * 50
    kernels[short] = dict(FETCH_SIZE_KB_mean=fm, FETCH_SIZE_launches=len(f[name]), WRITE_SIZE_KB_mean=wm,
                          WRITE_SIZE_launches=len(w[name]), hbm_bytes_per_launch=(2 * fm + wm) * 1024,
                          algorithmic_bytes_per_launch=alg)
json.dump(dict(note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on `python3 bench.py` (C2, %s, %d "
                    "frames); units KB as reported; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B "
                    "requests at 64 B for coalesced streaming reads)" % (dtype, n_frames), kernels=kernels),
          open(out, "w"), indent=1)
print(json.dumps(kernels, indent=1))
