# -*- coding: utf-8 -*-
"""Fold the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- collected in SEPARATE runs of the same command) into
profiles/<tag>_pmc_traffic_<dtype>.json, which bench.py reports as roofline.traffic.  gfx950 correction
(MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE tallies a 128-byte request as 64 bytes for coalesced
streaming reads, so it is doubled; units are KB.  Launches are grouped by (kernel, grid size): one command may launch a
kernel on workloads of different sizes.

    python tools/pmc_traffic.py FETCH.csv WRITE.csv OUT.json N_FRAMES DTYPE [STATES [DIM]]"""
import csv, json, sys, collections

fetch_csv, write_csv, out, n_frames, dtype = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
S = int(sys.argv[6]) if len(sys.argv) > 6 else 50
D = int(sys.argv[7]) if len(sys.argv) > 7 else 39
esz = 8 if dtype == "f64" else 4


def collect(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[(r["Kernel_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    return acc


f, w = collect(fetch_csv, "FETCH_SIZE"), collect(write_csv, "WRITE_SIZE")
kernels = {}
for (name, grid) in f:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if not (short.startswith("loglik") or short.startswith("viterbi") or short.startswith("lattice_backtrace")
            or short.startswith("bw_fused") or short.startswith("fb_chain") or short.startswith("fb_seq")
            or short.startswith("seq_backtrace") or short.startswith("fb_kernel") or short.startswith("cut_segments")):
        continue
    fv, wv = f[(name, grid)], w.get((name, grid), [0.0])
    fm, wm = sum(fv) / len(fv), sum(wv) / max(1, len(wv))
    if short.startswith("loglik"):
        alg = n_frames * esz * (D + S)               # features in once + likelihoods out once
    elif short.startswith("viterbi_seq") or short.startswith("fb_seq"):
        alg = None                                   # (only the states of the transcript are read: see the caller's note)
    elif short.startswith("viterbi"):
        alg = n_frames * (esz * S + 4)               # likelihoods in once + 4 B of path per frame
    else:
        alg = None
    key = short if sum(1 for (n2, g2) in f if n2 == name) == 1 else "%s [grid %s]" % (short, grid)
    kernels[key] = dict(FETCH_SIZE_KB_mean=fm, FETCH_SIZE_launches=len(fv), WRITE_SIZE_KB_mean=wm, WRITE_SIZE_launches=len(wv),
                        hbm_bytes_per_launch=(2 * fm + wm) * 1024, algorithmic_bytes_per_launch=alg)
json.dump(dict(note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes of the same command; %s, %d frames, "
                    "%d states, %d dims); units KB as reported; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies "
                    "128-B requests at 64 B for coalesced streaming reads); algorithmic bytes only apply to the launches "
                    "over the whole workload" % (dtype, n_frames, S, D), kernels=kernels),
          open(out, "w"), indent=1)
print(json.dumps(kernels, indent=1))
