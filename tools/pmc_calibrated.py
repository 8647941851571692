# -*- coding: utf-8 -*-
"""HBM read traffic of one kernel from rocprofv3's FETCH_SIZE, CALIBRATED on this chip in the kernel's own access width
(MI355X_MICROARCH.md, HBM section: the counter is documented for 16 B per lane only -- it reports half the bytes there --
and "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").

    python tools/pmc_calibrated.py CALIB.csv KERNEL.csv PATTERN COMPULSORY_BYTES [CALIB_KERNEL [WRITE.csv]] > out.json

CALIB.csv: counter_collection.csv of `rocprofv3 --pmc FETCH_SIZE ... -- tools/bin/hbm_stream calib` (kernels k_read,
k_read8, k_read8_tiles over 2 GiB each).  KERNEL.csv: the same counter for the workload.  The factor of CALIB_KERNEL
(default k_read8: 8 B per lane, contiguous) turns the kernel's FETCH_SIZE into bytes."""
import collections, csv, json, sys

calib_csv, kern_csv, pat, compulsory = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
calib_kernel = sys.argv[5] if len(sys.argv) > 5 else "k_read8"
write_csv = sys.argv[6] if len(sys.argv) > 6 else None
KNOWN = float(2 << 30)


def mean_by_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            acc[(name, r["Grid_Size"])].append(float(r["Counter_Value"]))
    return acc


cal = mean_by_kernel(calib_csv, "FETCH_SIZE")
factors = {}
for (name, grid), v in cal.items():
    if name.startswith("k_read"):
        factors.setdefault(name, []).extend(v)
factors = {k: KNOWN / (sum(v) / len(v) * 1024.0) for k, v in factors.items()}      # bytes per reported KB-unit byte
use = factors[calib_kernel] if calib_kernel in factors else None
out = dict(note="FETCH_SIZE in KB as rocprofv3 reports it; factor = known bytes / reported bytes of the calibration kernel "
                "(tools/hbm_stream.hip calib, 2 GiB per kernel, one pass); corrected = reported x factor",
           calibration_factors=factors, calibration_kernel=calib_kernel, kernels={})
ker = mean_by_kernel(kern_csv, "FETCH_SIZE")
wr = mean_by_kernel(write_csv, "WRITE_SIZE") if write_csv else {}
best = None
for (name, grid), v in sorted(ker.items()):
    if pat in name:
        m = sum(v) / len(v)
        w = wr.get((name, grid))
        e = dict(grid=grid, launches=len(v), FETCH_SIZE_KB_mean=m, reported_bytes=m * 1024.0,
                 corrected_read_bytes=None if use is None else m * 1024.0 * use, compulsory_read_bytes=compulsory)
        if e["corrected_read_bytes"]:
            e["read_over_compulsory"] = e["corrected_read_bytes"] / compulsory
        if w:
            e["WRITE_SIZE_KB_mean"] = sum(w) / len(w)
            e["write_bytes"] = e["WRITE_SIZE_KB_mean"] * 1024.0
        out["kernels"]["%s [grid %s]" % (name[-50:], grid)] = e
json.dump(out, sys.stdout, indent=1)
