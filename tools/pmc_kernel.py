# -*- coding: utf-8 -*-
"""Average the counters of one rocprofv3 --pmc pass per kernel (largest grid of every kernel whose name contains PATTERN).
    python tools/pmc_kernel.py counter_collection.csv PATTERN"""
import csv, sys, collections
path, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    if pat in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size"], r.get("VGPR_Count", "?"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(acc.items()):
    print(key)
    for c, v in sorted(cs.items()):
        print("   %-28s %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
