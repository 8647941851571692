#!/usr/bin/env python3
"""Instruction mix of the hottest basic block of a kernel in a hipcc -S listing.
usage: isa_stats.py file.s <substring of the mangled kernel name> [opcode that marks the block, default v_add_f64]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
key = sys.argv[2]
mark = sys.argv[3] if len(sys.argv) > 3 else "v_add_f64"
for m in re.finditer(r"\n(_Z\w*" + re.escape(key) + r"\w*):", s):
    i = m.start(); j = s.index(".end_amdhsa_kernel", i); body = s[i:j]
    meta = {k: re.search(re.escape(k) + r"\s+(\S+)", body).group(1) for k in (".amdhsa_next_free_vgpr", ".amdhsa_next_free_sgpr", ".amdhsa_private_segment_fixed_size")}
    print(m.group(1)[:100], meta)
    best = (0, None)
    for b in re.split(r"\n(?=\.LBB\d+_\d+:)", body):
        ins = [l.strip() for l in b.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        n = sum(1 for x in ins if x.startswith(mark))
        if n > best[0]:
            best = (n, ins)
    if best[1]:
        c = Counter(x.split()[0] for x in best[1])
        print("  hottest block: %d instructions, %d VALU, %d SALU, %d mem" % (len(best[1]), sum(v for k, v in c.items() if k.startswith("v_")),
              sum(v for k, v in c.items() if k.startswith("s_")), sum(v for k, v in c.items() if k.startswith(("global_", "ds_", "buffer_", "scratch_", "flat_")))))
        print("  ", c.most_common(14))
