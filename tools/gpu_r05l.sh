#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
rm -rf $O/prof_ct
rocprofv3 --kernel-trace --output-format csv -d $O/prof_ct -o ct -- python3 tools/time_ctrain.py 2000 7 4 > $O/ctrain_trace.log 2>&1; echo "exit $?"
f=$(find $O/prof_ct -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$f")) if "refit_" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last outer iteration only: take the last 230 launches
out=[]
for r in rows[-260:]:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    nm="km" if "refit_km_kernel" in r["Kernel_Name"] else ("em" if "refit_em_kernel" in r["Kernel_Name"] else "up")
    out.append("%s:%s:%.0f" % (nm, r.get("Grid_Size", r.get("Grid_Size_X", "?")), d))
print(" ".join(out))
PY
rm -rf $O/prof_ct
