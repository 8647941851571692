#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
for k in 8 4; do
  rm -rf $O/prof_refit
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_refit -o r -- python3 tools/time_refit.py $k > $O/refit_prof_k$k.log 2>&1; echo "prof k$k exit $?"
  grep -E "equal|call" $O/refit_prof_k$k.log
  f=$(find $O/prof_refit -name "*kernel_stats.csv" | head -1); cp $f $O/refit_k${k}_kernel_stats.csv
  python3 - <<PY
import csv
for r in csv.DictReader(open("$O/refit_k${k}_kernel_stats.csv")):
    n = r["Name"]
    n = n[n.find("::")+2:] if "::" in n else n
    if "refit" in n:
        print("%-60s calls %5s total %8.2f ms avg %8.1f us min %7.1f max %7.1f" % (n[:60], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done
rm -rf $O/prof_refit
