#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
for cfg in "4 256" "4 512" "4 1024" "8 512"; do
  set -- $cfg; k=$1; item=$2
  rm -rf $O/prof_refit
  GMMHMM_REFIT_ITEM=$item rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_refit -o r -- python3 tools/time_refit.py $k > $O/refit_prof_k${k}_i$item.log 2>&1; echo "prof k$k item $item exit $?"
  grep -E "equal|rel diff|call" $O/refit_prof_k${k}_i$item.log
  f=$(find $O/prof_refit -name "*kernel_stats.csv" | head -1); cp $f $O/refit_k${k}_i${item}_kernel_stats.csv
  python3 - <<PY
import csv
for r in csv.DictReader(open("$O/refit_k${k}_i${item}_kernel_stats.csv")):
    n = r["Name"]
    n = n[n.find("::")+2:] if "::" in n else n
    if float(r["TotalDurationNs"]) > 1.5e6 or "refit" in n:
        print("%-60s calls %5s total %8.2f ms avg %8.1f us min %7.1f max %7.1f" % (n[:60], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done
rm -rf $O/prof_refit
