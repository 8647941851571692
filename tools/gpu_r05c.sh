#!/bin/bash
# continuous_train: steady-state time per outer iteration + kernel stats
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
CTRAIN_PROFILE=0 timeout -k 10 600 python3 tools/time_ctrain.py 2000 7 8 > $O/ctrain.log 2>&1; echo "ctrain exit $?"; tail -4 $O/ctrain.log
rm -rf $O/prof_ctrain
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ctrain -o ct -- python3 tools/time_ctrain.py 2000 7 6 > $O/ctrain_prof.log 2>&1; echo "ctrain prof exit $?"
f=$(find $O/prof_ctrain -name "*kernel_stats.csv" | head -1); cp $f $O/ctrain_kernel_stats.csv
python3 - <<PY
import csv
tot = 0; n = 0
for r in csv.DictReader(open("$O/ctrain_kernel_stats.csv")):
    nm = r["Name"]
    nm = nm[nm.find("::")+2:] if "::" in nm else nm
    tot += float(r["TotalDurationNs"]); n += int(r["Calls"])
    if float(r["TotalDurationNs"]) > 0.4e6:
        print("%-60s calls %5s total %8.2f ms avg %8.1f us min %7.1f max %7.1f" % (nm[:60], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
print("total %.1f ms in %d launches (6 outer iterations)" % (tot / 1e6, n))
PY
rm -rf $O/prof_ctrain
