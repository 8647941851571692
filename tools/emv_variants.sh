#!/bin/bash
# Diagnostic builds of the lock-step tile kernels (timing only, results wrong): what a full pass of em_multi_kernel spends
# its time on.  usage (on the GPU box): bash tools/emv_variants.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ${EMV_LIST:-"" "-DEMV_NOSTAGE" "-DEMV_NOPHASE1" "-DEMV_NOEXP" "-DEMV_NOPHASE2" "-DEMV_NOPHASE1 -DEMV_NOPHASE2" "-DEMV_NOSTAGE -DEMV_NOPHASE1 -DEMV_NOPHASE2"}; do
  touch speech-recognition_amd/csrc/gh_lockstep.hip
  GMMHMM_EXTRA_FLAGS="$v" python3 speech-recognition_amd/build.py > /dev/null
  rm -rf gpurun_out/prof_emv
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_emv -o v -- python3 tools/time_em_multi.py > /dev/null 2>&1
  f=$(find gpurun_out/prof_emv -name "*kernel_stats.csv" | head -1)
  echo "== variant '${v}': $(python3 -c "
import csv, sys
for r in csv.DictReader(open('$f')):
    if 'em_multi_kernel' in r['Name'] or 'kmeans_multi_kernel' in r['Name']:
        print(r['Name'][27:46], '%.1f us (max %.1f) |' % (float(r['AverageNs']) / 1e3, float(r['MaxNs']) / 1e3), end=' ')
")"
done
rm -rf gpurun_out/prof_emv
touch speech-recognition_amd/csrc/gh_lockstep.hip
python3 speech-recognition_amd/build.py > /dev/null
