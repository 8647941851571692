#!/bin/bash
# sweep of the fused Baum-Welch statistics kernel's workgroups per CU (GMMHMM_BWF_WGS): time per EM iteration
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in 3 4 6 9 12; do
  echo "== GMMHMM_BWF_WGS=$w"
  GMMHMM_BWF_WGS=$w timeout -k 10 200 python3 tools/time_em.py 12500 2>&1 | grep -E "device-resident|bw_stats_ms" | sed 's/.*"bw_stats_ms": \([0-9.]*\).*/bw_stats_ms \1/'
done
