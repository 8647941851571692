#!/bin/bash
# round 5, first contact of the streaming refit kernels with the GPU: equality with the tile kernels, timing, tests
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 300 python3 tools/time_refit.py 4 > $O/refit_k4.log 2>&1; echo "refit k4 exit $?"; cat $O/refit_k4.log
timeout -k 10 300 python3 tools/time_refit.py 8 > $O/refit_k8.log 2>&1; echo "refit k8 exit $?"; cat $O/refit_k8.log
timeout -k 10 300 python3 tools/time_refit.py 2 6 500 13 > $O/refit_k2_small.log 2>&1; echo "refit k2 small exit $?"; cat $O/refit_k2_small.log
