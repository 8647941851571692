#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
GMMHMM_REFIT_TAIL=0 bash tools/gpu_r05m.sh 2>&1 | grep -E "prof k|refit_(em|km)_kernel|call"
timeout -k 10 600 python3 -m pytest tests/test_gpu_refit.py tests/test_gpu_lockstep.py tests/test_gpu_api.py -m gpu -x -q > $O/pytest_f.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_f.log
for cfg in "1 512" "0 512" "1 512" "0 512"; do
set -- $cfg
GMMHMM_REFIT_ITEM=$2 GMMHMM_REFIT_TAIL=$1 CTRAIN_PROFILE=0 timeout -k 10 600 python3 tools/time_ctrain.py 2000 7 10 > $O/ctrain_tail$1_$2.log 2>&1; echo "ctrain tail=$1 item=$2: $(tail -1 $O/ctrain_tail$1_$2.log)"
done
