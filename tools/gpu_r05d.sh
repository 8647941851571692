#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_lockstep.py tests/test_gpu_train_words.py tests/test_gpu_api.py tests/test_gpu_e2e.py -m gpu -x -q > $O/pytest_b.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest_b.log
for item in 512 1024; do
GMMHMM_REFIT_ITEM=$item CTRAIN_PROFILE=0 timeout -k 10 600 python3 tools/time_ctrain.py 2000 7 8 > $O/ctrain_i$item.log 2>&1; echo "ctrain item $item exit $?"; tail -3 $O/ctrain_i$item.log
done
timeout -k 10 500 python3 tools/time_ctrain.py 2000 7 8 > $O/ctrain_hostprof.log 2>&1; echo "hostprof exit $?"
