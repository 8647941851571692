#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_refit.py tests/test_gpu_lockstep.py tests/test_gpu_train_words.py tests/test_gpu_api.py -m gpu -x -q > $O/pytest_d.log 2>&1; echo "pytest exit $?"; tail -6 $O/pytest_d.log
CTRAIN_PROFILE=0 timeout -k 10 600 python3 tools/time_ctrain.py 2000 7 8 > $O/ctrain.log 2>&1; echo "ctrain exit $?"; tail -3 $O/ctrain.log
