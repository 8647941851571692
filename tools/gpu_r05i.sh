#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_seq.py tests/test_gpu_layers.py -m gpu -x -q > $O/pytest_seq.log 2>&1; echo "pytest exit $?"; tail -6 $O/pytest_seq.log
timeout -k 10 300 python3 tools/time_seq16.py 4000 4 16 > $O/seq16.log 2>&1; echo "seq16 exit $?"; cat $O/seq16.log
timeout -k 10 300 python3 tools/time_seq16.py 4000 4 12 > $O/seq12.log 2>&1; echo "seq12 exit $?"; cat $O/seq12.log
